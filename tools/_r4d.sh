#!/bin/bash
mkdir -p gpurun_out/r4d
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/r4d/envelope.jsonl
FIESTA_ENVELOPE_LOG=$PWD/gpurun_out/r4d/envelope.jsonl timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r4d/gputests.log 2>&1; echo "gpu suite rc=$?"
grep -v new_size gpurun_out/r4d/gputests.log | grep -E "^(FAILED|ERROR)|passed|failed" | head -40
