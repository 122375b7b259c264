#!/bin/bash
cd $GRAFT_REPO_ROOT
export FIESTA_REV=af560c7
timeout 2800 bash tools/collect_evidence.sh r04c > gpurun_out/r04c_collect.log 2>&1
tail -3 gpurun_out/r04c_collect.log
cat gpurun_out/r04c/pytest_gpu.txt
cut -c1-300 gpurun_out/r04c/bench_default.json
