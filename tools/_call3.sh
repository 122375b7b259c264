#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_dense_parity.py tests/test_gpu_cells.py tests/test_gpu_fuzz.py tests/test_gpu_sharded.py tests/test_gpu_checkpoint.py -q -x 2>&1 | grep -aE "passed|failed|^FAILED|^ERROR" | tail -4
bash tools/_call.sh 2>&1 | tail -2
