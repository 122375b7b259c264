#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_level_grid.py -q -m gpu 2>&1 | grep -aE "passed|failed|AssertionError|^E  " | head -30 | cut -c1-900
