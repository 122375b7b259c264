#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python tools/dev/abort_experiment.py 2>&1 | grep -v "amdgpu.ids\|new_size"
