set -x
python -m pytest tests/test_gpu_masked.py -x -q -m gpu -s 2>&1 | grep -E "passed|failed|differ|Error|error" | head -30
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/mk_prof -o partial -- python $GRAFT_REPO_ROOT/bench.py --unobserved 0.27 --no-cpu-baseline --steps 10 2>&1 | grep '^{"metric' > $GRAFT_REPO_ROOT/gpurun_out/mk.json
