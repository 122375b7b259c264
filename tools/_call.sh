mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_raycast_parity.py -q -m gpu -x -k "long_rays or frames_counts" > gpurun_out/pt.log 2>&1; echo "rc=$?"; grep -E "passed|failed|rror" gpurun_out/pt.log | tail -8
