mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dense_parity.py tests/test_gpu_hash_parity.py tests/test_gpu_fuzz.py tests/test_gpu_sharded.py -q -m gpu > gpurun_out/pt.log 2>&1; grep -E "passed|failed|error" gpurun_out/pt.log | tail -2
