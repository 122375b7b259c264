python -m pytest tests/test_gpu_sharded.py -m gpu -q --timeout 900 -k "2048_long or bulk_then" 2>&1 | grep -v new_size | tail -15
