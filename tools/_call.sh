python -m pytest tests/test_gpu_bulk_gate.py -m gpu -q --timeout 900 2>&1 | grep -v new_size | tail -5
