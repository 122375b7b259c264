python bench.py --no-cpu-baseline 2>&1 | grep metric > /tmp/b.json
python - <<'PY'
import json
d=json.load(open('/tmp/b.json'))
print("frac", d["roofline"]["frac"], "ms/step", d["ms_per_step"], "esdf p50", d["update_esdf_p50_ms"], "dev", d["update_esdf_device_p50_ms"], d["roofline"]["phases_p50_ms"], "verify", d["verify"]["mismatches"])
PY
python -m pytest tests/test_gpu_full_size.py tests/test_gpu_sharded.py -m gpu -q -x --timeout 900 2>&1 | grep -v new_size | tail -3
