set -u
python -m pytest tests/test_gpu_sharded.py -m gpu -q -x --timeout 900 2>&1 | grep -v new_size | tail -4
for mode in "" "--force-sharded"; do
python bench.py --no-cpu-baseline $mode 2>&1 | grep metric > /tmp/b.json
python - <<PY
import json
d=json.load(open('/tmp/b.json'))
print("mode [$mode] frac", round(d["roofline"]["frac"],4), "ms/step", round(d["ms_per_step"],4), "esdf p50", round(d["update_esdf_p50_ms"],4), "dev", round(d["update_esdf_device_p50_ms"],4), "verify", d["verify"]["mismatches"], d["verify"]["ranks_seen_by_rccl"])
PY
done
