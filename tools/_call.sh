set -u
R=gpurun_out/r03d
mkdir -p $R
( time python -m pytest tests/test_gpu_full_size.py -m gpu -q --timeout 900 -k "tier or spill" 2>&1 | grep -v new_size | tail -5 ) 2>&1
python bench.py --delta-sweep --steps 3 2>&1 | grep metric > $R/delta_sweep.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03d/delta_sweep.json")); print("sweep worst", d["value"])
for e in d['table']:
    if e['engine']!='rounds': print(e['scene'], e['engine'], e['delta'], round(e['update_esdf_p50_ms'],3))
PY
