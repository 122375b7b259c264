#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4l
timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -aE "^(FAILED|ERROR)|[0-9]+ (passed|failed)|^E  " | head -30
timeout 600 python bench.py --delta-sweep --no-cpu-baseline 2>&1 | grep -E "^\{" > gpurun_out/r4l/delta_sweep.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4l/delta_sweep.json').readline())
print('worst', d['value'])
for e in d['table']:
    if e['engine']=='auto' and e['delta']<=500: print(e['scene'],e['delta'],round(e['update_esdf_p50_ms'],3),e.get('bulk_updates'),round(e.get('auto_slower_than_best_fixed',0),3))
PY
