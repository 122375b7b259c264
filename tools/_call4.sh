#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
python -m pytest tests/test_gpu_cells.py -q -x 2>&1 | grep -aE "passed|failed" | tail -1
for S in 60; do
python bench.py --no-cpu-baseline --steps $S --warmup 2 > gpurun_out/r5a/bench_long.json 2> gpurun_out/r5a/bench_long.err
python - <<'PY'
import json
d=[json.loads(l) for l in open("gpurun_out/r5a/bench_long.json") if l.startswith('{"metric')][-1]
r=d["roofline"]
print("steps", d["steps"], "ms/step", round(d["ms_per_step"],4), "p50", round(d["update_esdf_p50_ms"],4), "frac", round(r["frac"],4), r["engine_steps"], r["phases_p50_ms"], d["verify"]["mismatches"])
PY
done
tail -2 gpurun_out/r5a/bench_long.err
