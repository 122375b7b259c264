#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out/r5a
mkdir -p $R
for W in c3 c4; do
for E in auto rounds; do
timeout 300 python bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline --engine $E > $R/bench_${W}_$E.json 2> $R/bench_${W}_$E.err
python - $R/bench_${W}_$E.json <<'PY'
import json,sys
d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric')][-1]
print(sys.argv[1], {k: (round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k in ("ms_per_step","update_esdf_p50_ms","frame_p50_ms","raycast_p50_ms")}, d.get("esdf", d.get("update_esdf", {})) if False else "", (d.get("map_update") or d.get("esdf_summary") or {}).get("engine"))
PY
done; done
