#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
show() { python - "$1" "${@:2}" <<'PY'
import json,sys
d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric')][-1]
print(sys.argv[1], {k:d.get(k) for k in sys.argv[2:]})
PY
}
for e in auto rounds levels; do
timeout 300 python bench.py --workload c4 --steps 40 --warmup 5 --engine $e --no-cpu-baseline > gpurun_out/r4f/c4_$e.json 2> gpurun_out/r4f/c4_$e.err; show gpurun_out/r4f/c4_$e.json ms_per_step update_esdf_p50_ms update_esdf parity
done
timeout 300 python bench.py --workload c3 --steps 20 --warmup 4 --engine levels --no-cpu-baseline > gpurun_out/r4f/c3_levels.json 2> gpurun_out/r4f/c3_levels.err; show gpurun_out/r4f/c3_levels.json ms_per_step update_esdf_p50_ms update_esdf
timeout 300 python bench.py --gpus 1 --force-sharded --no-cpu-baseline > gpurun_out/r4f/sharded1.json 2> gpurun_out/r4f/sharded1.err; show gpurun_out/r4f/sharded1.json ms_per_step update_esdf_p50_ms
