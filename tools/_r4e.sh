#!/bin/bash
mkdir -p gpurun_out/r4e
cd $GRAFT_REPO_ROOT
show() { python - "$1" "${@:2}" <<'PY'
import json,sys
d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric')][-1]
print({k:d.get(k) for k in sys.argv[2:]})
PY
}
timeout 300 python bench.py --workload c3 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4e/c3.json 2> gpurun_out/r4e/c3.err; show gpurun_out/r4e/c3.json ms_per_step update_esdf_p50_ms level_trace_of_a_median_frame
tail -2 gpurun_out/r4e/c3.err
