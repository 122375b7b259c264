#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4i
show() { python - "$1" "${@:2}" <<'PY'
import json,sys
d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric')][-1]
print(sys.argv[1], {k:d.get(k) for k in sys.argv[2:]})
PY
}
timeout 900 python -m pytest tests/test_gpu_level_grid.py tests/test_gpu_hash_parity.py tests/test_gpu_fuzz.py tests/test_gpu_raycast_parity.py -q -m gpu 2>&1 | grep -aE "^(FAILED|ERROR)|[0-9]+ (passed|failed)|^E  " | head -30
timeout 300 python bench.py --workload c4 --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r4i/c4.json 2> gpurun_out/r4i/c4.err; show gpurun_out/r4i/c4.json ms_per_step update_esdf_p50_ms update_esdf
timeout 300 python bench.py --workload c3 --steps 20 --warmup 4 --no-cpu-baseline > gpurun_out/r4i/c3.json 2> gpurun_out/r4i/c3.err; show gpurun_out/r4i/c3.json ms_per_step update_esdf_p50_ms
