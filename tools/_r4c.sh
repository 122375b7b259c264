#!/bin/bash
mkdir -p gpurun_out/r4c
cd $GRAFT_REPO_ROOT
show() { python - "$1" "${@:2}" <<'PY'
import json,sys
d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric')][-1]
print({k:d.get(k) for k in sys.argv[2:]})
PY
}
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 120 python tools/dev/floor_latency.py 2>&1 | grep -v amdgpu.ids | cut -c1-300
timeout 300 python bench.py --workload c3 --steps 20 --warmup 3 > gpurun_out/r4c/c3.json 2> gpurun_out/r4c/c3.err; show gpurun_out/r4c/c3.json ms_per_step raycast_p50_ms update_esdf_p50_ms update_esdf
timeout 300 python bench.py --workload c4 --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r4c/c4.json 2> gpurun_out/r4c/c4.err; show gpurun_out/r4c/c4.json ms_per_step update_esdf_p50_ms updated_voxels_per_frame update_esdf
timeout 900 python -m pytest tests/test_gpu_dense_parity.py tests/test_gpu_fuzz.py tests/test_gpu_bulk_gate.py -x -q -m gpu > gpurun_out/r4c/t_dense.log 2>&1; echo "dense/fuzz/gate rc=$?"; grep -v new_size gpurun_out/r4c/t_dense.log | tail -3
