#!/bin/bash
mkdir -p gpurun_out/r4b
cd $GRAFT_REPO_ROOT
show() { python - "$1" "${@:2}" <<'PY'
import json,sys
d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric')][-1]
print({k:d.get(k) for k in sys.argv[2:]})
PY
}
timeout 300 python __graft_entry__.py smoke > gpurun_out/r4b/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r4b/smoke.log
timeout 120 python tools/dev/floor_latency.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4b/floor.log | cut -c1-400
timeout 300 python bench.py --workload c3 --steps 20 --warmup 3 > gpurun_out/r4b/c3.json 2> gpurun_out/r4b/c3.err; show gpurun_out/r4b/c3.json ms_per_step raycast_p50_ms update_occupancy_p50_ms update_esdf_p50_ms
timeout 300 python bench.py --workload c4 --steps 40 --warmup 5 > gpurun_out/r4b/c4.json 2> gpurun_out/r4b/c4.err; show gpurun_out/r4b/c4.json ms_per_step update_occupancy_p50_ms update_esdf_p50_ms updated_voxels_per_frame
timeout 900 python -m pytest tests/test_gpu_dense_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "levels" > gpurun_out/r4b/t_dense_levels.log 2>&1; echo "dense/fuzz levels rc=$?"; grep -v new_size gpurun_out/r4b/t_dense_levels.log | tail -4
timeout 900 python -m pytest tests/test_gpu_raycast_parity.py tests/test_gpu_hash_parity.py -x -q -m gpu > gpurun_out/r4b/t_ray_hash.log 2>&1; echo "raycast/hash rc=$?"; grep -v new_size gpurun_out/r4b/t_ray_hash.log | tail -4
