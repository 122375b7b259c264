#!/bin/bash
# round 4, first GPU contact of the level engine
mkdir -p gpurun_out/r4a
cd $GRAFT_REPO_ROOT
timeout 300 python __graft_entry__.py smoke > gpurun_out/r4a/smoke.log 2>&1; echo "smoke rc=$?" 
tail -3 gpurun_out/r4a/smoke.log
timeout 900 python -m pytest tests/test_gpu_dense_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "levels" > gpurun_out/r4a/t_dense_levels.log 2>&1; echo "dense/fuzz levels rc=$?"
tail -5 gpurun_out/r4a/t_dense_levels.log
timeout 900 python -m pytest tests/test_gpu_raycast_parity.py tests/test_gpu_hash_parity.py -x -q -m gpu > gpurun_out/r4a/t_ray_hash.log 2>&1; echo "raycast/hash rc=$?"
tail -5 gpurun_out/r4a/t_ray_hash.log
timeout 120 python tools/dev/floor_latency.py > gpurun_out/r4a/floor.log 2>&1; cat gpurun_out/r4a/floor.log | tail -3
timeout 300 python bench.py --workload c3 --steps 20 --warmup 3 > gpurun_out/r4a/c3.json 2> gpurun_out/r4a/c3.err; tail -c 1500 gpurun_out/r4a/c3.json
timeout 300 python bench.py --workload c4 --steps 40 --warmup 5 > gpurun_out/r4a/c4.json 2> gpurun_out/r4a/c4.err; tail -c 1500 gpurun_out/r4a/c4.json
