mkdir -p gpurun_out
timeout 300 python tools/dev/floor_latency.py 2>&1 | grep -E "^(array|hash)|Error|error" | cut -c1-120 | tail
timeout 300 python bench.py --workload c3 --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3',{k:round(d[k],4) for k in ('ms_per_step','raycast_p50_ms','update_occupancy_p50_ms','update_esdf_p50_ms')})"
timeout 300 python bench.py --workload c4 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4',{k:round(d[k],4) for k in ('ms_per_step','observe_p50_ms','update_occupancy_p50_ms','update_esdf_p50_ms')})"
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pt.log 2>&1; grep -E "passed|failed|error" gpurun_out/pt.log | tail -3; grep -E "^(FAILED|ERROR)" gpurun_out/pt.log | head -20
