mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_raycast_parity.py tests/test_gpu_golden.py tests/test_cpp_facade.py -q -m gpu > gpurun_out/pt.log 2>&1; grep -E "passed|failed|error" gpurun_out/pt.log | tail -5; grep -E "^(FAILED|ERROR)" gpurun_out/pt.log | head
timeout 300 python bench.py --workload c3 --steps 20 --warmup 4 > gpurun_out/r03f_bench_c3.json 2> gpurun_out/c3.err; python - <<'P'
import json
d=json.load(open('gpurun_out/r03f_bench_c3.json'))
print({k:d[k] for k in ('value','ms_per_step','raycast_p50_ms','update_occupancy_p50_ms','update_esdf_p50_ms')}, d['cpu_baseline'].get('counters_bit_identical'))
P
tail -3 gpurun_out/c3.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c3 -o c3 --output-format csv -- python bench.py --workload c3 --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/c3p.json 2> gpurun_out/c3p.err
