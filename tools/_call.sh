mkdir -p gpurun_out
timeout 900 python bench.py --gpus 1 --force-sharded --grid 1024 --no-cpu-baseline --steps 5 --warmup 1 2>gpurun_out/g1024.err | grep metric > gpurun_out/r03g_bench_1024_one_shard.json; python - <<'P'
import json
d=json.load(open('gpurun_out/r03g_bench_1024_one_shard.json'))
print(round(d['value']/1e9,1), d['ms_per_step'], d.get('update_esdf_p50_ms'), d['roofline']['frac'], d['roofline'].get('phases_p50_ms'), d['verify'])
P
tail -2 gpurun_out/g1024.err
