mkdir -p gpurun_out
timeout 280 python tools/c5_smoke.py 2>gpurun_out/c5.err | tail -1 > gpurun_out/r03g_c5_two_shards_bulk.json; cat gpurun_out/r03g_c5_two_shards_bulk.json | cut -c1-900; tail -2 gpurun_out/c5.err
