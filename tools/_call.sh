set -u
mkdir -p gpurun_out/r03e
export FIESTA_BENCH_ALL_RANKS_ON_GPU0=1
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --grid 128 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03e/two_ranks_one_gpu.log 2>&1
echo rc=$?
grep -iE "error|duplicate|invalid|metric" gpurun_out/r03e/two_ranks_one_gpu.log | cut -c1-300 | head -12
