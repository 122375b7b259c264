set -u
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03e/stats2 -o bench -- python bench.py --no-cpu-baseline --steps 5 --verify-samples 0 > /tmp/p.log 2>&1
grep -E "k_ft_" gpurun_out/r03e/stats2/bench_kernel_stats.csv | cut -c1-120
