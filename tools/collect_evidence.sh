#!/usr/bin/env bash
# Reproduces the files under profiles/ for one round tag (run ON a GPU box, from the repo root):
#     tools/collect_evidence.sh r02a
# rocprofv3 wants a writable TMPDIR; counters are collected in their own passes (never together with sys/hip traces).
set -euo pipefail
TAG=${1:?round tag, e.g. r02a}
R=gpurun_out/$TAG
mkdir -p "$R"
export TMPDIR=/tmp
# C2 headline under the kernel trace (the JSON line carries roofline + cpu_baseline)
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/stats" -o bench -- python bench.py > "$R/bench_default.log" 2>&1
grep metric "$R/bench_default.log" > "$R/bench_default.json"
# HBM traffic of the dominant kernel: two PMC passes, calibrated inside the same runs (tools/pmc_traffic.py)
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$R/pmc_$C" -o bench -- \
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$R/pmc_$C.log" 2>&1
done
python tools/pmc_traffic.py "$R/pmc_FETCH_SIZE" "$R/pmc_WRITE_SIZE" k_relax_q > "$R/pmc_traffic_k_relax_q.json"
# the other configurations
python bench.py --workload c3 --steps 20 --warmup 4 2>&1 | grep metric > "$R/bench_c3.json"
python bench.py --workload c4 --steps 40 --warmup 5 2>&1 | grep metric > "$R/bench_c4.json"
# copy what is to be judged into profiles/ (gpurun_out/ is scratch)
cp "$R/stats/bench_kernel_stats.csv" "profiles/${TAG}_bench_kernel_stats_default.csv"
cp "$R/bench_default.json" "profiles/${TAG}_bench_default.json"
cp "$R/bench_c3.json" "profiles/${TAG}_bench_c3.json"
cp "$R/bench_c4.json" "profiles/${TAG}_bench_c4.json"
cp "$R/pmc_FETCH_SIZE/bench_counter_collection.csv" "profiles/${TAG}_pmc_FETCH_SIZE_counter_collection.csv"
cp "$R/pmc_WRITE_SIZE/bench_counter_collection.csv" "profiles/${TAG}_pmc_WRITE_SIZE_counter_collection.csv"
cp "$R/pmc_traffic_k_relax_q.json" "profiles/${TAG}_pmc_traffic_k_relax_q.json"
echo "evidence for $TAG written; describe it in profiles/README.md"
