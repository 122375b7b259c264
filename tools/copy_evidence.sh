#!/usr/bin/env bash
# After `gpurun -- tools/collect_evidence.sh TAG` (whose own copies into profiles/ stay on the GPU box): the same copies, here.
set -uo pipefail
TAG=${1:?round tag}
R=gpurun_out/$TAG
cp "$R/stats/bench_kernel_stats.csv" "profiles/${TAG}_bench_kernel_stats_default.csv"
cp "$R/stats_partial/partial_kernel_stats.csv" "profiles/${TAG}_c2_partial_kernel_stats.csv" 2>/dev/null
for f in "$R"/*.json; do cp "$f" "profiles/${TAG}_$(basename "$f")"; done
for f in floor_latency.txt pytest_gpu.txt smoke.txt envelope_reports.jsonl; do [ -f "$R/$f" ] && cp "$R/$f" "profiles/${TAG}_$f"; done
cp "$R/stats_c3/c3_kernel_stats.csv" "profiles/${TAG}_c3_kernel_stats.csv" 2>/dev/null
cp "$R/stats_c4/c4_kernel_stats.csv" "profiles/${TAG}_c4_kernel_stats.csv" 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE SQ LDS; do
  [ -f "$R/pmc_$C/bench_counter_collection.csv" ] && cp "$R/pmc_$C/bench_counter_collection.csv" "profiles/${TAG}_pmc_${C}_counter_collection.csv"
done
ls profiles | grep "^${TAG}_" | wc -l
