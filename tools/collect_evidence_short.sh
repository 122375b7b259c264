#!/usr/bin/env bash
# The end-of-round subset of tools/collect_evidence.sh (same commands, same file names): the parity suite with the envelope log,
# the headline as the driver runs it, its kernel stats and counter passes, the shard lines, smoke.   tools/collect_evidence_short.sh r05g
set -uo pipefail
TAG=${1:?round tag}
R=gpurun_out/$TAG
mkdir -p "$R"
export TMPDIR=/tmp
export FIESTA_ENVELOPE_LOG=$PWD/$R/envelope_reports.jsonl
rm -f "$FIESTA_ENVELOPE_LOG"
python -m pytest tests -x -q -m gpu > "$R/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$R/pytest_gpu.log"
grep -aE "^(FAILED|ERROR)|[0-9]+ (passed|failed)|pytest rc" "$R/pytest_gpu.log" > "$R/pytest_gpu.txt"
unset FIESTA_ENVELOPE_LOG
python -c "import __graft_entry__ as g; g.smoke()" > "$R/smoke.txt" 2>&1; echo "smoke rc=$?" >> "$R/smoke.txt"
python bench.py > "$R/bench_default.log" 2>&1
grep '^{"metric' "$R/bench_default.log" > "$R/bench_default.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/stats" -o bench -- python bench.py --no-cpu-full > "$R/bench_default_profiled.log" 2>&1
grep '^{"metric' "$R/bench_default_profiled.log" > "$R/bench_default_profiled.json"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$R/pmc_$C" -o bench -- \
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$R/pmc_$C.log" 2>&1
done
python tools/pmc_traffic.py "$R/pmc_FETCH_SIZE" "$R/pmc_WRITE_SIZE" k_nn_ > "$R/pmc_traffic_cells.json"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --output-format csv -d "$R/pmc_SQ" -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$R/pmc_SQ.log" 2>&1
python bench.py --gpus 1 --force-sharded --no-cpu-baseline 2>&1 | grep '^{"metric' > "$R/bench_sharded_1rank.json"
python bench.py --gpus 1 --force-sharded --grid 1024 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | grep '^{"metric' > "$R/bench_1024_one_shard.json"
python tools/c5_smoke.py > "$R/c5_two_shards_bulk.json" 2> "$R/c5_smoke.err"
python bench.py --workload c3 --steps 20 --warmup 4 --no-cpu-baseline 2>&1 | grep '^{"metric' > "$R/bench_c3.json"
python bench.py --workload c4 --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | grep '^{"metric' > "$R/bench_c4.json"
cp "$R/stats/bench_kernel_stats.csv" "profiles/${TAG}_bench_kernel_stats_default.csv"
for f in bench_default bench_default_profiled pmc_traffic_cells bench_sharded_1rank bench_1024_one_shard c5_two_shards_bulk bench_c3 bench_c4; do
  cp "$R/$f.json" "profiles/${TAG}_$f.json"
done
cp "$R/pytest_gpu.txt" "profiles/${TAG}_pytest_gpu.txt"
cp "$R/smoke.txt" "profiles/${TAG}_smoke.txt"
cp "$R/envelope_reports.jsonl" "profiles/${TAG}_envelope_reports.jsonl"
for C in FETCH_SIZE WRITE_SIZE SQ; do
  cp "$R/pmc_$C/bench_counter_collection.csv" "profiles/${TAG}_pmc_${C}_counter_collection.csv"
done
find "$R" -name "*.db" -delete 2>/dev/null
echo "evidence for $TAG written"
