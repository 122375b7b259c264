#!/usr/bin/env bash
# The smallest end-of-round record: the parity suite, smoke, the headline as the driver runs it and its kernel stats.
#     tools/collect_evidence_min.sh r05k
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
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/stats" -o bench -- python bench.py --no-cpu-baseline > "$R/bench_default_profiled.log" 2>&1
grep '^{"metric' "$R/bench_default_profiled.log" > "$R/bench_default_profiled.json"
find "$R" -name "*.db" -delete 2>/dev/null
echo "evidence for $TAG written"
