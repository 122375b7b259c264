#!/bin/bash
# Counter passes of the frontier rounds (k_relax_q): HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes, calibrated in-run by
# tools/pmc_traffic.py) and the issue / wait / LDS counters, on C2 with the rounds pinned and on C2-partial (27 % of the map
# never observed: the transform's gate is shut).   usage (GPU box): tools/pmc_rounds.sh r05f   -> gpurun_out/r05f/
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r05f}
R=gpurun_out/$TAG
mkdir -p "$R"
export TMPDIR=/tmp
for W in rounds partial; do
  if [ $W = rounds ]; then ARGS="--engine rounds"; else ARGS="--unobserved 0.27"; fi
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf "$R/pmc_${W}_$C"
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$R/pmc_${W}_$C" -o bench -- \
      python bench.py --steps 3 --warmup 1 --no-cpu-baseline $ARGS > "$R/pmc_${W}_$C.log" 2>&1
  done
  python tools/pmc_traffic.py "$R/pmc_${W}_FETCH_SIZE" "$R/pmc_${W}_WRITE_SIZE" k_relax_q > "$R/pmc_traffic_k_relax_q_$W.json" 2> "$R/pmc_traffic_$W.err"
  rm -rf "$R/pmc_${W}_SQ" "$R/pmc_${W}_LDS"
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    --output-format csv -d "$R/pmc_${W}_SQ" -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline $ARGS > "$R/pmc_${W}_SQ.log" 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --output-format csv -d "$R/pmc_${W}_LDS" -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline $ARGS > "$R/pmc_${W}_LDS.log" 2>&1
done
python - "$R" <<'PY'
import collections, csv, glob, json, sys
R = sys.argv[1]
out = {}
for W in ("rounds", "partial"):
    acc = collections.defaultdict(list)
    for P in ("SQ", "LDS"):
        for f in glob.glob(f"{R}/pmc_{W}_{P}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_relax_q" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out[W] = {c: {"per_launch_mean": sum(v) / len(v), "launches": len(v)} for c, v in acc.items()}
json.dump(out, open(f"{R}/pmc_k_relax_q_counters.json", "w"), indent=1)
print(json.dumps({w: {c: round(d["per_launch_mean"] / 1e6, 2) for c, d in v.items()} for w, v in out.items()}))
PY
# keep what travels back small: the per-dispatch csv files only
find "$R" -name "*.db" -delete 2>/dev/null
du -sh "$R"
