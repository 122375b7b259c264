#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out/r5a
mkdir -p $R
export TMPDIR=/tmp
rm -rf $R/pmc_SQ
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/pmc_SQ -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/pmc_SQ.log 2>&1
python - <<'PY'
import csv,glob,collections
for f in glob.glob("gpurun_out/r5a/pmc_SQ/**/*counter_collection.csv", recursive=True):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "k_nn" in k or "k_fuse" in k or "k_observe_vox" in k:
            acc[k.split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        print(k, {c.replace("SQ_",""): round(sum(x)/len(x)/1e6,2) for c,x in v.items()}, "n", len(next(iter(v.values()))))
PY
