#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out/r5a
mkdir -p $R; export TMPDIR=/tmp; rm -rf $R/trace
python -m pytest tests/test_gpu_dense_parity.py tests/test_gpu_cells.py tests/test_gpu_fuzz.py -q -x 2>&1 | grep -aE "passed|failed" | tail -2
rocprofv3 --kernel-trace --output-format csv -d $R/trace -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/trace.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r5a/trace/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "k_nn_fill" in r["Kernel_Name"]]
a,b=idx[-2],idx[-1]
t0=int(rows[a]["End_Timestamp"])
for r in rows[a:b+1]:
    print(f'{(int(r["Start_Timestamp"])-t0)/1e3:9.1f} +{(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:7.1f}  {r["Kernel_Name"].split("(")[0][:60]}')
PY
