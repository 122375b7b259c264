#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4p
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/r4p/sq -o c3 -- python bench.py --workload c3 --steps 20 --warmup 4 --no-cpu-baseline > gpurun_out/r4p/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES --output-format csv -d gpurun_out/r4p/sq2 -o c3 -- python bench.py --workload c3 --steps 20 --warmup 4 --no-cpu-baseline > gpurun_out/r4p/sq2.log 2>&1
python - <<'PY'
import csv,glob,collections
for d in ('sq','sq2'):
    f=glob.glob('gpurun_out/r4p/%s/**/c3_counter_collection.csv'%d,recursive=True)
    if not f: print(d,'no file'); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    disp=set()
    for r in csv.DictReader(open(f[0])):
        k=r['Kernel_Name'].split('(')[0][-45:]
        if 'level' not in k: continue
        acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
        disp.add((k,r['Dispatch_Id']))
    for k in acc:
        n=len([1 for kk,_ in disp if kk==k])
        print(d,k,'dispatches',n,{c:round(v/n) for c,v in acc[k].items()})
PY
tail -3 gpurun_out/r4p/sq2.log | cut -c1-300
