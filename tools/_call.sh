mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c3g -o c3 --output-format csv -- python bench.py --workload c3 --steps 12 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c4g -o c4 --output-format csv -- python bench.py --workload c4 --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
ls gpurun_out/prof_c3g gpurun_out/prof_c4g | head
