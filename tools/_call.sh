cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_surf -o s --output-format csv -- python bench.py --scene surfaces --no-cpu-baseline --steps 10 > gpurun_out/surf.json 2>/dev/null
FIESTA_HIP_FT_S0=16 python bench.py --scene surfaces --no-cpu-baseline --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('S0=16', d['update_esdf_p50_ms'], d['roofline']['phases_p50_ms'], d['roofline']['ring_overflows'])"
FIESTA_HIP_FT_S0=32 python bench.py --scene surfaces --no-cpu-baseline --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('S0=32', d['update_esdf_p50_ms'], d['roofline']['phases_p50_ms'], d['roofline']['ring_overflows'])"
