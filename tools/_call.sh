mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pt.log 2>&1; grep -E "passed|failed|error" gpurun_out/pt.log | tail -3; grep -E "^(FAILED|ERROR)|^E " gpurun_out/pt.log | head -20
python bench.py --no-cpu-full 2>/dev/null | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['value']/1e9, d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['verify']['mismatches'])"
