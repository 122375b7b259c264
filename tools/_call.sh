mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pt.log 2>&1; grep -E "passed|failed|error" gpurun_out/pt.log | tail -3; grep -E "^(FAILED|ERROR)|^E " gpurun_out/pt.log | head -20
python bench.py --gpus 1 --force-sharded --no-cpu-baseline --steps 10 2>/dev/null | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sharded1', d['ms_per_step'], d['update_esdf_p50_ms'], d['roofline']['frac'], d['verify']['mismatches'])"
