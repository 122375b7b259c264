#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4n
show() { python - "$1" "${@:2}" <<'PY'
import json,sys
d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric')][-1]
print(sys.argv[1], {k:d.get(k) for k in sys.argv[2:]})
PY
}
cp fiesta_amd/libfiesta_hip.so /tmp/new.so
for v in new old new old; do
  cp /tmp/new.so fiesta_amd/libfiesta_hip.so
  [ $v = old ] && cp fiesta_amd/libfiesta_hip_old.so fiesta_amd/libfiesta_hip.so
  echo "== $v"
  timeout 300 python bench.py --workload c3 --steps 20 --warmup 4 --no-cpu-baseline > gpurun_out/r4n/c3_$v.json 2> gpurun_out/r4n/c3.err; show gpurun_out/r4n/c3_$v.json ms_per_step update_esdf_p50_ms
  timeout 100 python tools/dev/floor_latency.py 2>&1 | tail -2 | cut -c1-120
done
