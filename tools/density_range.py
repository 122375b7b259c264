#!/usr/bin/env python3
"""The cell transform across obstacle densities: bench.py's C2 map with other obstacle counts, the library's
choice (`auto`) against each transform pinned.  One JSON document on stdout (profiles/rNN_density_range.json).  Run ON a GPU box:
    python tools/density_range.py [counts ...] > gpurun_out/density_range.json"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COUNTS = [7000, 10000, 13000, 17000, 25000, 50000, 100000, 150000, 250000, 330000, 500000, 670000]


def main():
    counts = [int(a) for a in sys.argv[1:]] or COUNTS
    out = {"command": "python bench.py --obstacles N --engine {auto,cells,envelope} --steps 8 --warmup 2 --no-cpu-baseline --verify-samples 2000",
           "runs": {}}
    for n in counts:
        for eng in ("auto", "cells", "envelope"):
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--obstacles", str(n), "--engine", eng, "--steps", "8", "--warmup", "2",
                                "--no-cpu-baseline", "--verify-samples", "2000"], capture_output=True, text=True, cwd=ROOT)
            line = [l for l in r.stdout.splitlines() if l.startswith('{"metric')]
            if not line:
                out["runs"][f"{n}_{eng}"] = {"error": r.stderr[-400:]}
                continue
            d = json.loads(line[-1])
            out["runs"][f"{n}_{eng}"] = {
                "obstacles": n, "density": n / 512.0 ** 3, "engine": eng, "update_esdf_p50_ms": d["update_esdf_p50_ms"], "frac": d["roofline"]["frac"],
                "engine_steps": d["roofline"]["engine_steps"], "phases_p50_ms": d["roofline"]["phases_p50_ms"],
                "verify_mismatches": d["verify"]["mismatches"] if d.get("verify") else None}
            print(n, eng, d["update_esdf_p50_ms"], d["roofline"]["engine_steps"], file=sys.stderr, flush=True)
    out["revision"] = open(os.path.join(ROOT, ".fiesta_rev")).read().strip() if os.path.exists(os.path.join(ROOT, ".fiesta_rev")) else None
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
