"""dev: the floor of one UpdateESDF -- a single new obstacle next to an existing one (one cheap tile visit) -- dense and hash."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fiesta_amd
from scenarios import P_DEFAULT
for mode in ("array", "hash"):
    if mode == "array":
        m = fiesta_amd.ESDFMap((-25.6,) * 3, 0.1, (51.2,) * 3)
    else:
        m = fiesta_amd.ESDFMap((0.0, 0.0, 0.0), 0.1, reserve_size=1000000, mode="hash")
    m.SetParameters(*P_DEFAULT); m.SetOriginalRange()
    base = np.array([[100, 100, 100]], np.int32) if mode == "array" else np.array([[40, 40, 40]], np.int32)
    # a small observed block with one obstacle
    blk = np.stack(np.meshgrid(*[np.arange(-6, 7)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.int32) + base
    for _ in range(3):
        m.SetOccupancy(blk, np.zeros(len(blk), np.int32), want_ret=False); m.UpdateOccupancy(True); m.UpdateESDF()
    ts = {"occ": [], "esdf": [], "dev": [], "relax": []}
    for k in range(40):
        v = base + np.array([[k % 5 - 2, (k // 5) % 5 - 2, k // 25]], np.int32)
        for cyc in range(6):
            m.SetOccupancy(v, np.ones(1, np.int32), want_ret=False)
            m.synchronize()
            t0 = time.perf_counter(); m.UpdateOccupancy(True); m.synchronize(); t1 = time.perf_counter()
            st = m.UpdateESDF(); t2 = time.perf_counter()
            if k >= 8 and st["inserted"] == 1:
                ts["occ"].append((t1 - t0) * 1e6); ts["esdf"].append((t2 - t1) * 1e6); ts["dev"].append(st["device_ms"] * 1e3); ts["relax"].append(st["relax_ms"] * 1e3)
                last = st
    print(mode, {k: round(float(np.median(v)), 1) for k, v in ts.items()}, "us; in-kernel ns", last["prof"][0], "last stats:", len(ts["esdf"]), {k: last[k] for k in ("inserted", "deleted", "rounds", "tile_visits", "relax_launches", "sweeps", "voxel_writes")}, "prof", list(last["prof"]))
