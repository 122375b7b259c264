"""dev: where a level of k_level_grid spends its time -- a wide update (300 scattered inserts into an observed 128x128x64
map) on the level engine with 32 / 16 / 8 / 4 work-groups and without the grid (pairs of launches)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fiesta_amd
from scenarios import P_DEFAULT, all_voxels
for groups in (32, 16, 8, 4, 0):
    m = fiesta_amd.ESDFMap((0, 0, 0), 0.1, (12.8, 12.8, 6.4), update_engine="levels")
    m.level_tuning(groups, -1)
    m.SetParameters(*P_DEFAULT); m.SetOriginalRange()
    g = all_voxels(m.grid_size)
    m.SetOccupancy(g, np.zeros(len(g), np.int32), want_ret=False); m.UpdateOccupancy(True); m.UpdateESDF()
    rng = np.random.RandomState(1)
    rows = []
    for rep in range(6):
        S = (rng.randint(0, 1 << 20, (300, 3)) % np.array(m.grid_size)).astype(np.int32)
        for c in range(3):
            m.SetOccupancy(S, np.ones(len(S), np.int32), want_ret=False); m.UpdateOccupancy(True)
        m.synchronize(); t0 = time.perf_counter(); st = m.UpdateESDF(); t1 = time.perf_counter()
        rows.append(((t1 - t0) * 1e6, st))
    us, st = rows[-1]
    tr, nl = m.level_trace()
    print("groups", groups, "host us %.0f" % us, "device us %.0f" % (st["device_ms"] * 1e3), "levels", st["rounds"], "grid levels", st["grid_levels"], "launches", st["relax_launches"],
          "in-kernel us %.1f" % (st["prof"][0] / 1e3), "phases us", [round(st["prof"][k] / 1e3, 1) for k in (2, 3, 4, 5)], "entries", st["prof"][6], "peak", st["prof"][7])
    print("   trace", tr[:24])
