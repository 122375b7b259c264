"""dev: what do the two phase events inside the cell transform cost?  (a hack build reads NOEV)"""
import os, sys, time, statistics
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import fiesta_amd
from bench import P_DEFAULT
G, res = 512, 0.1
m = fiesta_amd.ESDFMap((0, 0, 0), res, (G * res,) * 3, update_engine="cells")
m.SetParameters(*P_DEFAULT); m.SetOriginalRange()
m.SetOccupancyBox((0, 0, 0), (G - 1,) * 3, 0); m.UpdateOccupancy(True); m.UpdateESDF()
S = np.random.RandomState(5).randint(0, G, (50000, 3)).astype(np.int32)
for _ in range(3):
    m.SetOccupancy(S, 1, want_ret=False); m.UpdateOccupancy(True)
m.UpdateESDF()
flip = np.array([[7, 9, 11]], np.int32)
ts, dev = [], []
for i in range(60):
    for _ in range(6):
        m.SetOccupancy(flip, (i + 1) & 1, want_ret=False); m.UpdateOccupancy(True)
    t0 = time.perf_counter(); st = m.UpdateESDF(); ts.append(time.perf_counter() - t0); dev.append(st["device_ms"])
    assert st["cells"] == 1
print("NOEV" if os.environ.get("NOEV") else "events", "host p50 %.4f ms, device p50 %.4f ms" % (statistics.median(ts[10:]) * 1e3, statistics.median(dev[10:])))
