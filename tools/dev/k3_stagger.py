"""dev: do k_nn_fill_full's persistent waves run in lock step (all computing, then all storing)?  A hack build delays the
work-groups' start by (blockIdx % 8) * NNSTAG * ~1 us."""
import os, sys, statistics
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
k = [0]
def run(n):
    out = []
    for i in range(n):
        k[0] += 1
        for _ in range(6):
            m.SetOccupancy(flip, k[0] & 1, want_ret=False); m.UpdateOccupancy(True)
        st = m.UpdateESDF(); out.append(st["nn_fill_ms"])
    return statistics.median(out[3:])
for s in (0, 1, 2, 3, 0):
    os.environ["NNSTAG"] = str(s)
    print("stagger", s, "fill %.1f us" % (run(20) * 1e3))
