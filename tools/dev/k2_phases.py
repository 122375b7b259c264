import os, sys, statistics
import numpy as np
sys.path.insert(0, "/root/repo")
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
        st = m.UpdateESDF(); out.append(st["nn_lists_ms"])
    return statistics.median(out[3:])
print("full", run(20))
os.environ["NNSTOP"] = "1"; print("staging only", run(20))
os.environ["NNSTOP"] = "2"; print("staging + competitor + sort", run(20))
del os.environ["NNSTOP"]; print("full", run(20))
