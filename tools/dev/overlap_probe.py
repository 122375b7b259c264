"""dev: do two cell transforms overlap?  Two 512^3 maps (own streams), the same scatter scene; UpdateESDF of both from two
threads against one after the other.  An upper estimate of what running k_nn_lists beside k_nn_fill would buy."""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import fiesta_amd
from bench import P_DEFAULT

G, res = 512, 0.1
maps = []
rng = np.random.RandomState(5)
S = rng.randint(0, G, (50000, 3)).astype(np.int32)
for k in range(2):
    m = fiesta_amd.ESDFMap((0, 0, 0), res, (G * res,) * 3, update_engine="cells")
    m.SetParameters(*P_DEFAULT)
    m.SetOriginalRange()
    m.SetOccupancyBox((0, 0, 0), (G - 1,) * 3, 0)
    m.UpdateOccupancy(True)
    m.UpdateESDF()
    for _ in range(3):
        m.SetOccupancy(S, 1, want_ret=False)
        m.UpdateOccupancy(True)
    st = m.UpdateESDF()
    assert st["cells"] == 1, st
    maps.append(m)

flip = np.array([[7, 9, 11]], np.int32)
state = [0]
def toggle(m, _):
    occ = state[0] & 1
    for _ in range(6):
        m.SetOccupancy(flip, occ, want_ret=False)
        m.UpdateOccupancy(True)

def seq(n):
    t = 0.0
    for i in range(n):
        state[0] += 1
        for m in maps: toggle(m, 0)
        t0 = time.perf_counter()
        for m in maps:
            st = m.UpdateESDF()
        t += time.perf_counter() - t0
        assert st["cells"] == 1
    return t / n

def par(n):
    t = 0.0
    for i in range(n):
        state[0] += 1
        for m in maps: toggle(m, 0)
        bar = threading.Barrier(3)
        def work(m):
            bar.wait(); m.UpdateESDF(); bar.wait()
        th = [threading.Thread(target=work, args=(m,)) for m in maps]
        for x in th: x.start()
        bar.wait(); t0 = time.perf_counter(); bar.wait(); t += time.perf_counter() - t0
        for x in th: x.join()
    return t / n

seq(5); par(5)
print("sequential: %.3f ms for two updates" % (seq(30) * 1e3))
print("concurrent: %.3f ms for two updates" % (par(30) * 1e3))
print("sequential: %.3f ms for two updates" % (seq(30) * 1e3))
