import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import fiesta_amd
from scenarios import P_DEFAULT
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
m = fiesta_amd.ESDFMap((0, 0, 0), 0.1, ((n - .5) * .1,) * 3, update_engine="cells")
m.SetParameters(*P_DEFAULT); m.SetOriginalRange()
m.SetOccupancyBox((0, 0, 0), (n - 1,) * 3, 0); m.UpdateOccupancy(True); m.UpdateESDF()
rng = np.random.RandomState(12345)
S = rng.randint(0, n, (int(50000 * (n / 512) ** 3), 3)).astype(np.int32)
for _ in range(3):
    m.SetOccupancy(S, 1, want_ret=False); m.UpdateOccupancy(True)
st = m.UpdateESDF()
print({k: st[k] for k in ("bulk", "cells", "nn_failed", "nn_entries", "nn_lists_ms", "nn_fill_ms", "nn_cells_ms")})
