"""dev: the level engine's hand-over to the frontier rounds -- find the voxels that did not push."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fiesta_amd
from oracle import pyoracle
from scenarios import P_DEFAULT, DIRS24, D2_INF
n = 24
origin, res = (-3.0, -3.0, -1.0), 0.25
m = fiesta_amd.ESDFMap(origin, res, (n * res,) * 3, update_engine=int(sys.argv[1]) if len(sys.argv) > 1 else 2)
m.SetParameters(*P_DEFAULT); m.SetOriginalRange()
rng = np.random.RandomState(5)
gs = m.grid_size
prev_occ = prev_d2 = None
for cycle in range(8):
    pos = np.array(origin) + (rng.rand(5000, 3) * 1.2 - 0.1) * n * 0.25
    occ = (rng.rand(5000) < 0.45).astype(np.int32)
    occ[::97] = 2
    m.SetOccupancy(pos, occ)
    m.UpdateOccupancy(True)
    st = m.UpdateESDF()
    f = m.download_field()
    d2, coc = f["d2"].astype(np.int64), f["coc"].astype(np.int64)
    idx = np.arange(len(d2))
    V = np.stack([idx // (gs[1] * gs[2]), (idx // gs[2]) % gs[1], idx % gs[2]], -1)
    fin = (d2 >= 0) & (d2 != D2_INF)
    bad = []
    for e in DIRS24:
        N = V + e
        ok = np.all((N >= 0) & (N < np.array(gs)), axis=1)
        ni = np.where(ok, (N[:, 0] * gs[1] + N[:, 1]) * gs[2] + N[:, 2], 0)
        finn = ok & (d2[ni] >= 0) & (d2[ni] != D2_INF)
        push = fin & finn & (((N - coc) ** 2).sum(-1) < d2[ni])
        for i in np.flatnonzero(push)[:3]:
            bad.append((tuple(V[i]), int(d2[i]), tuple(coc[i]), tuple(N[i]), int(d2[ni[i]]), tuple(coc[ni[i]])))
    print(cycle, {k: st[k] for k in ("inserted", "deleted", "rounds", "levels", "tile_visits", "voxel_writes")}, "prof", st["prof"], "violations", len(bad))
    for b in bad[:4]:
        was = prev_occ[(b[0][0] * gs[1] + b[0][1]) * gs[2] + b[0][2]] if prev_occ is not None else -1
        pn = prev_d2[(b[3][0] * gs[1] + b[3][1]) * gs[2] + b[3][2]] if prev_occ is not None else -1
        print("   v", tuple(map(int, b[0])), "d2", b[1], "was-occupied", int(was), "-> n", tuple(map(int, b[3])), "d2", b[4], "coc", tuple(map(int, b[5])), "n's d2 before", int(pn))
    prev_occ, prev_d2 = f["occ"].copy(), d2.copy()
