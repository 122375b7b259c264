"""dev (CPU only): hash fuzz seed 63, last step -- where the level-schedule model is closer than the FIFO restatement, and
what the neighbourhood of such a voxel looks like in both."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import pyoracle
from scenarios import P_DEFAULT, all_voxels, d2_from_dist, hash_key
sched = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.RandomState(63)
origin, res = tuple(float(v) for v in rng.uniform(-1, 1, 3)), float(rng.choice([0.05, 0.1]))
rng.choice([0, 1000, 50000])
maps = [pyoracle.OracleMap(origin, res, reserve_size=1000, mode="hash", kind="port") for _ in range(2)]
maps[1].set_schedule(sched)
for m in maps: m.SetParameters(*P_DEFAULT); m.SetOriginalRange()
centre = rng.randint(-30, 30, 3); live = np.zeros((0, 3), np.int32)
for step in range(5):
    centre = centre + rng.randint(-6, 7, 3); ext = rng.randint(8, 22, 3)
    box = (all_voxels(tuple(int(v) for v in ext)) + (centre - ext // 2)).astype(np.int32)
    new = box[rng.rand(len(box)) < 0.01]; gone = live[rng.rand(len(live)) < 0.4]
    for k in range(3):
        for m in maps:
            if k == 0: m.SetOccupancyVox(box, 0)
            for vv, o in ((new, 1), (gone, 0)):
                if len(vv): m.SetOccupancyVox(vv, o)
            m.UpdateOccupancy(True)
    if step == 4:
        pre = [m.dump_hash() for m in maps]
        print("last step: inserts", len(new), "deletes", len(gone), "box", box.min(0), box.max(0))
    for m in maps: m.UpdateESDF()
    live = np.concatenate([live, new]); rng.uniform(-25, 25, (150, 3))
D = [m.dump_hash() for m in maps]
def table(d):
    ok = d["vox"][:, 0] != -10000
    return {tuple(v): (float(dd), tuple(c), int(o)) for v, dd, c, o in zip(d["vox"][ok], d["dist"][ok], d["coc"][ok], d["occ"][ok])}
A, B = table(D[0]), table(D[1])
PA = table(pre[0])
diff = [v for v in A if v in B and abs(A[v][0] - B[v][0]) > 1e-9]
print("voxels that differ:", len(diff), "closer in the model:", sum(B[v][0] < A[v][0] for v in diff))
dirs = [(int(a), int(b), int(c)) for a, b, c in np.array(pyoracle.STENCIL24)] if hasattr(pyoracle, "STENCIL24") else None
for v in sorted(diff)[:6]:
    print("voxel", v, "FIFO:", A[v], "model:", B[v], "before the update:", PA.get(v))
    cm = B[v][1]
    holders = []
    for dx in range(-2, 3):
        for dy in range(-2, 3):
            for dz in range(-2, 3):
                u = (v[0] + dx, v[1] + dy, v[2] + dz)
                if u in A and (dx, dy, dz) != (0, 0, 0) and dx * dx + dy * dy + dz * dz <= 6:
                    holders.append((u, "FIFO coc", A[u][1], round(A[u][0] / res) if A[u][0] < 1e4 else None, "model coc", B[u][1], "pre", PA.get(u, (None, None))[1]))
    for h in holders:
        if h[2] == cm or h[5] == cm: print("    nb", h)
