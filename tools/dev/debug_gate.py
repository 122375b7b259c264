import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, fiesta_amd
from scenarios import *
n=40; res=0.1
m = fiesta_amd.ESDFMap((0,0,0), res, (n*res,)*3, update_engine="bulk")
m.SetParameters(*P_DEFAULT); m.SetOriginalRange()
def cyc(o, f, k):
    for _ in range(k):
        if len(o): m.SetOccupancy(o, 1, want_ret=False)
        if len(f): m.SetOccupancy(f, 0, want_ret=False)
        m.UpdateOccupancy(True)
def show(tag):
    st = m.UpdateESDF(); print(tag, {k: st[k] for k in ("inserted","deleted","bulk","observed_voxels","occupied_voxels","rounds")})
rng = np.random.RandomState(2)
m.SetOccupancy(all_voxels(n), 0, want_ret=False); m.UpdateOccupancy(True); show("observe")
S = rng.randint(1, n-1, (200,3)).astype(np.int32)
cyc(S, [], 3); show("insert")
m.SetUpdateRange((0.0,0.0,0.0),(1.9,n*0.1,n*0.1))
low = S[S[:,0]<18]
cyc(rng.randint(2,16,(30,3)).astype(np.int32), low[:40], 6); show("windowed")
m.SetOriginalRange()
new = rng.randint(1,n-1,(60,3)).astype(np.int32)
cyc(new, S[100:140], 6); show("wide")
occ = np.argwhere(m.download_field(("occ",))["occ"].reshape((n,)*3)==1).astype(np.int32)
cyc([], occ, 6); show("delete all")
print("occupied now", int(m.download_field(("occ",))["occ"].sum()))
cyc(S[:50], [], 3); show("insert again")
