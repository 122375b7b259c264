import sys, numpy as np
sys.path.insert(0,'/root/repo')
import fiesta_amd
G=int(sys.argv[1]); reserve=int(sys.argv[2]); chunk=int(sys.argv[3])
m = fiesta_amd.ESDFMap((0, 0, 0), 0.1, reserve_size=reserve, mode="hash")
m.SetParameters(0.70, 0.35, 0.12, 0.97, 0.80); m.SetOriginalRange()
off=-G//2
for x0 in range(0, G, chunk):
    g = np.stack(np.meshgrid(np.arange(x0, x0 + chunk), np.arange(G), np.arange(G), indexing="ij"), -1).reshape(-1, 3)
    m.SetOccupancy((g + off).astype(np.int32), 0, want_ret=False)
    m.synchronize(); print("obs", x0, flush=True)
print("fuse", m.UpdateOccupancy(True), flush=True)
print("esdf", m.UpdateESDF()["rounds"], flush=True)
S = (np.random.RandomState(1).randint(0, G, (200, 3)) + off).astype(np.int32)
for _ in range(3):
    m.SetOccupancy(S, 1, want_ret=False); m.synchronize(); print("obs ok", flush=True); r=m.UpdateOccupancy(True); m.synchronize(); print("fuse", r, m.last_insert, flush=True)
print("esdf", m.UpdateESDF(), flush=True)
