"""dev: the one point of `bench.py --delta-sweep` where `auto` loses: 50 inserts + 50 deletes on the surfaces scene (512^3)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, fiesta_amd
G = 512
dev = torch.device("cuda", 0)
for engine in (sys.argv[1:] or ("auto", "rounds", "levels")):
    m = fiesta_amd.ESDFMap((0, 0, 0), 0.1, (G * 0.1,) * 3, update_engine=engine)
    m.SetParameters(*bench.P_DEFAULT); m.SetOriginalRange()
    m.SetOccupancyBox((0, 0, 0), (G - 1, G - 1, G - 1), 0); m.UpdateOccupancy(True); m.UpdateESDF()
    w = bench.Workload(G, 50000, scene="surfaces")
    def observe(vox, occ):
        v = torch.from_numpy(np.ascontiguousarray(vox, dtype=np.int32)).to(dev); o = torch.from_numpy(np.ascontiguousarray(occ, dtype=np.int32)).to(dev)
        m.SetOccupancyDevice(v.data_ptr(), o.data_ptr(), v.shape[0]); m.synchronize()
    for _ in range(3):
        observe(w.initial(), np.ones(50000, np.int32)); m.UpdateOccupancy(True)
    m.UpdateESDF()
    w.half = 50
    for r in range(4):
        new, old = w.next_step()
        for c in range(3):
            if c < 2: observe(new, np.ones(len(new), np.int32))
            else: observe(np.concatenate([new, old]), np.concatenate([np.ones(len(new), np.int32), np.zeros(len(old), np.int32)]))
            m.UpdateOccupancy(True)
        m.synchronize(); t0 = time.perf_counter(); st = m.UpdateESDF(); t1 = time.perf_counter()
    print(engine, "host ms %.3f" % ((t1 - t0) * 1e3), {k: st[k] for k in ("inserted", "deleted", "invalidated", "rounds", "tile_visits", "relax_launches", "voxel_writes", "levels", "grid_levels", "bulk")}, "device_ms %.3f relax_ms %.3f" % (st["device_ms"], st["relax_ms"]), "prof", list(st["prof"]))
    print("   trace", m.level_trace())
    m.close()
