"""dev: per-phase cycle counters of the relaxation kernel on the C4 stream (FIESTA_HIP_PROF=1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import fiesta_amd
from scenarios import c4_frame, P_DEFAULT
m = fiesta_amd.ESDFMap((0.0, 0.0, 0.0), 0.05, reserve_size=1000000, mode="hash")
m.SetParameters(*P_DEFAULT); m.SetOriginalRange()
dev = torch.device("cuda", 0)
acc = None; n = 0
for k in range(30):
    lo, hi, occ = c4_frame(k)
    v = torch.from_numpy(occ).to(dev); o = torch.ones(len(occ), dtype=torch.int32, device=dev)
    m.SetOccupancyBox(lo, hi, 0); m.SetOccupancyDevice(v.data_ptr(), o.data_ptr(), len(occ)); m.synchronize()
    m.UpdateOccupancy(True)
    st = m.UpdateESDF()
    if k >= 8:
        row = np.array([st["tile_visits"], st["sweeps"], st["voxel_writes"], st["rounds"], st["relax_ms"] * 1e3] + list(st["prof"]), float)
        acc = row if acc is None else acc + row; n += 1
acc /= n
names = ["tile_visits", "levels(sum)", "voxel_writes", "rounds", "relax_us", "cyc_stage", "cyc_propagate", "cyc_writeback", "items", "cyc_compact", "cyc_process", "pulls", "succ"]
for a, b in zip(names, acc): print(f"{a:14s} {b:12.1f}")
v = acc[0]
print("per visit: levels %.1f  stage %.0f cyc  propagate %.0f cyc (compact %.0f, process %.0f)  writeback %.0f cyc  items %.0f" % (acc[1]/v, acc[5]/v, acc[6]/v, acc[9]/v, acc[10]/v, acc[7]/v, acc[8]/v))
