"""dev: per-phase counters of the relaxation kernel on the C3 depth-frame stream (FIESTA_HIP_PROF=1|3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fiesta_amd
from scenarios import render_depth, yaw_pose, P_DEFAULT, INTRINSICS as intr
G, res = 512, 0.1
half = G * res / 2
origin, size = (-half,) * 3, (G * res,) * 3
m = fiesta_amd.ESDFMap(origin, res, size)
m.SetParameters(*P_DEFAULT); m.SetOriginalRange()
spheres = [((1.5, 0.5, 0.0), 0.5), ((-1.0, 2.0, 0.3), 0.7), ((0.5, -2.0, -0.5), 0.4), ((-2.0, -1.0, 0.5), 0.6), ((2.2, -1.8, 0.2), 0.3)]
lc, rc = origin, tuple(np.add(origin, size))
acc = None; n = 0
for f in range(24):
    T = yaw_pose(2.0 * f, (0.0, 0.0, 0.0))
    depth = render_depth(T, rows=480, cols=640, spheres=spheres, intr=intr)
    m.RaycastDepth(depth, intr["fx"], intr["fy"], intr["cx"], intr["cy"], T, T[:3, 3], 0.5, 5.0, lc, rc, dedup=1)
    m.UpdateOccupancy(True)
    st = m.UpdateESDF()
    if f >= 4:
        row = np.array([st["tile_visits"], st["sweeps"], st["voxel_writes"], st["rounds"], st["relax_ms"] * 1e3, st["inserted"], st["deleted"]] + list(st["prof"]), float)
        acc = row if acc is None else acc + row; n += 1
acc /= n
names = ["tile_visits", "levels(sum)", "voxel_writes", "rounds", "relax_us", "inserted", "deleted", "p0", "p1", "p2", "p3", "p4", "p5", "p6", "p7"]
for a, b in zip(names, acc): print(f"{a:14s} {b:12.1f}")
