"""dev: C3's depth frames through the ray cast of a HASH map (no bench line covers that): p50 per stage."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fiesta_amd
from scenarios import render_depth, yaw_pose, P_DEFAULT, INTRINSICS as intr
m = fiesta_amd.ESDFMap((0.0, 0.0, 0.0), 0.1, reserve_size=1000000, mode="hash")
m.SetParameters(*P_DEFAULT); m.SetOriginalRange()
spheres = [((1.5, 0.5, 0.0), 0.5), ((-1.0, 2.0, 0.3), 0.7), ((0.5, -2.0, -0.5), 0.4), ((-2.0, -1.0, 0.5), 0.6), ((2.2, -1.8, 0.2), 0.3)]
lc, rc = (-25.6,) * 3, (25.6,) * 3
t = {"ray": [], "occ": [], "esdf": []}
for f in range(24):
    T = yaw_pose(2.0 * f, (0.0, 0.0, 0.0))
    depth = render_depth(T, rows=480, cols=640, spheres=spheres, intr=intr)
    m.synchronize()
    t0 = time.perf_counter()
    m.RaycastDepth(depth, intr["fx"], intr["fy"], intr["cx"], intr["cy"], T, T[:3, 3], 0.5, 5.0, lc, rc, dedup=1)
    m.synchronize()
    t1 = time.perf_counter()
    m.UpdateOccupancy(True); m.synchronize()
    t2 = time.perf_counter()
    m.UpdateESDF()
    t3 = time.perf_counter()
    if f >= 4:
        t["ray"].append((t1 - t0) * 1e3); t["occ"].append((t2 - t1) * 1e3); t["esdf"].append((t3 - t2) * 1e3)
print("hash map, 640x480 depth frames: p50 ms", {k: round(float(np.median(v)), 3) for k, v in t.items()})
