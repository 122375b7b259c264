"""dev (CPU only): the depth-frame scenario of tests/test_levelsync_model.py under schedule 1 / 2 / 3 of the model."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import pyoracle
import test_levelsync_model as T
from scenarios import depth_to_points, render_depth, yaw_pose
kind = "ref" if pyoracle.available("ref", "array") else "port"
for sched in (1, 2, 3, 4, 5, 6):
    origin, size, res = (-6.4, -6.4, -3.2), (12.75, 12.75, 6.35), 0.1
    p = T.Pair(pyoracle, kind, origin, res, size, k=4)
    p.eng.set_schedule(sched)
    lc, rc = origin, tuple(np.array(origin) + np.array(size))
    spheres = [((1.5, 0.5, 0.0), 0.5), ((-1.0, 2.0, 0.3), 0.7), ((0.5, -2.0, -0.5), 0.4)]
    pos = np.array([0.13, -0.21, 0.05]); intr = dict(fx=96.1, fy=96.1, cx=80.7, cy=58.9)
    out = []
    for f in range(5):
        Tm = yaw_pose(20.0 * f, pos + 0.05 * f)
        pts = depth_to_points(render_depth(Tm, rows=120, cols=160, spheres=spheres, intr=intr), intr=intr)
        pts[::501] = np.nan
        o = Tm[:3, 3]
        p.both(lambda m: m.raycast_frame(pts, Tm, o, 0.5, 5.0, lc, rc)); p.fuse(); p.esdf()
        e = p.judge(); out.append((e["closer"], e["farther"], e["disagree"], max(e["leave_one_out"])))
    print("schedule", sched, "(closer, farther, disagree, leave-one-out) per frame:", out, flush=True)
