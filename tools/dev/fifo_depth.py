"""how deep are the ORDER dependencies inside a layer of the reference's update queue?  CPU only.

The restatement (oracle/esdf_port.cpp, pinned id-for-id to the verbatim reference by tests/test_oracle_port_vs_ref.py -- it
runs the reference's FIFO order exactly) carries a probe in relax(): per layer of update_queue_ the number of processed
entries, how many of them read a word that an EARLIER entry of the SAME layer wrote, and the longest such chain.  This
script drives it with the frames of config 3 (640 x 480 depth images, reference intrinsics, yaw sweep) and of config 4 (hash
map, streaming boxes) and prints a histogram.  A layer of depth 1 is order-free; depth k needs k ordered sub-steps."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle  # noqa: E402
from scenarios import INTRINSICS, P_DEFAULT, c4_frame, depth_to_points, render_depth, yaw_pose  # noqa: E402


def summarise(name, rows, per_update):
    rows = np.asarray(rows, np.int64).reshape(-1, 3)
    depth = rows[:, 2]
    hist = np.bincount(np.minimum(depth, 64))
    out = {
        "workload": name, "updates": len(per_update), "layers": int(len(rows)), "entries": int(rows[:, 0].sum()),
        "entries_depending_on_an_earlier_entry_of_their_layer": int(rows[:, 1].sum()),
        "dependent_fraction": float(rows[:, 1].sum() / max(1, rows[:, 0].sum())),
        "layer_depth_max": int(depth.max()), "layer_depth_p50": float(np.median(depth)), "layer_depth_p90": float(np.percentile(depth, 90)),
        "layer_depth_histogram": {str(d): int(c) for d, c in enumerate(hist) if c},
        "sum_of_layer_depths_per_update_p50": float(np.median([u[0] for u in per_update])),
        "layers_per_update_p50": float(np.median([u[1] for u in per_update])),
    }
    print(json.dumps(out))
    return out


def c3(G=512, frames=6):
    res = 0.1
    half = G * res / 2
    origin, size = (-half, -half, -half), (G * res,) * 3
    cpu = pyoracle.OracleMap(origin, res, size, kind="port")
    cpu.SetParameters(*P_DEFAULT)
    cpu.SetOriginalRange()
    lc, rc = origin, tuple(np.add(origin, size))
    spheres = [((1.5, 0.5, 0.0), 0.5), ((-1.0, 2.0, 0.3), 0.7), ((0.5, -2.0, -0.5), 0.4), ((-2.0, -1.0, 0.5), 0.6),
               ((2.2, -1.8, 0.2), 0.3)]
    rows, per_update = [], []
    for f in range(frames):
        T = yaw_pose(2.0 * f, (0.0, 0.0, 0.0))
        depth = render_depth(T, rows=480, cols=640, spheres=spheres, intr=INTRINSICS)
        cpu.raycast_frame(depth_to_points(depth, INTRINSICS), T, T[:3, 3], 0.5, 5.0, lc, rc)
        cpu.UpdateOccupancy(True)
        cpu.fifo_probe(True)
        st = cpu.UpdateESDF()
        lay = cpu.fifo_layers()
        cpu.fifo_probe(False)
        if f == 0:
            continue   # (the first frame builds the map from nothing: not a sensor-rate update)
        rows.extend(lay.tolist())
        per_update.append((int(lay[:, 2].sum()), len(lay), int(st["inserted"]), int(st["deleted"])))
    cpu.close()
    return summarise(f"C3: {G}^3 @0.1 m, 640x480 depth frames, frames 2..{frames}", rows, per_update)


def c4(frames=12):
    cpu = pyoracle.OracleMap((0, 0, 0), 0.05, reserve_size=1000000, mode="hash", kind="port")
    cpu.SetParameters(*P_DEFAULT)
    cpu.SetOriginalRange()
    rows, per_update = [], []
    for k in range(frames):
        free_lo, free_hi, occ = c4_frame(k)
        from scenarios import box_voxels
        cpu.SetOccupancyVox(box_voxels(free_lo, free_hi), 0)
        if len(occ):
            cpu.SetOccupancyVox(occ, 1)
        cpu.UpdateOccupancy(True)
        cpu.fifo_probe(True)
        st = cpu.UpdateESDF()
        lay = cpu.fifo_layers()
        cpu.fifo_probe(False)
        if k == 0 or len(lay) == 0:
            continue
        rows.extend(lay.tolist())
        per_update.append((int(lay[:, 2].sum()), len(lay), int(st["inserted"]), int(st["deleted"])))
    cpu.close()
    return summarise(f"C4: hash map @0.05 m, streaming boxes, frames 2..{frames}", rows, per_update)


if __name__ == "__main__":
    pyoracle.build("port")
    out = [c3(int(sys.argv[1]) if len(sys.argv) > 1 else 512), c4()]
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)
