#!/usr/bin/env python3
"""BASELINE config 3: 512^3 grid @0.1 m fed by 640x480 synthetic depth frames through the HIP ray cast
(fiesta_hip_raycast_depth) -> UpdateOccupancy -> UpdateESDF, single MI355X.

Scene: 6x6x3 m box room with 5 spheres, sensor at the grid centre, yaw sweep 2 deg/frame, reference intrinsics shape,
ray window 0.5-5.0 m, reference de-duplication semantics (dedup=1).  Prints one JSON line: rays/s, per-stage p50 ms,
end-to-end p50, and (with --cpu-frames K) the CPU oracle's per-stage time for the first K frames, checking that the
hit/miss counters of those frames are bit-identical.
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from scenarios import P_DEFAULT, depth_to_points, render_depth, yaw_pose  # noqa: E402

INTR = dict(fx=384.4, fy=384.4, cx=323.1, cy=235.5)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--cpu-frames", type=int, default=2)
    a = ap.parse_args()
    import fiesta_amd
    G, res = a.grid, 0.1
    half = G * res / 2
    origin, size = (-half, -half, -half), (G * res,) * 3
    m = fiesta_amd.ESDFMap(origin, res, size)
    m.SetParameters(*P_DEFAULT)
    m.SetOriginalRange()
    cpu = None
    if a.cpu_frames:
        from oracle import pyoracle
        pyoracle.build("port")
        kind = "ref" if pyoracle.available("ref", "array") else "port"
        cpu = pyoracle.OracleMap(origin, res, size, kind=kind)
        cpu.SetParameters(*P_DEFAULT)
        cpu.SetOriginalRange()
    spheres = [((1.5, 0.5, 0.0), 0.5), ((-1.0, 2.0, 0.3), 0.7), ((0.5, -2.0, -0.5), 0.4), ((-2.0, -1.0, 0.5), 0.6), ((2.2, -1.8, 0.2), 0.3)]
    frames = []
    for f in range(a.frames):
        T = yaw_pose(2.0 * f, (0.0, 0.0, 0.0))
        frames.append((T, render_depth(T, rows=480, cols=640, spheres=spheres, intr=INTR)))
    lc, rc = origin, tuple(np.add(origin, size))
    t_ray, t_fuse, t_esdf, t_all, cpu_t = [], [], [], [], []
    for f, (T, depth) in enumerate(frames):
        t0 = time.perf_counter()
        m.RaycastDepth(depth, INTR["fx"], INTR["fy"], INTR["cx"], INTR["cy"], T, T[:3, 3], 0.5, 5.0, lc, rc, dedup=1)
        m.synchronize()
        t1 = time.perf_counter()
        if cpu is not None and f < a.cpu_frames:
            pts = depth_to_points(depth, INTR)
            c0 = time.perf_counter()
            cpu.raycast_frame(pts, T, T[:3, 3], 0.5, 5.0, lc, rc)
            c1 = time.perf_counter()
            gh, gm = m.download_counts()
            ch, cm = cpu.dump_counts()
            assert np.array_equal(gm, cm) and np.array_equal(gh, ch), "hit/miss counters differ from the oracle"
            t1 = time.perf_counter()  # the download is not part of the pipeline
        m.UpdateOccupancy(True)
        m.synchronize()
        t2 = time.perf_counter()
        st = m.UpdateESDF()
        t3 = time.perf_counter()
        if cpu is not None and f < a.cpu_frames:
            c2 = time.perf_counter()
            cpu.UpdateOccupancy(True)
            c3 = time.perf_counter()
            sc = cpu.UpdateESDF()
            c4 = time.perf_counter()
            assert (st["inserted"], st["deleted"]) == (sc["inserted"], sc["deleted"])
            cpu_t.append({"raycast_ms": (c1 - c0) * 1e3, "fuse_ms": (c3 - c2) * 1e3, "esdf_ms": (c4 - c3) * 1e3})
        t_ray.append((t1 - t0) * 1e3 if not (cpu is not None and f < a.cpu_frames) else None)
        t_fuse.append((t2 - t1) * 1e3)
        t_esdf.append((t3 - t2) * 1e3)
        t_all.append((t3 - t0) * 1e3 if t_ray[-1] is not None else None)
    ray = [t for t in t_ray if t is not None]
    tot = [t for t in t_all if t is not None]
    out = {
        "config": f"C3: {G}^3 @0.1 m, {a.frames} frames of 640x480 depth (307200 rays), yaw 2 deg/frame, dedup=1",
        "raycast_p50_ms": statistics.median(ray), "rays_per_sec": 307200 / (statistics.median(ray) * 1e-3),
        "update_occupancy_p50_ms": statistics.median(t_fuse), "update_esdf_p50_ms": statistics.median(t_esdf),
        "frame_p50_ms": statistics.median(tot), "frames_per_sec": 1e3 / statistics.median(tot),
        "depth_upload": "host uint16 image -> device inside the timed raycast (PCIe-inclusive)",
        "cpu_oracle_first_frames": cpu_t, "counters_bit_identical_on_first_frames": bool(cpu_t),
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
