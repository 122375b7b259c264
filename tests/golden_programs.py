"""Deterministic call sequences ("programs") shared by the golden-fixture generator and the tests.

A program drives any map object that exposes the oracle method names (OracleMap) -- or the HIP map through
`GpuAsOracle` -- and yields named checkpoints.  tests/golden/make_golden.py ran every program on the
reference's own sources compiled verbatim (oracle/_ref) and stored what the reference produced at each
checkpoint; the tests replay the same program on the restatement (CPU) and on the HIP engine (GPU).
"""
from __future__ import annotations

import numpy as np

from scenarios import P_DEFAULT, all_voxels, depth_to_points, render_depth, yaw_pose

SMALL_INTR = dict(fx=48.0, fy=48.0, cx=40.0, cy=30.0)


def _cycles(m, occ_vox, free_vox, n, log):
    for _ in range(n):
        if len(occ_vox):
            m.SetOccupancyVox(occ_vox, 1)
        if len(free_vox):
            m.SetOccupancyVox(free_vox, 0)
        log.append((bool(m.UpdateOccupancy(True)), int(m.last_insert), int(m.last_delete)))


def prog_dense_scatter(make):
    """24^3 fully observed: insert 150, mixed insert/delete, delete all (config-1 shape, scaled down)."""
    n, res = 24, 0.1
    m = make((0, 0, 0), res, (n * res,) * 3)
    m.SetParameters(*P_DEFAULT)
    m.SetOriginalRange()
    log = []
    _cycles(m, [], all_voxels(m.grid_size), 1, log)
    st = m.UpdateESDF()
    rng = np.random.RandomState(12345)
    S = rng.randint(0, n, (150, 3)).astype(np.int32)
    _cycles(m, S, [], 3, log)
    st = m.UpdateESDF()
    yield "insert", m, dict(queues=np.array(log), stats=st)
    _cycles(m, rng.randint(0, n, (60, 3)).astype(np.int32), S[:75], 6, log)
    st = m.UpdateESDF()
    yield "mixed", m, dict(queues=np.array(log), stats=st)
    q = 0.15 + rng.rand(500, 3) * (n * res - 0.4)
    yield "queries", m, dict(pos=q)


def prog_dense_pillars(make):
    """The workload documented in the reference's test/test_ESDF_Map.cpp:42-104, at its own size:
    origin (-5,-5,0), size (10,10,5), resolution 0.2; 25 pillars, inserted then half deleted."""
    origin, size, res = (-5.0, -5.0, 0.0), (10.0, 10.0, 5.0), 0.2
    m = make(origin, res, size)
    m.SetParameters(*P_DEFAULT)
    m.SetOriginalRange()
    log = []
    _cycles(m, [], all_voxels(m.grid_size), 1, log)
    m.UpdateESDF()
    pillars = [(x, y) for x in (-4, -2, 0, 2, 4) for y in (-4, -2, 0, 2, 4)]
    order = np.random.RandomState(0).permutation(len(pillars))
    zs = np.arange(0, 5, 0.1)

    def pillar(k):
        return np.stack([np.full_like(zs, pillars[k][0] + 0.01), np.full_like(zs, pillars[k][1] + 0.01), zs + 0.01], -1)
    for k in order:
        for _ in range(3):
            m.SetOccupancyPos(pillar(k), 1)
            m.UpdateOccupancy(True)
        st = m.UpdateESDF()
    yield "inserted", m, dict(stats=st)
    for k in order[:13]:
        for _ in range(6):
            m.SetOccupancyPos(pillar(k), 0)
            m.UpdateOccupancy(True)
        st = m.UpdateESDF()
    yield "deleted", m, dict(stats=st)


def prog_dense_walls_ragged(make):
    """Non-cubic grid with ceil() rounding (4.8/0.1 -> 49) and wall-like obstacles (many distance ties)."""
    res = 0.1
    m = make((-1.0, 2.0, 0.5), res, (2.4, 1.7, 3.5))
    m.SetParameters(*P_DEFAULT)
    m.SetOriginalRange()
    gs = m.grid_size
    log = []
    _cycles(m, [], all_voxels(gs), 1, log)
    m.UpdateESDF()
    ys, zs = np.meshgrid(np.arange(gs[1]), np.arange(gs[2]), indexing="ij")
    wall = np.stack([np.full(ys.size, 10), ys.ravel(), zs.ravel()], -1).astype(np.int32)
    xs, ys2 = np.meshgrid(np.arange(gs[0]), np.arange(gs[1]), indexing="ij")
    floor_ = np.stack([xs.ravel(), ys2.ravel(), np.full(xs.size, 3)], -1).astype(np.int32)
    _cycles(m, np.concatenate([wall, floor_]), [], 3, log)
    st = m.UpdateESDF()
    yield "walls", m, dict(stats=st)
    _cycles(m, [], wall, 6, log)
    st = m.UpdateESDF()
    yield "wall_removed", m, dict(stats=st)


def prog_raycast_frames(make):
    """Three 80x60 synthetic depth frames of a box room with two spheres through RaycastProcess."""
    res = 0.1
    origin, size = (-4.0, -4.0, -2.0), (8.0, 8.0, 4.0)
    m = make(origin, res, size)
    m.SetParameters(*P_DEFAULT)
    m.SetOriginalRange()
    spheres = [((1.5, 0.5, 0.0), 0.6), ((-1.0, -1.5, -0.3), 0.5)]
    for f in range(3):
        T = yaw_pose(20.0 * f, (0.1 * f, -0.05 * f, 0.02))
        pts = depth_to_points(render_depth(T, rows=60, cols=80, spheres=spheres, intr=SMALL_INTR), intr=SMALL_INTR)
        m.raycast_frame(pts, T, T[:3, 3], 0.5, 5.0, origin, np.add(origin, size))
        hit, miss = m.dump_counts()
        m.UpdateOccupancy(True)
        st = m.UpdateESDF()
        yield f"frame{f}", m, dict(num_hit=hit.astype(np.int16), num_miss=miss.astype(np.int16), stats=st,
                                    queues=np.array([[m.last_insert, m.last_delete]]))


PROGRAMS = {
    "dense_scatter": prog_dense_scatter,
    "dense_pillars": prog_dense_pillars,
    "dense_walls_ragged": prog_dense_walls_ragged,
    "raycast_frames": prog_raycast_frames,
}


def golden_rays():
    """Inputs of the Raycast known-answer vectors (voxel units; src/raycast.cpp:56-158)."""
    rng = np.random.RandomState(42)
    lo, hi = np.array([-20.0, -20.0, -5.0]), np.array([20.0, 20.0, 5.0])
    rays = []
    for k in range(120):
        a = rng.uniform(-25, 25, 3) * [1, 1, 0.2]
        b = a + rng.uniform(-30, 30, 3) * [1, 1, 0.2]
        if k % 7 == 0:
            b[rng.randint(3)] = a[rng.randint(3)]
        if k % 11 == 0:
            a = np.round(a)
        rays.append((a, b))
    return rays, lo, hi
