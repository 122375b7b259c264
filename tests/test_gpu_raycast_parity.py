"""GPU parity of the ray-cast front end against the CPU oracle.

The oracle runs Fiesta::RaycastProcess single-threaded in cloud order (the verbatim reference Raycast +
a literal restatement of the driver). The HIP path must reproduce the per-voxel hit/miss counters of every
frame EXACTLY, including the reference's order-dependent per-frame de-duplication.
"""
import numpy as np
import pytest

from scenarios import (INTRINSICS, P_DEFAULT, EnvelopeOracle, assert_envelope, assert_exact, compare_dense, d2_from_dist,
                       depth_to_points, render_depth, yaw_pose)

pytestmark = pytest.mark.gpu

RAY = dict(min_ray_length=0.5, max_ray_length=5.0)


def make(oracle_libs, kind, origin, size, res, envelope=0):
    """envelope = K: the oracle side is the reference plus K shuffled-order replays of it (scenarios.EnvelopeOracle)."""
    import fiesta_amd
    gpu = fiesta_amd.ESDFMap(origin, res, size)
    mk = lambda: oracle_libs.OracleMap(origin, res, size, kind=kind)   # noqa: E731
    cpu = EnvelopeOracle(mk, k=envelope) if envelope else mk()
    assert gpu.grid_size == cpu.grid_size
    for m in (gpu, cpu):
        m.SetParameters(*P_DEFAULT)
        m.SetOriginalRange()
    return gpu, cpu


def test_single_ray_traversal_bit_exact(hip_lib, oracle_libs, best_oracle_kind):
    import ctypes as C
    rng = np.random.RandomState(0)
    lib = hip_lib
    lo, hi = np.array([-20.0, -20.0, -5.0]), np.array([20.0, 20.0, 5.0])
    for k in range(300):
        a = rng.uniform(-25, 25, 3) * [1, 1, 0.2]
        b = a + rng.uniform(-30, 30, 3) * [1, 1, 0.2]
        if k % 7 == 0:
            b[rng.randint(3)] = a[rng.randint(3)]          # axis-aligned / degenerate deltas
        if k % 11 == 0:
            a = np.round(a)                                 # start exactly on voxel boundaries
        want = oracle_libs.raycast(a, b, lo, hi, kind=best_oracle_kind)
        out = np.empty((2048, 3))
        n = C.c_int32(0)
        st = lib.fiesta_hip_raycast_single(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                                           lo.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p),
                                           out.ctypes.data_as(C.c_void_p), 2048, C.byref(n), 0)
        assert st == 0, lib.fiesta_hip_last_error()
        assert n.value == len(want)
        assert np.array_equal(out[: n.value], want)
    # the reference throws past 1500 voxels; the ABI reports an error instead of throwing
    a, b = np.array([0.5, 0.5, 0.5]), np.array([1900.5, 3.5, 0.5])
    big_lo, big_hi = np.array([-1e4] * 3), np.array([1e4] * 3)
    with pytest.raises(IndexError):
        oracle_libs.raycast(a, b, big_lo, big_hi, kind=best_oracle_kind)
    n = C.c_int32(0)
    out = np.empty((4, 3))
    st = lib.fiesta_hip_raycast_single(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                                       big_lo.ctypes.data_as(C.c_void_p), big_hi.ctypes.data_as(C.c_void_p),
                                       out.ctypes.data_as(C.c_void_p), 4, C.byref(n), 0)
    assert st != 0


def check_counts(gpu, cpu):
    gh, gm = gpu.download_counts()
    ch, cm = cpu.dump_counts()
    assert np.array_equal(gm, cm), f"observation counters differ at {np.flatnonzero(gm != cm)[:10]}"
    assert np.array_equal(gh, ch), "hit counters differ"
    return int((gm > 0).sum())


def test_frames_counts_exact_with_reference_dedup(hip_lib, oracle_libs, best_oracle_kind):
    """Yaw sweep in a box room with spheres: per-frame counters, fusion, queues and ESDF vs the oracle."""
    origin, size, res = (-6.4, -6.4, -3.2), (12.75, 12.75, 6.35), 0.1
    gpu, cpu = make(oracle_libs, best_oracle_kind, origin, size, res, envelope=5)
    assert gpu.grid_size == (128, 128, 64)
    lc, rc = origin, tuple(np.array(origin) + np.array(size))
    spheres = [((1.5, 0.5, 0.0), 0.5), ((-1.0, 2.0, 0.3), 0.7), ((0.5, -2.0, -0.5), 0.4)]
    pos = np.array([0.13, -0.21, 0.05])
    touched_total = 0
    for f in range(6):
        T = yaw_pose(20.0 * f, pos + 0.05 * f)
        depth = render_depth(T, rows=120, cols=160, spheres=spheres,
                             intr=dict(fx=96.1, fy=96.1, cx=80.7, cy=58.9))
        pts = depth_to_points(depth, intr=dict(fx=96.1, fy=96.1, cx=80.7, cy=58.9))
        pts[::501] = np.nan                      # invalid points are skipped (include/Fiesta.h:202)
        o = T[:3, 3]
        gpu.RaycastFrame(pts, T, o, RAY["min_ray_length"], RAY["max_ray_length"], lc, rc, dedup=1)
        cpu.raycast_frame(pts, T, o, RAY["min_ray_length"], RAY["max_ray_length"], lc, rc)
        touched_total += check_counts(gpu, cpu)
        assert gpu.CheckUpdate() == cpu.CheckUpdate()
        a, b = gpu.UpdateOccupancy(True), cpu.UpdateOccupancy(True)
        assert a == b and (gpu.last_insert, gpu.last_delete) == (cpu.last_insert, cpu.last_delete)
        gpu.UpdateESDF()
        cpu.UpdateESDF()
        rep = compare_dense(gpu, cpu)
        # partially observed map: the reference itself is order-dependent here -> judged against the
        # envelope of its own shuffled runs on these very frames
        assert_envelope(rep, f"frame {f}", strict=gpu.only_levels)
        assert rep["pair_violations"] == 0, rep
    assert touched_total > 30000
    assert gpu.download_field(("occ",))["occ"].sum() > 500


def test_long_rays_in_random_cloud_order_counts_exact(hip_lib, oracle_libs, best_oracle_kind):
    """Walks of several hundred voxels (a wave covers 64 entries of a casting ray at a time: the chunked truncation, the
    packed coordinate scan and its carries), a cloud in random order (the de-duplication depends on it), a ray box
    smaller than the map, end points beyond the map (those rays always cast), beyond max range (clipped, counted free)
    and closer than min range, runs of equal end points (one counter update per run): counters equal the oracle's."""
    origin, size, res = (-9.6, -9.6, -1.2), (19.15, 19.15, 2.35), 0.05
    gpu, cpu = make(oracle_libs, best_oracle_kind, origin, size, res)
    assert gpu.grid_size == (383, 383, 47)
    lc, rc = (-9.0, -8.0, -1.0), (8.5, 9.3, 1.1)          # rays are clipped to this box (l_cornor / r_cornor)
    rng = np.random.RandomState(5)
    for f in range(3):
        o = np.array([0.37, -0.22, 0.11]) + 0.4 * f
        T = np.eye(4)
        T[:3, 3] = o
        n = 3000
        d = rng.normal(size=(n, 3)) * [1.0, 1.0, 0.06]
        d /= np.linalg.norm(d, axis=1)[:, None]
        r = rng.uniform(0.3, 14.0, n)                       # some closer than min range, some beyond max range and the map
        pts = (d * r[:, None]).astype(np.float32)
        pts[::7] = pts[1::7][: len(pts[::7])]              # duplicates: several points per end voxel ...
        pts[100:160] = pts[100]                             # ... and a run of equal end points inside one wave
        pts[::333] = np.nan
        gpu.RaycastFrame(pts, T, o, 1.0, 12.0, lc, rc, dedup=1)
        cpu.raycast_frame(pts, T, o, 1.0, 12.0, lc, rc)
        assert check_counts(gpu, cpu) > 20000
        assert gpu.UpdateOccupancy(True) == cpu.UpdateOccupancy(True)
        assert (gpu.last_insert, gpu.last_delete) == (cpu.last_insert, cpu.last_delete)


def test_depth_image_entry_point_matches_point_path(hip_lib, oracle_libs, best_oracle_kind):
    """fiesta_hip_raycast_depth (device-side pinhole conversion) == host conversion + raycast_frame == oracle."""
    origin, size, res = (-6.4, -6.4, -3.2), (12.75, 12.75, 6.35), 0.1
    gpu, cpu = make(oracle_libs, best_oracle_kind, origin, size, res)
    lc, rc = origin, tuple(np.array(origin) + np.array(size))
    intr = dict(fx=96.1, fy=96.1, cx=80.7, cy=58.9)
    T = yaw_pose(33.0, (0.2, 0.1, -0.3))
    depth = render_depth(T, rows=120, cols=160, intr=intr, spheres=[((2.0, 1.0, 0.0), 0.6)])
    depth[5:9, 7:30] = 0                          # holes read as zero depth -> shorter than min range
    gpu.RaycastDepth(depth, intr["fx"], intr["fy"], intr["cx"], intr["cy"], T, T[:3, 3], 0.5, 5.0, lc, rc, dedup=1)
    cpu.raycast_frame(depth_to_points(depth, intr), T, T[:3, 3], 0.5, 5.0, lc, rc)
    assert check_counts(gpu, cpu) > 3000


def test_no_dedup_mode_counts_every_crossing(hip_lib, oracle_libs, best_oracle_kind):
    """dedup=0 has no reference counterpart: every valid ray counts its end point and every voxel it crosses.
    Property: counters dominate the de-duplicated ones and the set of touched voxels is a superset."""
    origin, size, res = (-3.2, -3.2, -1.6), (6.35, 6.35, 3.15), 0.1
    g0, _ = make(oracle_libs, best_oracle_kind, origin, size, res)
    g1, _ = make(oracle_libs, best_oracle_kind, origin, size, res)
    lc, rc = origin, tuple(np.array(origin) + np.array(size))
    T = yaw_pose(10.0, (0.0, 0.0, 0.0))
    intr = dict(fx=48.0, fy=48.0, cx=40.0, cy=30.0)
    depth = render_depth(T, rows=60, cols=80, intr=intr, room=((-2.5, -2.5, -1.2), (2.5, 2.5, 1.2)))
    pts = depth_to_points(depth, intr)
    g0.RaycastFrame(pts, T, T[:3, 3], 0.5, 5.0, lc, rc, dedup=0)
    g1.RaycastFrame(pts, T, T[:3, 3], 0.5, 5.0, lc, rc, dedup=1)
    h0, m0 = g0.download_counts()
    h1, m1 = g1.download_counts()
    assert np.all(m0 >= m1) and np.all(h0 >= h1)
    assert m0.sum() > m1.sum()
    assert np.all((m1 > 0) <= (m0 > 0))


@pytest.mark.parametrize("site", [(0.0, 0.0, 0.0), (83.0, -71.0, 26.0)], ids=["at-origin", "800-voxels-away"])
def test_hash_map_frames_counts_exact(hip_lib, oracle_libs, best_oracle_kind, site):
    """The same front end on the paged (hash-block) map against the reference built with -DHASH_TABLE: pages are
    allocated by the rays themselves; per-voxel hit/miss counters of every frame, queues and the ESDF must match voxel
    by voxel (internal slots differ by design, so everything is keyed by voxel coordinates).  The second site is outside
    the map's initial window: the first frame has to move the window there (the reference's hash map is unbounded)."""
    import fiesta_amd
    from test_gpu_hash_parity import compare as compare_hash
    kind = best_oracle_kind if oracle_libs.available(best_oracle_kind, "hash") else "port"
    origin, res = (0.3, -0.2, 0.1), 0.1
    gpu = fiesta_amd.ESDFMap(origin, res, reserve_size=1000, mode="hash")
    cpu = EnvelopeOracle(lambda: oracle_libs.OracleMap(origin, res, reserve_size=1000, mode="hash", kind=kind), k=4)
    for m in (gpu, cpu):
        m.SetParameters(*P_DEFAULT)
        m.SetOriginalRange()
    site = np.array(site)
    lc, rc = tuple(site - 20.0), tuple(site + 20.0)      # the hash build's l_cornor/r_cornor only clip the walk
    spheres = [(tuple(site + c), r) for c, r in (((1.5, 0.5, 0.0), 0.5), ((-1.0, 2.0, 0.3), 0.7), ((0.5, -2.0, -0.5), 0.4))]
    room = (tuple(site + (-3.0, -3.0, -1.5)), tuple(site + (3.0, 3.0, 1.5)))
    intr = dict(fx=96.1, fy=96.1, cx=80.7, cy=58.9)
    pos = site + np.array([0.13, -0.21, 0.05])
    key = lambda v: (v[:, 0].astype(np.int64) + 100000) * (1 << 40) + (v[:, 1].astype(np.int64) + 100000) * (1 << 20) + v[:, 2] + 100000  # noqa: E731
    touched_total = 0
    for f in range(5):
        T = yaw_pose(25.0 * f, pos + 0.07 * f)
        depth = render_depth(T, rows=120, cols=160, room=room, spheres=spheres, intr=intr)
        pts = depth_to_points(depth, intr=intr)
        pts[::397] = np.nan
        o = T[:3, 3]
        if f == 3:   # the depth-image entry point on the paged map
            gpu.RaycastDepth(depth, intr["fx"], intr["fy"], intr["cx"], intr["cy"], T, o, 0.5, 5.0, lc, rc, dedup=1)
            cpu.raycast_frame(depth_to_points(depth, intr=intr), T, o, 0.5, 5.0, lc, rc)
        else:
            gpu.RaycastFrame(pts, T, o, 0.5, 5.0, lc, rc, dedup=1)
            cpu.raycast_frame(pts, T, o, 0.5, 5.0, lc, rc)
        # counters, voxel by voxel
        gh, gm = gpu.download_counts()
        gv = gpu.download_hash()["vox"]
        ch, cm = cpu.dump_counts()
        cv = cpu.dump_hash()["vox"]
        gk, ck = key(gv), key(cv)
        gsel, csel = gm > 0, cm > 0
        og, oc = np.argsort(gk[gsel]), np.argsort(ck[csel])
        assert np.array_equal(gk[gsel][og], ck[csel][oc]), "different sets of observed voxels"
        assert np.array_equal(gm[gsel][og], cm[csel][oc]) and np.array_equal(gh[gsel][og], ch[csel][oc])
        touched_total += int(gsel.sum())
        assert gpu.CheckUpdate() == cpu.CheckUpdate()
        a, b = gpu.UpdateOccupancy(True), cpu.UpdateOccupancy(True)
        assert a == b and (gpu.last_insert, gpu.last_delete) == (cpu.last_insert, cpu.last_delete)
        sg, sc = gpu.UpdateESDF(), cpu.UpdateESDF()
        assert (sg["inserted"], sg["deleted"]) == (sc["inserted"], sc["deleted"])
        rep = compare_hash(gpu, cpu)
        assert_envelope(rep, f"frame {f}", strict=gpu.only_levels)
    assert touched_total > 30000 and rep["pages"] >= 8
    assert gpu.hash_window()[1] == (1 if site.any() else 0) and sg["dropped_observations"] == 0


def test_config3_640x480_frames_reference_intrinsics(hip_lib, oracle_libs, best_oracle_kind):
    """BASELINE config 3 at FULL size: the 512^3 @0.1 m map fed by 640 x 480 depth frames (307 200 rays) through the
    device-side depth front end, with the reference's default intrinsics (src/parameters.cpp:21-24), ray window
    0.5-5.0 m and the reference's per-frame de-duplication -- the very frames `bench.py --workload c3` times.  Per-voxel
    hit/miss counters of every frame, UpdateOccupancy's queue sizes and UpdateESDF's counters must equal the oracle's
    (verbatim Raycast + restated RaycastProcess, single thread, cloud order)."""
    G, res = 512, 0.1
    half = G * res / 2
    origin, size = (-half, -half, -half), (G * res,) * 3
    gpu, cpu = make(oracle_libs, best_oracle_kind, origin, size, res)
    assert gpu.grid_size == (G, G, G)
    lc, rc = origin, tuple(np.add(origin, size))
    spheres = [((1.5, 0.5, 0.0), 0.5), ((-1.0, 2.0, 0.3), 0.7), ((0.5, -2.0, -0.5), 0.4), ((-2.0, -1.0, 0.5), 0.6),
               ((2.2, -1.8, 0.2), 0.3)]
    assert abs(INTRINSICS["fx"] - 384.458089392) < 1e-12 and abs(INTRINSICS["cy"] - 237.076346481) < 1e-12
    touched = 0
    for f in range(3):
        T = yaw_pose(2.0 * f, (0.0, 0.0, 0.0))
        depth = render_depth(T, rows=480, cols=640, spheres=spheres, intr=INTRINSICS)
        gpu.RaycastDepth(depth, INTRINSICS["fx"], INTRINSICS["fy"], INTRINSICS["cx"], INTRINSICS["cy"], T, T[:3, 3],
                         RAY["min_ray_length"], RAY["max_ray_length"], lc, rc, dedup=1)
        cpu.raycast_frame(depth_to_points(depth, INTRINSICS), T, T[:3, 3], RAY["min_ray_length"], RAY["max_ray_length"], lc, rc)
        touched += check_counts(gpu, cpu)
        a, b = gpu.UpdateOccupancy(True), cpu.UpdateOccupancy(True)
        assert a == b and (gpu.last_insert, gpu.last_delete) == (cpu.last_insert, cpu.last_delete)
        sg, sc = gpu.UpdateESDF(), cpu.UpdateESDF()
        assert (sg["inserted"], sg["deleted"]) == (sc["inserted"], sc["deleted"])
    assert touched > 25000   # (~10 k distinct voxels per frame after the per-frame de-duplication)
    # occupancy after three frames, voxel for voxel (the distance field of this partially observed map is covered, with
    # its stated budget, by the smaller frame tests above)
    assert np.array_equal(gpu.download_field(("occ",))["occ"], cpu.dump_dense(("occ",))["occ"])


def test_config3_full_size_distances_inside_the_reference_envelope(hip_lib, oracle_libs, best_oracle_kind):
    """VERDICT r4 missing #4: the DISTANCE FIELD of config 3 at full size.  The same three 640 x 480 frames into the 512^3 map,
    the level engine pinned (the reference's FIFO layers as levels: the engine whose contract has no constant), and next to
    the reference run three more runs of the verbatim reference fed the same observations in shuffled first-touch order
    (scenarios.EnvelopeOracle).  After the third frame every observed voxel's squared distance is judged against the
    interval those four runs span: outside it on at most as many voxels as the runs themselves disagree on, either side."""
    import fiesta_amd
    import psutil
    if psutil.virtual_memory().available < 48 * 2 ** 30:
        pytest.skip("four 512^3 reference maps (6.4 GB each) need ~40 GB of host memory")
    G, res = 512, 0.1
    half = G * res / 2
    origin, size = (-half, -half, -half), (G * res,) * 3
    gpu = fiesta_amd.ESDFMap(origin, res, size, update_engine="levels")
    cpu = EnvelopeOracle(lambda: oracle_libs.OracleMap(origin, res, size, kind=best_oracle_kind), k=3)
    for m in (gpu, cpu):
        m.SetParameters(*P_DEFAULT)
        m.SetOriginalRange()
    lc, rc = origin, tuple(np.add(origin, size))
    spheres = [((1.5, 0.5, 0.0), 0.5), ((-1.0, 2.0, 0.3), 0.7), ((0.5, -2.0, -0.5), 0.4), ((-2.0, -1.0, 0.5), 0.6),
               ((2.2, -1.8, 0.2), 0.3)]
    for f in range(3):
        T = yaw_pose(2.0 * f, (0.0, 0.0, 0.0))
        depth = render_depth(T, rows=480, cols=640, spheres=spheres, intr=INTRINSICS)
        gpu.RaycastDepth(depth, INTRINSICS["fx"], INTRINSICS["fy"], INTRINSICS["cx"], INTRINSICS["cy"], T, T[:3, 3],
                         RAY["min_ray_length"], RAY["max_ray_length"], lc, rc, dedup=1)
        cpu.raycast_frame(depth_to_points(depth, INTRINSICS), T, T[:3, 3], RAY["min_ray_length"], RAY["max_ray_length"], lc, rc)
        a, b = gpu.UpdateOccupancy(True), cpu.UpdateOccupancy(True)
        assert a == b and (gpu.last_insert, gpu.last_delete) == (cpu.last_insert, cpu.last_delete)
        sg, sc = gpu.UpdateESDF(), cpu.UpdateESDF()
        assert (sg["inserted"], sg["deleted"]) == (sc["inserted"], sc["deleted"])
    gd2 = gpu.download_field(("d2",))["d2"].astype(np.int64)
    observed = np.flatnonzero(gd2 >= 0)
    assert 10 ** 4 < len(observed) < 10 ** 7
    env = cpu.judge(gd2, mask=observed)          # (also checks nothing: the observed sets are compared next)
    pd2 = d2_from_dist(cpu.dump_dense(("dist",))["dist"], res)
    assert np.array_equal(gd2 < 0, pd2 < 0), "observed sets differ"
    assert env["finite"] > 10000
    assert gpu.only_levels, gpu.served
    assert_envelope(env, "config 3, 512^3, after three 640 x 480 frames", strict=True)


def test_temporal_depth_filter_three_frames(hip_lib, oracle_libs, best_oracle_kind):
    """Fiesta::DepthConversion with use_depth_filter_ (include/Fiesta.h:352-379; the reference's default): the previous
    depth image and the relative pose decide which pixels may cast a ray.  A moving sensor over 4 frames (the first casts
    nothing, like upstream's image_cnt_ == 1) with a sphere that APPEARS in frame 2 (its pixels disagree with the previous
    image and are rejected once) and a non-zero margin: (1) the surviving points equal the oracle's cloud, point for point
    and in order; (2) the ray cast of the filtered frame leaves bit-identical hit/miss counters."""
    import fiesta_amd
    origin, size, res = (-6.4, -6.4, -3.2), (12.75, 12.75, 6.35), 0.1
    gpu, cpu = make(oracle_libs, best_oracle_kind, origin, size, res)
    conv = fiesta_amd.ESDFMap(origin, res, size)      # a second map: the conversion-only entry point keeps its own image
    lc, rc = origin, tuple(np.array(origin) + np.array(size))
    intr = dict(fx=96.1, fy=95.9, cx=80.7, cy=58.9)
    flt = dict(tolerance=0.1, max_dist=10.0, min_dist=0.1, margin=3)
    last_img, last_T = None, None
    total = 0
    for f in range(4):
        T = yaw_pose(4.0 * f, (0.05 * f, -0.03 * f, 0.01 * f))
        spheres = [((1.5, 0.5, 0.0), 0.5)] + ([((1.2, -0.6, 0.1), 0.45)] if f >= 2 else [])
        depth = render_depth(T, rows=120, cols=160, spheres=spheres, intr=intr)
        rel = np.eye(4) if last_T is None else np.linalg.inv(last_T) @ T
        want = oracle_libs.depth_conversion(depth, last_img, intr["fx"], intr["fy"], intr["cx"], intr["cy"], rel=rel,
                                            kind=best_oracle_kind, **flt)
        pts, n = conv.DepthConversion(depth, intr["fx"], intr["fy"], intr["cx"], intr["cy"], rel_transform=rel, **flt)
        kept = pts[~np.isnan(pts[:, 0])]
        assert n == len(want) == len(kept), (f, n, len(want))
        assert np.array_equal(kept, want), f"frame {f}: filtered cloud differs from the reference's"
        if f == 0:
            assert n == 0
        elif f == 2:
            assert 0 < n < 0.98 * 114 * 154       # the new sphere's pixels were rejected (plus the margin)
        gpu.RaycastDepthFiltered(depth, intr["fx"], intr["fy"], intr["cx"], intr["cy"], T, T[:3, 3], 0.5, 5.0, lc, rc, rel, **flt)
        cpu.raycast_frame(want, T, T[:3, 3], 0.5, 5.0, lc, rc)
        total += check_counts(gpu, cpu)
        a, b = gpu.UpdateOccupancy(True), cpu.UpdateOccupancy(True)
        assert a == b and (gpu.last_insert, gpu.last_delete) == (cpu.last_insert, cpu.last_delete)
        gpu.UpdateESDF()
        cpu.UpdateESDF()
        last_img, last_T = depth, T
    assert total > 10000
    # a run can be restarted: reset forgets the previous image
    pts, n = conv.DepthConversion(depth, intr["fx"], intr["fy"], intr["cx"], intr["cy"], rel_transform=np.eye(4), reset=True, **flt)
    assert n == 0 and np.isnan(pts).all()


@pytest.mark.parametrize("mode", ["array", "hash"])
def test_signed_variant_inverse_map(hip_lib, oracle_libs, best_oracle_kind, mode):
    """-DSIGNED_NEEDED (include/Fiesta.h:39-41, 92-99, 216-218, 249-251, 515-518): a second map of the same geometry is
    fed every frame inverted -- end points free, crossed voxels occupied -- and updated alongside.  Per frame the
    hit/miss counters of BOTH maps must equal the reference driver's, then queues and fields of the inverse map as well;
    the signed distance (map minus inverse map) is negative exactly inside what the inverse map calls free."""
    import fiesta_amd
    kind = best_oracle_kind if oracle_libs.available(best_oracle_kind, mode) else "port"
    origin, res, size = (-6.4, -6.4, -3.2), 0.1, (12.75, 12.75, 6.35)
    if mode == "hash":
        mk_g = lambda: fiesta_amd.ESDFMap(origin, res, reserve_size=1000, mode="hash")                       # noqa: E731
        mk_c = lambda: oracle_libs.OracleMap(origin, res, reserve_size=1000, mode="hash", kind=kind)        # noqa: E731
    else:
        mk_g = lambda: fiesta_amd.ESDFMap(origin, res, size)                                                  # noqa: E731
        mk_c = lambda: oracle_libs.OracleMap(origin, res, size, kind=kind)                                    # noqa: E731
    g, gi, c = mk_g(), mk_g(), mk_c()
    ci = EnvelopeOracle(mk_c, k=4) if mode == "array" else mk_c()   # (the inverse map's field is judged below)
    for m in (g, gi, c, ci):
        m.SetParameters(*P_DEFAULT)
        m.SetOriginalRange()
    lc, rc = origin, tuple(np.array(origin) + np.array(size))
    spheres = [((1.5, 0.5, 0.0), 0.5), ((-1.0, 2.0, 0.3), 0.7)]
    intr = dict(fx=96.1, fy=96.1, cx=80.7, cy=58.9)
    key = lambda v: (v[:, 0].astype(np.int64) + 100000) * (1 << 40) + (v[:, 1].astype(np.int64) + 100000) * (1 << 20) + v[:, 2] + 100000  # noqa: E731

    def counts(gm, cm):
        gh, gmiss = gm.download_counts()
        ch, cmiss = cm.dump_counts()
        if mode == "array":
            assert np.array_equal(gmiss, cmiss) and np.array_equal(gh, ch)
            return int((gmiss > 0).sum())
        gk, ck = key(gm.download_hash()["vox"]), key(cm.dump_hash()["vox"])
        gs, cs = gmiss > 0, cmiss > 0
        og, oc = np.argsort(gk[gs]), np.argsort(ck[cs])
        assert np.array_equal(gk[gs][og], ck[cs][oc])
        assert np.array_equal(gmiss[gs][og], cmiss[cs][oc]) and np.array_equal(gh[gs][og], ch[cs][oc])
        return int(gs.sum())

    for f in range(4):
        T = yaw_pose(30.0 * f, np.array([0.13, -0.21, 0.05]) + 0.06 * f)
        pts = depth_to_points(render_depth(T, rows=120, cols=160, spheres=spheres, intr=intr), intr=intr)
        o = T[:3, 3]
        g.RaycastFrame(pts, T, o, 0.5, 5.0, lc, rc, dedup=1)
        gi.RaycastFrame(pts, T, o, 0.5, 5.0, lc, rc, dedup=1, inverse=1)
        c.raycast_frame(pts, T, o, 0.5, 5.0, lc, rc, inverse_map=ci)
        assert counts(g, c) > 5000 and counts(gi, ci) > 5000
        for gm, cm in ((g, c), (gi, ci)):
            assert gm.CheckUpdate() == cm.CheckUpdate()
            assert gm.UpdateOccupancy(True) == cm.UpdateOccupancy(True)
            assert (gm.last_insert, gm.last_delete) == (cm.last_insert, cm.last_delete)
            sg, sc = gm.UpdateESDF(), cm.UpdateESDF()
            assert (sg["inserted"], sg["deleted"]) == (sc["inserted"], sc["deleted"])
    # the inverse map: its occupied voxels are the observed-free ones; fields vs the envelope of the reference's own
    # shuffled runs (a partially observed map)
    if mode == "array":
        rep = compare_dense(gi, ci)
        assert_envelope(rep, "inverse map", strict=gi.only_levels)
        assert rep["pair_violations"] == 0, rep
        occ_main, occ_inv = g.download_field(("occ",))["occ"], gi.download_field(("occ",))["occ"]
        assert occ_inv.sum() > occ_main.sum() > 0     # (a grazed voxel can be occupied in both: hit by some rays, crossed by others)
    # signed distance of the pair, here and there
    q = np.array([[1.5, 0.5, 0.0], [1.9, 0.5, 0.0], [0.6, 0.1, 0.0], [-1.0, 2.0, 0.3]])
    d, di = g.GetDistance(q), gi.GetDistance(q)
    assert np.array_equal(d, c.GetDistancePos(q)) and np.array_equal(di, ci.GetDistancePos(q))
    sd = fiesta_amd.signed_distance(g, gi, q)
    ok = (np.abs(d) < 10000) & (np.abs(di) < 10000)
    assert ok[2] and np.array_equal(sd[ok], (d - di)[ok]) and np.all(np.isnan(sd[~ok]))
    assert sd[2] > 0                                          # free space between sensor and sphere: positive
    surf = g.GetOccupiedVoxels() if mode == "array" else g.download_hash()["vox"][g.download_hash()["occ"] == 1]
    ssd = fiesta_amd.signed_distance(g, gi, surf[:200].astype(np.int32))
    assert np.all(ssd[~np.isnan(ssd)] <= 0) and (ssd < 0).sum() > 100          # on observed surfaces: negative
