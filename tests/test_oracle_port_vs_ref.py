"""Pins the CPU restatement (oracle/esdf_port.cpp) against the reference's own sources compiled verbatim
(oracle/_ref, built by oracle/Makefile from /root/reference).  Runs only where oracle/_ref exists (this
container, or a GPU box that received the prebuilt binaries); the committed fixtures under tests/golden/
carry the same pin to machines that have neither (tests/test_oracle_golden.py).

The restatement follows the reference's FIFO order literally, so -- unlike the GPU engine -- it must match
the reference bit for bit INCLUDING closest-obstacle ids, log-odds, counters and the expansion counters the
reference prints (src/ESDFMap.cpp:277,394).
"""
import numpy as np
import pytest

from scenarios import P_DEFAULT, all_voxels, depth_to_points, render_depth, yaw_pose, INTRINSICS


@pytest.fixture(scope="module")
def pair_factory(oracle_libs):
    if not oracle_libs.available("ref", "array"):
        pytest.skip("oracle/_ref is not built on this machine (needs /root/reference)")

    def make(origin, res, size=None, mode="array", reserve=0):
        if mode == "hash" and not oracle_libs.available("ref", "hash"):
            pytest.skip("hash flavour of the reference is not built")
        ms = [oracle_libs.OracleMap(origin, res, size, reserve_size=reserve, mode=mode, kind=k)
              for k in ("ref", "port")]
        for m in ms:
            m.SetParameters(*P_DEFAULT)
            m.SetOriginalRange()
        return ms
    return make


def both(ms, fn):
    return [fn(m) for m in ms]


def same_dense(ref, port):
    a, b = ref.dump_dense(), port.dump_dense()
    for k in ("dist", "coc", "occ", "logodds"):
        assert np.array_equal(a[k], b[k]), f"{k} differs between the verbatim reference and the restatement"
    ha, hb = ref.dump_counts(), port.dump_counts()
    assert np.array_equal(ha[0], hb[0]) and np.array_equal(ha[1], hb[1])


def esdf_same(ms):
    sa, sb = both(ms, lambda m: m.UpdateESDF())
    for k in ("inserted", "deleted", "expanded", "change_num"):
        assert sa[k] == sb[k], (k, sa, sb)
    assert ms[0].CheckConsistency() and ms[1].CheckConsistency()
    return sa


def cycles(ms, occ_vox, free_vox, n):
    for _ in range(n):
        for m in ms:
            if len(occ_vox):
                m.SetOccupancyVox(occ_vox, 1)
            if len(free_vox):
                m.SetOccupancyVox(free_vox, 0)
        r = both(ms, lambda m: (m.UpdateOccupancy(True), m.last_insert, m.last_delete))
        assert r[0] == r[1]


def test_dense_insert_delete_mixed(pair_factory):
    n, res = 40, 0.1
    ms = pair_factory((0, 0, 0), res, (n * res,) * 3)
    assert ms[0].grid_size == ms[1].grid_size
    g = all_voxels(ms[0].grid_size)
    cycles(ms, [], g, 1)
    esdf_same(ms)
    rng = np.random.RandomState(1)
    S = rng.randint(0, n, (400, 3)).astype(np.int32)
    cycles(ms, S, [], 3)
    st = esdf_same(ms)
    assert st["inserted"] > 0
    same_dense(*ms)
    cycles(ms, rng.randint(0, n, (150, 3)).astype(np.int32), S[:200], 6)
    st = esdf_same(ms)
    assert st["deleted"] > 0 and st["inserted"] > 0
    same_dense(*ms)
    occ = np.argwhere(ms[0].dump_dense(("occ",))["occ"].reshape(ms[0].grid_size) == 1).astype(np.int32)
    cycles(ms, [], occ, 6)
    esdf_same(ms)
    same_dense(*ms)


def test_dense_partial_observation_positions_and_window(pair_factory):
    res = 0.25
    ms = pair_factory((-3.0, -3.0, -1.0), res, (6.1, 5.9, 3.3))  # ceil() rounding
    assert ms[0].grid_size == ms[1].grid_size
    rng = np.random.RandomState(5)
    for cyc in range(6):
        pos = np.array([-3.0, -3.0, -1.0]) + (rng.rand(4000, 3) * 1.2 - 0.1) * np.array([6.1, 5.9, 3.3])
        occ = (rng.rand(4000) < 0.45).astype(np.int32)
        occ[::97] = 2  # "occ value error!" (src/ESDFMap.cpp:402-405)
        if cyc == 3:
            for m in ms:
                m.SetUpdateRange((-1.0, -1.5, -0.5), (2.0, 2.2, 1.7))
        if cyc == 5:
            for m in ms:
                m.SetOriginalRange()
        ra, rb = both(ms, lambda m: m.SetOccupancyPos(pos, occ))
        assert np.array_equal(ra, rb)
        assert ms[0].CheckUpdate() == ms[1].CheckUpdate()
        r = both(ms, lambda m: (m.UpdateOccupancy(cyc % 2 == 0), m.last_insert, m.last_delete))
        assert r[0] == r[1]
        esdf_same(ms)
        same_dense(*ms)
    q = np.array([-3.0, -3.0, -1.0]) + rng.rand(2000, 3) * np.array([5.0, 5.0, 2.5]) + 0.3
    for name in ("GetDistancePos", "GetOccupancyPos"):
        a, b = both(ms, lambda m: getattr(m, name)(q))
        assert np.array_equal(a, b)
    (da, ga), (db, gb) = both(ms, lambda m: m.GetDistWithGradTrilinear(q))
    assert np.array_equal(da, db) and np.array_equal(ga, gb)
    out = np.array([[100.0, 0, 0], [-50.0, 1, 1]])
    a, b = both(ms, lambda m: m.GetDistWithGradTrilinear(out)[0])
    assert np.array_equal(a, b) and np.all(a == -1)


def test_hash_flavour(pair_factory):
    res = 0.1
    ms = pair_factory((0, 0, 0), res, mode="hash", reserve=20000)
    n = 32
    g = all_voxels(n) - 5  # negative block ids included (arithmetic shift, include/ESDFMap.h:22-35)
    cycles(ms, [], g, 1)
    esdf_same(ms)
    rng = np.random.RandomState(9)
    S = (rng.randint(0, n, (200, 3)) - 5).astype(np.int32)
    cycles(ms, S, [], 3)
    esdf_same(ms)
    cycles(ms, (rng.randint(0, n, (60, 3)) - 5).astype(np.int32), S[:100], 6)
    esdf_same(ms)
    a, b = ms[0].dump_hash(), ms[1].dump_hash()
    assert ms[0].grid_total_size == ms[1].grid_total_size
    for k in ("vox", "dist", "coc", "occ"):
        assert np.array_equal(a[k], b[k]), k


def _vis_same(ms, slices, bounds):
    for lo, hi in bounds:
        a, b = both(ms, lambda m: m.GetPointCloud(lo, hi))
        assert np.array_equal(a, b)            # message order included
    for z in slices:
        (pa, ca), (pb, cb) = both(ms, lambda m: m.GetSliceMarker(z, 1.3))
        assert np.array_equal(pa, pb) and np.array_equal(ca, cb)
    return len(a), len(pa)


@pytest.mark.parametrize("mode", ["array", "hash"])
def test_visualisation_getters(pair_factory, mode):
    """GetPointCloud / GetSliceMarker (src/ESDFMap.cpp:544-699): points, colours and message order of the restatement
    against the reference's own functions (compiled against the message stand-ins of oracle/shim)."""
    n, res = 24, 0.2
    ms = pair_factory((-1.0, 0.5, 0.0), res, (n * res,) * 3, mode=mode, reserve=20000)
    cycles(ms, [], all_voxels(n), 1)
    esdf_same(ms)
    S = np.random.RandomState(3).randint(2, n - 2, (90, 3)).astype(np.int32)
    cycles(ms, S, [], 3)
    esdf_same(ms)
    npc, nsl = _vis_same(ms, (3, 11), ((0, n), (5, 9), (7, 7), (30, 40)))
    assert nsl > 0
    for m in ms:
        m.SetUpdateRange((0.0, 1.0, 0.4), (2.5, 3.1, 3.0))
    npc, nsl = _vis_same(ms, (3, 11), ((0, n), (5, 9)))
    assert 0 < nsl < n * n


def test_raycast_free_function(oracle_libs, pair_factory):
    rng = np.random.RandomState(0)
    lo, hi = np.array([-20.0, -20.0, -5.0]), np.array([20.0, 20.0, 5.0])
    for k in range(400):
        a = rng.uniform(-25, 25, 3) * [1, 1, 0.2]
        b = a + rng.uniform(-30, 30, 3) * [1, 1, 0.2]
        if k % 7 == 0:
            b[rng.randint(3)] = a[rng.randint(3)]
        if k % 11 == 0:
            a = np.round(a)
        assert np.array_equal(oracle_libs.raycast(a, b, lo, hi, kind="ref"), oracle_libs.raycast(a, b, lo, hi, kind="port"))
    a, b = np.array([0.5, 0.5, 0.5]), np.array([1900.5, 3.5, 0.5])
    for kind in ("ref", "port"):
        with pytest.raises(IndexError):  # std::out_of_range past 1500 voxels (src/raycast.cpp:127-130)
            oracle_libs.raycast(a, b, [-1e4] * 3, [1e4] * 3, kind=kind)


def test_raycast_frames_depth(pair_factory):
    """RaycastProcess is restated in BOTH oracles (the header is ROS-entangled); what differs is the Raycast
    and SetOccupancy underneath: verbatim reference vs restatement."""
    res = 0.1
    origin, size = (-4.0, -4.0, -2.0), (8.0, 8.0, 4.0)
    ms = pair_factory(origin, res, size)
    spheres = [((1.5, 0.5, 0.0), 0.6), ((-1.0, -1.5, -0.3), 0.5)]
    for f in range(3):
        T = yaw_pose(20.0 * f, (0.1 * f, -0.05 * f, 0.02))
        pts = depth_to_points(render_depth(T, rows=60, cols=80, spheres=spheres,
                                           intr=dict(fx=48.0, fy=48.0, cx=40.0, cy=30.0)),
                              intr=dict(fx=48.0, fy=48.0, cx=40.0, cy=30.0))
        for m in ms:
            m.raycast_frame(pts, T, T[:3, 3], 0.5, 5.0, origin, np.add(origin, size))
        ha, hb = ms[0].dump_counts(), ms[1].dump_counts()
        assert np.array_equal(ha[0], hb[0]) and np.array_equal(ha[1], hb[1])
        r = both(ms, lambda m: (m.UpdateOccupancy(True), m.last_insert, m.last_delete))
        assert r[0] == r[1]
        esdf_same(ms)
    same_dense(*ms)
