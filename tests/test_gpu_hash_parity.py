"""GPU parity of the sparse ("hash-block") map (BASELINE config 4) against the CPU oracle built
with -DHASH_TABLE (verbatim reference when oracle/_ref is present, else the restatement).

The reference allocates 8^3 blocks in arrival order -- also for blocks its neighbour READS touch; the GPU map allocates
16x16x32 pages when a voxel in them is observed.  Internal indices and the allocated set therefore differ by design;
what must match, voxel by voxel over the UNION of both allocated sets, is the state: d^2 / occupancy bit-exact, closest
obstacle tie-equivalent, and a voxel that only one side has allocated must be in its pristine never-observed state.
"""
import numpy as np
import pytest

from scenarios import D2_INF, P_DEFAULT, EnvelopeOracle, assert_envelope

pytestmark = pytest.mark.gpu


def make(oracle_libs, kind, origin, res, reserve, envelope=0):
    """envelope = K: the oracle side is the reference plus K shuffled-order replays of it (scenarios.EnvelopeOracle)."""
    import fiesta_amd
    if kind == "ref" and not oracle_libs.available("ref", "hash"):
        kind = "port"
    gpu = fiesta_amd.ESDFMap(origin, res, reserve_size=reserve, mode="hash")
    mk = lambda: oracle_libs.OracleMap(origin, res, reserve_size=reserve, mode="hash", kind=kind)   # noqa: E731
    cpu = EnvelopeOracle(mk, k=envelope) if envelope else mk()
    for m in (gpu, cpu):
        m.SetParameters(*P_DEFAULT)
        m.SetOriginalRange()
    return gpu, cpu


def cycles(gpu, cpu, occ_vox, free_vox, n):
    for _ in range(n):
        for v, o in ((occ_vox, 1), (free_vox, 0)):
            if len(v):
                gpu.SetOccupancy(np.asarray(v, np.int32), o)
                cpu.SetOccupancyVox(v, o)
        assert gpu.CheckUpdate() == cpu.CheckUpdate()
        a, b = gpu.UpdateOccupancy(True), cpu.UpdateOccupancy(True)
        assert a == b and (gpu.last_insert, gpu.last_delete) == (cpu.last_insert, cpu.last_delete)
    sg, sc = gpu.UpdateESDF(), cpu.UpdateESDF()
    assert (sg["inserted"], sg["deleted"]) == (sc["inserted"], sc["deleted"])
    return sg


def compare(gpu, cpu):
    g, c = gpu.download_hash(), cpu.dump_hash()
    key = lambda v: (v[:, 0].astype(np.int64) + 100000) * (1 << 40) + (v[:, 1].astype(np.int64) + 100000) * (1 << 20) + v[:, 2] + 100000  # noqa: E731
    kg, kc = key(g["vox"]), key(c["vox"])
    assert len(np.unique(kg)) == len(kg)
    cvalid = c["vox"][:, 0] != -10000
    c = {k: v[cvalid] for k, v in c.items()}
    kc = kc[cvalid]
    og, oc = np.argsort(kg), np.argsort(kc)
    kg, kc = kg[og], kc[oc]
    g = {k: v[og] for k, v in g.items()}
    c = {k: v[oc] for k, v in c.items()}
    both_g = np.isin(kg, kc)
    both_c = np.isin(kc, kg)
    # voxels only one side allocated: pristine
    assert np.all(g["d2"][~both_g] == -1) and np.all(g["occ"][~both_g] == 0)
    assert np.all(c["dist"][~both_c] == -10000) and np.all(c["occ"][~both_c] == 0)
    gd2, gcoc, gocc, gvox = g["d2"][both_g].astype(np.int64), g["coc"][both_g].astype(np.int64), g["occ"][both_g], g["vox"][both_g].astype(np.int64)
    cdist, ccoc, cocc = c["dist"][both_c], c["coc"][both_c].astype(np.int64), c["occ"][both_c]
    assert np.array_equal(gocc, cocc)
    cd2 = np.where(cdist < 0, -1, np.where(ccoc[:, 0] == -10000, D2_INF, ((gvox - ccoc) ** 2).sum(-1)))
    mism = int((gd2 != cd2).sum())
    have = (gd2 >= 0) & (gd2 != D2_INF)
    assert np.array_equal(((gvox[have] - gcoc[have]) ** 2).sum(-1), gd2[have])
    # every closest obstacle is an occupied voxel
    occ_keys = set(kg[g["occ"] == 1].tolist())
    ck = key(gcoc[have].astype(np.int64))
    assert all(k in occ_keys for k in np.unique(ck).tolist())
    rep = {"common": int(both_g.sum()), "finite": int(have.sum()), "d2_mismatch": mism, "pages": len(kg) // 8192}
    if hasattr(cpu, "judge"):   # an EnvelopeOracle: the reference's own order spread on this scenario (scenarios.py)
        rep["envelope"] = cpu.judge(gd2, kg[both_g])
    return rep


def test_hash_insert_delete_fully_observed_region(hip_lib, oracle_libs, best_oracle_kind):
    gpu, cpu = make(oracle_libs, best_oracle_kind, (0.0, 0.0, 0.0), 0.1, 1000)
    n = 40
    g = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1).reshape(-1, 3).astype(np.int32) - 7
    cycles(gpu, cpu, [], g, 1)            # negative coordinates, several pages, pool growth from a tiny reserve
    rng = np.random.RandomState(9)
    S = (rng.randint(0, n, (250, 3)) - 7).astype(np.int32)
    cycles(gpu, cpu, S, [], 3)
    rep = compare(gpu, cpu)
    assert rep["d2_mismatch"] == 0 and rep["finite"] == n ** 3, rep
    cycles(gpu, cpu, (rng.randint(0, n, (80, 3)) - 7).astype(np.int32), S[:120], 6)
    rep = compare(gpu, cpu)
    assert rep["d2_mismatch"] == 0, rep
    # queries
    vox = (rng.randint(-3, n + 3, (600, 3)) - 7).astype(np.int32)
    assert np.array_equal(gpu.GetDistance(vox), cpu.GetDistanceVox(vox))
    assert np.array_equal(gpu.GetOccupancy(vox), cpu.GetOccupancyVox(vox))
    pos = (rng.rand(800, 3) * (n - 4) + 2 - 7) * 0.1
    assert np.array_equal(gpu.GetDistance(pos), cpu.GetDistancePos(pos))
    dg, gg = gpu.GetDistWithGradTrilinear(pos)
    dc, gc = cpu.GetDistWithGradTrilinear(pos)
    assert np.array_equal(dg, dc) and np.array_equal(gg, gc)
    # delete everything
    occ = np.array([v for v in map(tuple, np.concatenate([S, vox]))][:0], np.int32).reshape(0, 3)
    d = gpu.download_hash()
    occ = d["vox"][d["occ"] == 1]
    cycles(gpu, cpu, [], occ, 6)
    rep = compare(gpu, cpu)
    assert rep["d2_mismatch"] == 0 and rep["finite"] == 0, rep


def test_hash_streaming_window_positions(hip_lib, oracle_libs, best_oracle_kind):
    """Config-4 shape: a moving observation window streams in new space (pages appear), obstacles come and go."""
    res = 0.05
    gpu, cpu = make(oracle_libs, best_oracle_kind, (1.0, -2.0, 0.5), res, 100000, envelope=5)
    rng = np.random.RandomState(2)
    live = np.zeros((0, 3))
    for frame in range(5):
        c = np.array([0.4 * frame, 0.2 * frame, 0.0]) + [1.0, -2.0, 0.5]
        box = c + (np.stack(np.meshgrid(*[np.arange(28)] * 3, indexing="ij"), -1).reshape(-1, 3) + 0.5) * res
        new = c + rng.rand(60, 3) * 28 * res
        gone = live[: len(live) // 3]
        live = np.concatenate([live[len(live) // 3:], new])
        for k in range(4):
            for m, f in ((gpu, gpu.SetOccupancy), (cpu, cpu.SetOccupancyPos)):
                if k == 0:
                    f(box, 0)
                f(live, 1)
                if len(gone):
                    f(gone, 0)
            a, b = gpu.UpdateOccupancy(True), cpu.UpdateOccupancy(True)
            assert a == b and (gpu.last_insert, gpu.last_delete) == (cpu.last_insert, cpu.last_delete)
        sg, sc = gpu.UpdateESDF(), cpu.UpdateESDF()
        assert (sg["inserted"], sg["deleted"]) == (sc["inserted"], sc["deleted"])
        rep = compare(gpu, cpu)
        # the union of boxes is only partially observed at its rim: the reference's order dependence applies -> judged
        # against the envelope of its own shuffled runs
        assert_envelope(rep, f"frame {frame}", strict=gpu.only_levels)
    assert rep["pages"] >= 2


def test_hash_mode_errors_are_loud(hip_lib):
    import fiesta_amd
    m = fiesta_amd.ESDFMap((0, 0, 0), 0.1, reserve_size=0, mode="hash")
    m.SetParameters(*P_DEFAULT)
    # the map is unbounded (src/ESDFMap.cpp:46-48): a voxel outside the current window moves the window, nothing is lost
    assert m.SetOccupancy(np.array([[5, 5, 5]], np.int32), 1)[0] != -10000
    assert m.SetOccupancy(np.array([[600, 0, 0]], np.int32), 1)[0] != -10000
    org, moves = m.hash_window()
    assert moves == 1 and org[0] <= 600 < org[0] + 1024 and org[0] % 16 == 0 and tuple(org[1:]) == (-512, -512)
    m.UpdateOccupancy(True)
    assert m.UpdateESDF()["dropped_observations"] == 0
    # one batch wider than the window cannot be held at once: its far ends are dropped, and counted
    m.SetOccupancy(np.array([[0, 0, 0], [0, 2000, 0]], np.int32), 1)
    m.UpdateOccupancy(True)
    assert m.UpdateESDF()["dropped_observations"] >= 1
    with pytest.raises(fiesta_amd.FiestaHipError):
        m.snapshot_restore(0)   # hash-mode maps keep one copy of the state words for the benchmark unit: no restore
    with pytest.raises(fiesta_amd.FiestaHipError):
        m.snapshot_save(1)
    assert m.GetDistance(np.array([[5, 5, 5]], np.int32))[0] == 10000.0


def test_hash_wave_reaches_unallocated_space(hip_lib, oracle_libs, best_oracle_kind):
    """The observed region is exactly ONE page (tile-aligned 16x16x32 box): every wave runs into unallocated
    neighbour tiles, which must be neither woken nor visited (regression: bitmap writes at page index -1)."""
    gpu, cpu = make(oracle_libs, best_oracle_kind, (0.0, 0.0, 0.0), 0.1, 0, envelope=5)
    g = np.stack(np.meshgrid(np.arange(16), np.arange(16), np.arange(32), indexing="ij"), -1).reshape(-1, 3).astype(np.int32)
    cycles(gpu, cpu, [], g, 1)
    cycles(gpu, cpu, np.array([[8, 8, 16], [0, 0, 0], [15, 15, 31]], np.int32), [], 3)
    rep = compare(gpu, cpu)
    assert rep["d2_mismatch"] == 0 and rep["finite"] == 16 * 16 * 32 and rep["pages"] == 1, rep
    cycles(gpu, cpu, [], np.array([[8, 8, 16]], np.int32), 6)
    rep = compare(gpu, cpu)
    assert rep["d2_mismatch"] == 0, rep
    # grow sideways: the neighbours get pages now and join the propagation
    g2 = g + np.array([16, 0, 0], np.int32)
    cycles(gpu, cpu, [], g2, 1)
    cycles(gpu, cpu, np.array([[20, 3, 5]], np.int32), [], 3)
    rep = compare(gpu, cpu)
    assert rep["pages"] == 2, rep
    assert_envelope(rep, "freshly observed free space next to a field", strict=gpu.only_levels)  # (the order-dependent regime)


def test_hash_c4_stream_box_observe(hip_lib, oracle_libs, best_oracle_kind):
    """The bench's C4 stream (tests/scenarios.py: c4_frame) through the device-side box observe of the paged map, first
    frames, against the hash-table reference fed voxel by voxel; and the device-side "updated voxels" unit against a
    count made from two downloads."""
    from scenarios import box_voxels, c4_frame
    gpu, cpu = make(oracle_libs, best_oracle_kind, (0.0, 0.0, 0.0), 0.05, 1000000, envelope=3)
    for k in range(4):
        lo, hi, occ = c4_frame(k)
        gpu.SetOccupancyBox(lo, hi, 0)
        gpu.SetOccupancy(occ, 1)
        cpu.SetOccupancyVox(box_voxels(lo, hi), 0)
        cpu.SetOccupancyVox(occ, 1)
        a, b = gpu.UpdateOccupancy(True), cpu.UpdateOccupancy(True)
        assert a == b and (gpu.last_insert, gpu.last_delete) == (cpu.last_insert, cpu.last_delete), k
        before = gpu.download_hash() if k == 3 else None
        gpu.snapshot_save(0)
        sg, sc = gpu.UpdateESDF(), cpu.UpdateESDF()
        assert (sg["inserted"], sg["deleted"]) == (sc["inserted"], sc["deleted"])
        n_upd = gpu.snapshot_count_updated(0)
        if before is not None:
            after = gpu.download_hash()
            nb = len(before["d2"])
            assert np.array_equal(after["vox"][:nb], before["vox"])          # pages are only appended
            d2_changed = int((after["d2"][:nb] != before["d2"]).sum()) + int((after["d2"][nb:] != -1).sum())
            any_changed = d2_changed + int(((after["d2"][:nb] == before["d2"]) & np.any(after["coc"][:nb] != before["coc"], axis=1)).sum())
            assert d2_changed <= n_upd <= any_changed, (d2_changed, n_upd, any_changed)
    rep = compare(gpu, cpu)
    # frames 0..2 only observe (3 hits make an obstacle): frame 3 inserts the whole visible surface at once
    assert rep["finite"] > 500000
    assert_envelope(rep, "C4 frame 3", strict=gpu.only_levels)


def test_hash_visualisation_getters(hip_lib, oracle_libs, best_oracle_kind):
    """GetPointCloud / GetSliceMarker on the block store (src/ESDFMap.cpp:547-566, 657-677) against the reference's own
    messages, with and without an update range."""
    from test_gpu_dense_parity import check_vis_getters
    gpu, cpu = make(oracle_libs, best_oracle_kind, (0.3, -0.2, 0.1), 0.1, 1000)
    n = 36
    g = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1).reshape(-1, 3).astype(np.int32) - 11
    cycles(gpu, cpu, [], g, 1)
    S = (np.random.RandomState(8).randint(0, n, (200, 3)) - 11).astype(np.int32)
    cycles(gpu, cpu, S, [], 3)
    assert check_vis_getters(gpu, cpu, (-11, 0, 13), ((-100, 100), (-3, 4), (5, 5), (60, 70))) > 3000
    for m in (gpu, cpu):
        m.SetUpdateRange((0.0, -0.5, -0.3), (1.9, 1.4, 1.6))
    assert 0 < check_vis_getters(gpu, cpu, (0, 13), ((-100, 100), (2, 9))) < 3000


def _island(c, half, rng, n_obst):
    lo = np.asarray(c) - half
    box = np.stack(np.meshgrid(*[np.arange(2 * h) for h in half], indexing="ij"), -1).reshape(-1, 3).astype(np.int32) + lo
    obst = (lo + np.stack([rng.randint(0, 2 * h, n_obst) for h in half], -1)).astype(np.int32)
    return box.astype(np.int32), obst


def test_hash_window_follows_a_travelling_sensor(hip_lib, oracle_libs, best_oracle_kind):
    """The map is unbounded like the reference's: observed islands strung along 2400 voxels of travel (the window is 1024
    wide and has to move several times; islands do not see each other through unobserved space, so every one of them --
    resident or parked -- must equal the unbounded reference bit for bit), negative and positive coordinates."""
    gpu, cpu = make(oracle_libs, best_oracle_kind, (0.0, 0.0, 0.0), 0.1, 1000)
    rng = np.random.RandomState(12)
    stations = [(-900 + 300 * k, 40 * k - 100, 10 * k) for k in range(9)]   # x from -900 to +1500
    for c in stations:
        box, obst = _island(c, (14, 14, 10), rng, 25)
        cycles(gpu, cpu, [], box, 1)
        cycles(gpu, cpu, obst, [], 3)
        cycles(gpu, cpu, [], obst[:8], 6)       # some obstacles vanish again
        org, moves = gpu.hash_window()
        assert all(org[k] <= c[k] - 14 and c[k] + 14 <= org[k] + 1024 for k in range(3))
    assert moves >= 2
    rep = compare(gpu, cpu)
    assert rep["d2_mismatch"] == 0 and rep["finite"] == len(stations) * 28 * 28 * 20, rep
    # queries: resident voxels answer like the reference -- and so do PARKED ones (the map-wide page table of the query
    # kernels: a planner may ask about a goal far from the sensor, src/ESDFMap.cpp:732-765 answers for any voxel)
    for st_ in stations[:-1]:
        cc = np.array(st_, np.int32)
        vq = (cc + rng.randint(-16, 16, (300, 3))).astype(np.int32)
        assert np.array_equal(gpu.GetDistance(vq), cpu.GetDistanceVox(vq)), st_
        assert np.array_equal(gpu.GetOccupancy(vq), cpu.GetOccupancyVox(vq)), st_
        pq = (cc + rng.rand(200, 3) * 20 - 10) * 0.1
        dq, gq = gpu.GetDistWithGradTrilinear(pq)
        dr, gr = cpu.GetDistWithGradTrilinear(pq)
        assert np.array_equal(dq, dr) and np.array_equal(gq, gr), st_
    org, _ = gpu.hash_window()
    assert not all(org[k] <= stations[0][k] < org[k] + 1024 for k in range(3)), "the first island must be parked by now"
    c = np.array(stations[-1], np.int32)
    vox = (c + rng.randint(-16, 16, (400, 3))).astype(np.int32)
    assert np.array_equal(gpu.GetDistance(vox), cpu.GetDistanceVox(vox))
    pos = (c + rng.rand(300, 3) * 20 - 10) * 0.1
    dg, gg = gpu.GetDistWithGradTrilinear(pos)
    dc, gc = cpu.GetDistWithGradTrilinear(pos)
    assert np.array_equal(dg, dc) and np.array_equal(gg, gc)
    far = np.array([stations[0]], np.int32)
    assert gpu.GetDistance(far)[0] == cpu.GetDistanceVox(far)[0] != 10000.0
    # the window brought back over them: they take part in updates again
    gpu.hash_recentre(stations[0])
    assert gpu.UpdateESDF()["dropped_observations"] == 0
    vox = (np.array(stations[0], np.int32) + rng.randint(-14, 14, (400, 3))).astype(np.int32)
    assert np.array_equal(gpu.GetDistance(vox), cpu.GetDistanceVox(vox))
    rep = compare(gpu, cpu)
    assert rep["d2_mismatch"] == 0, rep


def test_hash_window_moves_for_device_batches(hip_lib, oracle_libs, best_oracle_kind):
    """fiesta_hip_set_occupancy_vox_dev: the batch lives in HBM, so the mark kernel reports "outside the window", the
    bounding box is reduced on the device, and the window moves before anything is counted -- nothing may be dropped."""
    import torch
    gpu, cpu = make(oracle_libs, best_oracle_kind, (0.0, 0.0, 0.0), 0.1, 1000)
    rng = np.random.RandomState(3)
    for c in ((0, 0, 0), (700, -650, 40), (-3000, 90, 2100)):
        box, obst = _island(c, (10, 10, 8), rng, 12)
        for v, o, reps in ((box, 0, 1), (obst, 1, 3)):
            for _ in range(reps):
                dv = torch.from_numpy(v).cuda()
                do = torch.full((len(v),), o, dtype=torch.int32, device="cuda")
                torch.cuda.synchronize()
                gpu.SetOccupancyDevice(dv.data_ptr(), do.data_ptr(), len(v))
                cpu.SetOccupancyVox(v, o)
                assert gpu.UpdateOccupancy(True) == cpu.UpdateOccupancy(True)
            sg, sc = gpu.UpdateESDF(), cpu.UpdateESDF()
            assert (sg["inserted"], sg["deleted"], sg["dropped_observations"]) == (sc["inserted"], sc["deleted"], 0)
    assert gpu.hash_window()[1] == 2
    rep = compare(gpu, cpu)
    assert rep["d2_mismatch"] == 0 and rep["finite"] == 3 * 20 * 20 * 16, rep


def _compare_away_from_faces(gpu, cpu, margin):
    """Resident voxels farther than `margin` from the window's x faces against the unbounded reference: (mismatches, n)."""
    g, c = gpu.download_hash(), cpu.dump_hash()
    key = lambda v: (v[:, 0].astype(np.int64) + 100000) * (1 << 40) + (v[:, 1].astype(np.int64) + 100000) * (1 << 20) + v[:, 2] + 100000  # noqa: E731
    ok = c["vox"][:, 0] != -10000
    kc = key(c["vox"][ok])
    order = np.argsort(kc)
    kc, cdist, cocc = kc[order], c["dist"][ok][order], c["occ"][ok][order]
    o = gpu.hash_window()[0]
    gd = gpu.distance_from_d2(g["d2"])
    x = g["vox"][:, 0].astype(np.int64)
    sel = (gd != -10000.0) & (x >= o[0]) & (x < o[0] + 1024) & (np.abs(x - o[0]) > margin) & (np.abs(x - (o[0] + 1024)) > margin)
    kg = key(g["vox"][sel])
    at = np.searchsorted(kc, kg)
    assert np.all(at < len(kc)) and np.array_equal(kc[at], kg)     # everything the GPU observed, the reference holds
    bad = (np.abs(cdist[at] - gd[sel]) > 1e-12) | (cocc[at] != g["occ"][sel])
    return int(bad.sum()), int(sel.sum())


def test_hash_window_corridor_and_return(hip_lib, oracle_libs, best_oracle_kind):
    """One CONTIGUOUS observed corridor 1500 voxels long, travelled twice (out: observe free space; back: obstacles
    appear).  What the window leaves behind is parked; the field inside the window is that of the obstacles inside the
    window, so away from the window's faces (farther than any distance in the scene) every resident voxel equals the
    unbounded reference; pages that rejoin are rebuilt and pick up what changed next to them while they were parked."""
    gpu, cpu = make(oracle_libs, best_oracle_kind, (0.0, 0.0, 0.0), 0.1, 100000)
    rng = np.random.RandomState(5)
    step, half = 48, (32, 12, 8)
    xs = list(range(0, 1500, step))
    for x in xs:                                   # out
        box, _ = _island((x, 0, 0), half, rng, 1)
        cycles(gpu, cpu, [], box, 1)
    org, moves = gpu.hash_window()
    assert moves >= 1 and org[0] > -512
    for x in reversed(xs):                         # and back
        _, obst = _island((x, 0, 0), half, rng, 30)
        cycles(gpu, cpu, obst, [], 3)
    org, moves2 = gpu.hash_window()
    assert moves2 > moves and org[0] <= -32
    bad, n = _compare_away_from_faces(gpu, cpu, 40)
    assert bad == 0 and n > 150000, (bad, n)
    # while the far end of the corridor is parked, obstacles appear right at the window's face
    top = int(org[0]) + 1024
    edge = np.array([[top - 2, 0, 0], [top - 5, 3, -2], [top - 1, -6, 4]], np.int32)
    cycles(gpu, cpu, edge, [], 3)
    # travel out again: the parked pages rejoin and are rebuilt -- they must pick up the new obstacles next to them
    gpu.hash_recentre((top + 200, 0, 0))
    gpu.UpdateESDF()
    bad, n = _compare_away_from_faces(gpu, cpu, 40)
    assert bad == 0 and n > 150000, (bad, n)
    near = (edge[0] + np.stack(np.meshgrid(np.arange(1, 24), np.arange(-8, 8), np.arange(-6, 6), indexing="ij"), -1).reshape(-1, 3)).astype(np.int32)
    assert np.array_equal(gpu.GetDistance(near), cpu.GetDistanceVox(near))


def test_hash_scalar_queries_through_the_host_brick_cache(hip_lib, oracle_libs, best_oracle_kind):
    """r06: ONE position per call on a hash-block map is answered from a host-side cache of 16^3-voxel bricks
    (hash_map.hip: HostBricks) instead of a kernel launch per call.  Same bits as the batch kernels and as the reference --
    allocated space, never-allocated space, negative coordinates --, one fetch per brick, and every call that may change the
    field invalidates the cache."""
    gpu, cpu = make(oracle_libs, best_oracle_kind, (0.0, 0.0, 0.0), 0.1, 1000)
    n = 40
    g = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1).reshape(-1, 3).astype(np.int32) - 7
    cycles(gpu, cpu, [], g, 1)
    rng = np.random.RandomState(19)
    S = (rng.randint(0, n, (250, 3)) - 7).astype(np.int32)
    cycles(gpu, cpu, S, [], 3)
    pos = (rng.rand(500, 3) * (n - 4) + 2 - 7) * 0.1
    vox = (rng.randint(-3, n + 3, (300, 3)) - 7).astype(np.int32)       # some of them in space nobody ever allocated
    dgb, ggb = gpu.GetDistWithGradTrilinear(pos)                        # the batch kernels ...
    dc, gc = cpu.GetDistWithGradTrilinear(pos)                          # ... equal the reference
    assert np.array_equal(dgb, dc) and np.array_equal(ggb, gc)
    before = gpu.host_cache_fetches
    for i, p in enumerate(pos):                                         # ONE position per call
        d, gr = gpu.GetDistWithGradTrilinear(p)
        assert d == dc[i] and np.array_equal(gr, gc[i]), (i, p)
        assert gpu.GetDistance(p) == cpu.GetDistancePos(p[None])[0]
        assert gpu.GetOccupancy(p) == cpu.GetOccupancyPos(p[None])[0]
    for v in vox:
        assert gpu.GetDistance(v) == cpu.GetDistanceVox(v[None])[0], v
        assert gpu.GetOccupancy(v) == cpu.GetOccupancyVox(v[None])[0], v
    fetched = gpu.host_cache_fetches - before
    assert 0 < fetched <= 5 ** 3, fetched                               # the queried space spans at most 5 bricks per axis
    for p in pos[:50]:
        gpu.GetDistWithGradTrilinear(p)
    assert gpu.host_cache_fetches - before <= fetched + 8                # (two-way sets: a third brick in a set evicts, rarely)
    # the field changes: the cache must not answer from before
    cycles(gpu, cpu, [], S[:120], 6)
    for i, p in enumerate(pos[:150]):
        d, gr = gpu.GetDistWithGradTrilinear(p)
        d1, g1 = cpu.GetDistWithGradTrilinear(p[None])
        assert d == d1[0] and np.array_equal(gr, g1[0]), (i, p)
    assert gpu.host_cache_fetches - before > fetched
