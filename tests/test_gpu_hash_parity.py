"""GPU parity of the sparse ("hash-block") map (BASELINE config 4) against the CPU oracle built
with -DHASH_TABLE (verbatim reference when oracle/_ref is present, else the restatement).

The reference allocates 8^3 blocks in arrival order -- also for blocks its neighbour READS touch; the GPU map allocates
16x16x32 pages when a voxel in them is observed.  Internal indices and the allocated set therefore differ by design;
what must match, voxel by voxel over the UNION of both allocated sets, is the state: d^2 / occupancy bit-exact, closest
obstacle tie-equivalent, and a voxel that only one side has allocated must be in its pristine never-observed state.
"""
import numpy as np
import pytest

from scenarios import D2_INF, P_DEFAULT

pytestmark = pytest.mark.gpu


def make(oracle_libs, kind, origin, res, reserve):
    import fiesta_amd
    if kind == "ref" and not oracle_libs.available("ref", "hash"):
        kind = "port"
    gpu = fiesta_amd.ESDFMap(origin, res, reserve_size=reserve, mode="hash")
    cpu = oracle_libs.OracleMap(origin, res, reserve_size=reserve, mode="hash", kind=kind)
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
    return {"common": int(both_g.sum()), "finite": int(have.sum()), "d2_mismatch": mism, "pages": len(kg) // 8192}


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
    gpu, cpu = make(oracle_libs, best_oracle_kind, (1.0, -2.0, 0.5), res, 100000)
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
        # the union of boxes is only partially observed at its rim: the reference's order dependence applies
        assert rep["d2_mismatch"] <= max(30, 0.02 * rep["finite"]), rep
    assert rep["pages"] >= 2


def test_hash_mode_errors_are_loud(hip_lib):
    import fiesta_amd
    m = fiesta_amd.ESDFMap((0, 0, 0), 0.1, reserve_size=0, mode="hash")
    m.SetParameters(*P_DEFAULT)
    # outside the 1024^3 virtual window: rejected like an out-of-map position in array mode
    assert m.SetOccupancy(np.array([[600, 0, 0]], np.int32), 1)[0] == -10000
    assert m.SetOccupancy(np.array([[5, 5, 5]], np.int32), 1)[0] != -10000
    with pytest.raises(fiesta_amd.FiestaHipError):
        m.snapshot_restore(0)   # hash-mode maps keep one copy of the state words for the benchmark unit: no restore
    with pytest.raises(fiesta_amd.FiestaHipError):
        m.snapshot_save(1)
    assert m.GetDistance(np.array([[5, 5, 5]], np.int32))[0] == 10000.0


def test_hash_wave_reaches_unallocated_space(hip_lib, oracle_libs, best_oracle_kind):
    """The observed region is exactly ONE page (tile-aligned 16x16x32 box): every wave runs into unallocated
    neighbour tiles, which must be neither woken nor visited (regression: bitmap writes at page index -1)."""
    gpu, cpu = make(oracle_libs, best_oracle_kind, (0.0, 0.0, 0.0), 0.1, 0)
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
    assert rep["d2_mismatch"] <= 2 and rep["pages"] == 2, rep  # freshly observed free space: order-dependent regime


def test_hash_c4_stream_box_observe(hip_lib, oracle_libs, best_oracle_kind):
    """The bench's C4 stream (tests/scenarios.py: c4_frame) through the device-side box observe of the paged map, first
    frames, against the hash-table reference fed voxel by voxel; and the device-side "updated voxels" unit against a
    count made from two downloads."""
    from scenarios import box_voxels, c4_frame
    gpu, cpu = make(oracle_libs, best_oracle_kind, (0.0, 0.0, 0.0), 0.05, 1000000)
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
    assert rep["finite"] > 500000 and rep["d2_mismatch"] <= 0.02 * rep["finite"], rep
