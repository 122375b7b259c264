"""fiesta_hip_save / fiesta_hip_load: a map restored from a checkpoint file continues exactly like the map that wrote it
-- checked in the middle of a frame (observations counted but not fused, insert/delete queues not yet consumed) and
against the oracle, array and hash-block maps (the latter with a moved window and parked pages)."""
import numpy as np
import pytest

from scenarios import P_DEFAULT, all_voxels

pytestmark = pytest.mark.gpu


def _same_dense(a, b, ids=True):
    fa, fb = a.download_field(), b.download_field()
    for k in fa:          # (after an update, which of several equidistant obstacles a voxel names is not deterministic)
        assert k == "coc" and not ids or np.array_equal(fa[k], fb[k]), k
    ca, cb = a.download_counts(), b.download_counts()
    assert np.array_equal(ca[0], cb[0]) and np.array_equal(ca[1], cb[1])


@pytest.mark.parametrize("engine", ["rounds", "bulk"])
def test_dense_checkpoint_mid_frame(hip_lib, oracle_libs, best_oracle_kind, tmp_path, engine):
    import fiesta_amd
    n, res = 40, 0.1
    mk = lambda: fiesta_amd.ESDFMap((0, 0, 0), res, (n * res,) * 3, update_engine=engine)   # noqa: E731
    a, cpu = mk(), oracle_libs.OracleMap((0, 0, 0), res, (n * res,) * 3, kind=best_oracle_kind)
    for m in (a, cpu):
        m.SetParameters(*P_DEFAULT)
        m.SetOriginalRange()
    rng = np.random.RandomState(2)
    S = rng.randint(0, n, (300, 3)).astype(np.int32)
    a.SetOccupancy(all_voxels(n), 0), cpu.SetOccupancyVox(all_voxels(n), 0)
    a.UpdateOccupancy(True), cpu.UpdateOccupancy(True)
    a.UpdateESDF(), cpu.UpdateESDF()
    for _ in range(3):
        a.SetOccupancy(S, 1), cpu.SetOccupancyVox(S, 1)
        a.UpdateOccupancy(True), cpu.UpdateOccupancy(True)
    # now: 300 inserts queued, and a further batch observed but not fused
    T = rng.randint(0, n, (200, 3)).astype(np.int32)
    a.SetOccupancy(T, 1), cpu.SetOccupancyVox(T, 1)
    a.SetUpdateRange((0.3, 0.2, 0.1), (2.5, 2.6, 2.7))      # (the range is part of the state: the getters show it)
    path = str(tmp_path / "dense.ckpt")
    a.save(path)
    b = mk()                                    # a fresh process would do exactly this
    b.SetParameters(0.6, 0.4, 0.2, 0.8, 0.7)    # (overwritten by the checkpoint)
    b.load(path)
    _same_dense(a, b)
    pa, pb = a.GetPointCloud(0, n), b.GetPointCloud(0, n)
    assert 0 < len(pa) < 300 and np.array_equal(pa[np.lexsort(pa.T)], pb[np.lexsort(pb.T)])
    for m in (a, b):
        m.SetOriginalRange()
        assert m.CheckUpdate()
    ra, rb, rc = a.UpdateOccupancy(True), b.UpdateOccupancy(True), cpu.UpdateOccupancy(True)
    assert ra == rb == rc and (a.last_insert, a.last_delete) == (b.last_insert, b.last_delete) == (cpu.last_insert, cpu.last_delete)
    sa, sb, sc = a.UpdateESDF(), b.UpdateESDF(), cpu.UpdateESDF()
    assert (sa["inserted"], sa["deleted"]) == (sb["inserted"], sb["deleted"]) == (sc["inserted"], sc["deleted"])
    _same_dense(a, b, ids=False)
    ref = cpu.dump_dense(("dist",))["dist"]
    got = b.distance_from_d2(b.download_field(("d2",))["d2"])
    assert np.array_equal(got, ref)
    q = rng.rand(500, 3) * (n - 3) * res + res
    assert np.array_equal(b.GetDistance(q), cpu.GetDistancePos(q))
    # a file of another geometry is refused, loudly
    other = fiesta_amd.ESDFMap((0, 0, 0), res, ((n + 1) * res,) * 3)
    with pytest.raises(fiesta_amd.FiestaHipError):
        other.load(path)
    with pytest.raises(fiesta_amd.FiestaHipError):
        b.load(str(tmp_path / "missing.ckpt"))
    # ... and so is a damaged one, BEFORE any state of the map is replaced: truncated, grown, and one whose
    # queue length field claims more entries than the grid has voxels
    import os
    raw = open(path, "rb").read()
    assert not os.path.exists(path + ".tmp")            # (written to a temporary name, renamed when complete)
    before = b.download_field()
    # (88-byte header, then the counters: C_TOUCHED, C_INSERT, C_DELETE lead)
    for name, blob in (("short", raw[:len(raw) // 2]), ("long", raw + b"\0" * 64),
                       ("queues", raw[:88 + 8] + (2 ** 40).to_bytes(8, "little") + raw[88 + 16:])):
        bad = str(tmp_path / f"{name}.ckpt")
        open(bad, "wb").write(blob)
        with pytest.raises(fiesta_amd.FiestaHipError):
            b.load(bad)
        after = b.download_field()
        assert all(np.array_equal(before[k], after[k]) for k in before), name
    b.load(path)                                        # the map is still usable


def test_hash_checkpoint_with_moved_window(hip_lib, oracle_libs, best_oracle_kind, tmp_path):
    import fiesta_amd
    from test_gpu_hash_parity import _island, compare, cycles, make
    a, cpu = make(oracle_libs, best_oracle_kind, (0.0, 0.0, 0.0), 0.1, 1000)
    rng = np.random.RandomState(7)
    for c in ((0, 0, 0), (900, 40, -20)):                  # the second island parks the first
        box, obst = _island(c, (12, 12, 8), rng, 20)
        cycles(a, cpu, [], box, 1)
        cycles(a, cpu, obst, [], 3)
    assert a.hash_window()[1] == 1
    box, obst = _island((930, 40, -20), (12, 12, 8), rng, 15)
    for m, f in ((a, a.SetOccupancy), (cpu, cpu.SetOccupancyVox)):     # pending: counted, not fused
        f(box, 0)
        f(obst, 1)
    path = str(tmp_path / "hash.ckpt")
    a.save(path)
    b = fiesta_amd.ESDFMap((0.0, 0.0, 0.0), 0.1, reserve_size=10, mode="hash")
    b.SetOccupancy(np.array([[5, 5, 5]], np.int32), 1)     # state the checkpoint has to wipe
    b.load(path)
    assert np.array_equal(b.hash_window()[0], a.hash_window()[0]) and b.hash_window()[1] == 1
    for _ in range(3):
        r = [m.UpdateOccupancy(True) for m in (a, b, cpu)]
        assert r[0] == r[1] == r[2]
        for m, f in ((a, a.SetOccupancy), (b, b.SetOccupancy), (cpu, cpu.SetOccupancyVox)):
            f(obst, 1)
    for m in (a, b, cpu):
        m.UpdateOccupancy(True)
    sa, sb, sc = a.UpdateESDF(), b.UpdateESDF(), cpu.UpdateESDF()
    assert (sa["inserted"], sa["deleted"]) == (sb["inserted"], sb["deleted"]) == (sc["inserted"], sc["deleted"])
    da, db = a.download_hash(), b.download_hash()
    for k in ("vox", "d2", "occ"):
        assert np.array_equal(da[k], db[k]), k
    rep = compare(b, cpu)
    assert rep["d2_mismatch"] <= 2, rep          # (freshly observed free space next to a field: order-dependent regime;
                                                 #  the envelope form of this bound: test_hash_wave_reaches_unallocated_space)
    # bring the window back: the parked island rejoins in the restored map as well
    b.hash_recentre((0, 0, 0))
    b.UpdateESDF()
    v = rng.randint(-10, 10, (200, 3)).astype(np.int32)
    assert np.array_equal(b.GetDistance(v), cpu.GetDistanceVox(v))
