"""The gate in front of the bulk feature transform (DenseMap::bulk_eligible).  The transform writes the exact
Euclidean transform of the occupied set; the reference's field equals that only while nothing has ever gated its
propagation.  Each scenario drives THREE maps through the same calls -- engine "bulk" (the transform whenever the gate lets
it), engine "rounds", and the verbatim reference -- across a state in which the transform would be wrong, and back; the
gate must keep the two engines voxel-for-voxel on the reference's side of the parity contract, and must re-open where that
is legitimate (checked through the `bulk` flag of the update's statistics)."""
import numpy as np
import pytest

from scenarios import P_DEFAULT, Both, EnvelopeOracle, all_voxels, assert_envelope, assert_exact, compare_dense

pytestmark = pytest.mark.gpu


def _trio(oracle_libs, kind, n, res=0.1, envelope=0):
    import fiesta_amd
    size = (n * res,) * 3
    mk = lambda: oracle_libs.OracleMap((0, 0, 0), res, size, kind=kind)   # noqa: E731
    cpu = EnvelopeOracle(mk, k=envelope) if envelope else mk()
    maps = [fiesta_amd.ESDFMap((0, 0, 0), res, size, update_engine=e) for e in ("bulk", "rounds")]
    pairs = [Both(m, cpu) for m in maps]
    for m in maps + [cpu]:
        m.SetParameters(*P_DEFAULT)
        m.SetOriginalRange()
    return maps, cpu, pairs


class Trio:
    """The same call on the bulk-engine map, the rounds-engine map and the oracle (the oracle only once)."""

    def __init__(self, maps, cpu):
        self.maps, self.cpu = maps, cpu

    def observe(self, vox, occ):
        for m in self.maps:
            m.SetOccupancy(vox, occ, want_ret=False)
        self.cpu.SetOccupancyVox(vox, occ)

    def fuse(self, global_map=True):
        r = [m.UpdateOccupancy(global_map) for m in self.maps] + [self.cpu.UpdateOccupancy(global_map)]
        assert r[0] == r[1] == r[2]
        q = [(m.last_insert, m.last_delete) for m in self.maps] + [(self.cpu.last_insert, self.cpu.last_delete)]
        assert q[0] == q[1] == q[2], q

    def cycles(self, occ_vox, free_vox, n):
        for _ in range(n):
            if len(occ_vox):
                self.observe(occ_vox, 1)
            if len(free_vox):
                self.observe(free_vox, 0)
            self.fuse()

    def esdf(self):
        st = [m.UpdateESDF() for m in self.maps]
        self.cpu.UpdateESDF()
        assert not st[1]["bulk"]
        return bool(st[0]["bulk"])

    def set_range(self, lo=None, hi=None):
        for m in self.maps + [self.cpu]:
            m.SetOriginalRange() if lo is None else m.SetUpdateRange(lo, hi)

    def same_fields(self):
        a, b = (m.download_field(("d2", "occ")) for m in self.maps)
        assert np.array_equal(a["occ"], b["occ"])
        return int((a["d2"] != b["d2"]).sum())


def test_first_obstacles_under_a_partial_window_keep_the_transform_off(hip_lib, oracle_libs, best_oracle_kind):
    """a fully observed map WITHOUT obstacles, the first obstacles inserted under a partial window (their waves stop
    at the window's faces: the far half keeps "no obstacle"), then the full window again.  The history flag is cleared for a map
    that held no obstacle before an update -- that clearing must come BEFORE the update's own window is recorded, or this
    sequence leaves the gate open and the exact transform overwrites the half the reference never reached."""
    n = 40
    maps, cpu, _ = _trio(oracle_libs, best_oracle_kind, n, envelope=4)
    t = Trio(maps, cpu)
    rng = np.random.RandomState(12)
    t.observe(all_voxels(n), 0)
    t.fuse()
    t.esdf()
    t.set_range((0.0, 0.0, 0.0), (1.9, n * 0.1, n * 0.1))     # the low-x half
    S = rng.randint(2, 17, (40, 3)).astype(np.int32)
    t.cycles(S, [], 3)
    assert t.esdf() is False                                  # a partial window: never the transform
    far = maps[0].download_field(("d2",))["d2"].reshape((n,) * 3)[24:]
    assert np.all(far == np.iinfo(np.int32).max), "the waves of the first obstacles must stop at the window"
    t.set_range()                                             # the whole map again
    t.cycles(rng.randint(22, n - 1, (20, 3)).astype(np.int32), [], 3)
    assert t.esdf() is False, "the far half was never reached by the first obstacles' waves: the transform would rewrite it"
    for m in maps:
        rep = compare_dense(m, cpu)
        assert_envelope(rep, "first obstacles under a partial window, then widened")
        assert rep["pair_violations"] == 0, rep


def test_window_narrowed_then_widened_keeps_the_transform_off(hip_lib, oracle_libs, best_oracle_kind):
    """Fully observed map; obstacles under the full window (gate open); then updates under a window that covers half the
    map -- inserts never reach the other half, the orphans of deletes over there keep what one pull gave them
    (src/ESDFMap.cpp:351,378) --; then the full window again.  From the first windowed update on the reference's field is
    a function of its history: the exact transform would overwrite the frozen half.  The gate must stay shut until the map
    has been without obstacles."""
    n = 40
    maps, cpu, _ = _trio(oracle_libs, best_oracle_kind, n, envelope=4)
    t = Trio(maps, cpu)
    rng = np.random.RandomState(2)
    t.observe(all_voxels(n), 0)
    t.fuse()
    t.esdf()
    S = rng.randint(1, n - 1, (200, 3)).astype(np.int32)
    t.cycles(S, [], 3)
    assert t.esdf() is True                                   # fully observed, full window: the transform
    assert t.same_fields() == 0
    assert_exact(compare_dense(maps[0], cpu.primary))
    t.set_range((0.0, 0.0, 0.0), (1.9, n * 0.1, n * 0.1))     # the low-x half
    low = S[S[:, 0] < 18]
    t.cycles(rng.randint(2, 16, (30, 3)).astype(np.int32), low[:40], 6)
    assert t.esdf() is False                                  # a partial window: the rounds
    t.set_range()                                             # ... and wide again
    new = rng.randint(1, n - 1, (60, 3)).astype(np.int32)
    t.cycles(new, S[100:140], 6)
    assert t.esdf() is False, "the transform must stay off: the far half still holds what the window froze"
    for m in maps:                                            # both engines inside the reference's own order envelope
        rep = compare_dense(m, cpu)
        # (the two planes of voxels just beyond the narrowed window, x = 19 and 20, held orphans of the deletes that ran under
        #  it: which of those the reference re-seeded follows its list order, which the engines approximate -- level_kernels.hpp:
        #  k_level_outside, dense_map.hip: k_reseed_outside; the allowance is 0.5 % of that shell, nothing anywhere else)
        # Measured in r05 on the rounds-only map too (1 voxel, x = 19: k_reseed_outside) -- so the allowance stands for both
        # engines, but it is PINNED TO THE SHELL: a voxel beyond the runs' own disagreement anywhere else fails.
        shell = 2 * n * n
        assert_envelope(rep, "after the window widened", farther_allow=max(rep["envelope"]["disagree"], shell // 200))
        if rep["envelope"]["farther"] > rep["envelope"]["disagree"]:
            out_x = np.asarray(rep["envelope"]["outside_idx"]) // (n * n)
            assert np.all(np.isin(out_x, (19, 20))), (out_x, rep["envelope"])
        # a fixed point of the reference's operator everywhere -- except, possibly, on that shell: a voxel the narrowed window
        # froze (as the reference freezes its own, :378) keeps its value until a wave passes again
        mi = np.asarray(rep["mismatch_idx"])
        frozen = rep["d2_mismatch"] <= len(mi) and np.all(np.isin(mi // (n * n), (19, 20)))
        assert rep["pair_violations"] == 0 or frozen, rep
    # everything deleted, one update without obstacles: the history is gone, the gate may open again
    occ = np.argwhere(maps[0].download_field(("occ",))["occ"].reshape((n,) * 3) == 1).astype(np.int32)
    t.cycles([], occ, 6)
    t.esdf()
    assert int(maps[0].download_field(("occ",))["occ"].sum()) == 0
    t.cycles(S[:50], [], 6)                                   # (six hits: these voxels sit at the lower clamp of the log-odds)
    assert maps[0].last_insert > 0
    assert t.esdf() is True
    assert t.same_fields() == 0
    assert_exact(compare_dense(maps[0], cpu.primary))
    for m in maps:
        m.close()
    cpu.close()


def test_loading_a_partially_observed_checkpoint_shuts_the_gate(hip_lib, oracle_libs, best_oracle_kind, tmp_path):
    """A map that is eligible (all observed) loads the checkpoint of one that is not (a block never observed, a voxel first
    observed while obstacles existed): observed count and the late-observation flag come from the file, the next update
    runs the rounds and equals the reference; loading the eligible checkpoint back re-opens the gate."""
    import fiesta_amd
    n, res = 32, 0.1
    size = (n * res,) * 3
    mk = lambda e: fiesta_amd.ESDFMap((0, 0, 0), res, size, update_engine=e)   # noqa: E731
    full, part = mk("bulk"), mk("bulk")
    cpu = EnvelopeOracle(lambda: oracle_libs.OracleMap((0, 0, 0), res, size, kind=best_oracle_kind), k=4)
    for m in (full, part, cpu):
        m.SetParameters(*P_DEFAULT)
        m.SetOriginalRange()
    g = all_voxels(n)
    hole = np.all((g >= 10) & (g < 18), axis=1)
    rng = np.random.RandomState(8)
    S = g[~hole][rng.choice((~hole).sum(), 120, replace=False)]
    # `full`: everything observed, obstacles, the transform
    full.SetOccupancy(g, 0, want_ret=False)
    full.UpdateOccupancy(True)
    full.UpdateESDF()
    for _ in range(3):
        full.SetOccupancy(S, 1, want_ret=False)
        full.UpdateOccupancy(True)
    assert full.UpdateESDF()["bulk"]
    # `part` and the oracle: a block never observed
    for m, f in ((part, lambda v, o: part.SetOccupancy(v, o, want_ret=False)), (cpu, cpu.SetOccupancyVox)):
        f(g[~hole], 0)
        m.UpdateOccupancy(True)
        m.UpdateESDF()
        for _ in range(3):
            f(S, 1)
            m.UpdateOccupancy(True)
    assert not part.UpdateESDF()["bulk"]
    cpu.UpdateESDF()
    p_full, p_part = str(tmp_path / "full.ckpt"), str(tmp_path / "part.ckpt")
    full.save(p_full)
    part.save(p_part)
    full.load(p_part)                                         # the eligible map becomes the partially observed one
    T = g[~hole][rng.choice((~hole).sum(), 40, replace=False)]
    for _ in range(3):
        for m, f in ((full, lambda v, o: full.SetOccupancy(v, o, want_ret=False)), (cpu, cpu.SetOccupancyVox)):
            f(T, 1)
            f(g[hole][:100], 0)                               # ... and part of the block is observed late
            m.UpdateOccupancy(True)
    st = full.UpdateESDF()
    cpu.UpdateESDF()
    assert not st["bulk"], st
    rep = compare_dense(full, cpu)
    assert_envelope(rep, "after loading the partially observed checkpoint")
    assert rep["pair_violations"] == 0, rep
    full.load(p_full)                                         # back: eligible again
    for _ in range(3):
        full.SetOccupancy(T, 1, want_ret=False)
        full.UpdateOccupancy(True)
    assert full.UpdateESDF()["bulk"]
    for m in (full, part):
        m.close()
    cpu.close()


def test_re_observation_does_not_inflate_the_observed_count(hip_lib, oracle_libs, best_oracle_kind):
    """C_OBSERVED counts FIRST observations: observing 7/8 of the map twice must not make it look fully observed."""
    n = 24
    maps, cpu, _ = _trio(oracle_libs, best_oracle_kind, n, envelope=4)
    t = Trio(maps, cpu)
    g = all_voxels(n)
    most = g[g[:, 0] < 21]
    for _ in range(3):                                        # 3 x 7/8 of the voxels > all voxels
        t.observe(most, 0)
        t.fuse()
    t.esdf()
    S = most[np.random.RandomState(1).choice(len(most), 80, replace=False)]
    t.cycles(S, [], 3)
    assert t.esdf() is False
    for m in maps:
        rep = compare_dense(m, cpu)
        assert_envelope(rep, "partially observed")
    t.observe(g[g[:, 0] >= 21], 0)                            # the rest, late: observed count complete, but stale voxels
    t.fuse()
    t.esdf()
    t.cycles(S[:10] + np.array([0, 1, 0], np.int32), [], 3)
    assert t.esdf() is False                                  # late observations still wait for their first wave
    for m in maps:
        rep = compare_dense(m, cpu)
        assert_envelope(rep, "late observation")
        m.close()
    cpu.close()


def test_one_ineligible_shard_keeps_the_whole_group_on_the_rounds(hip_lib, oracle_libs, best_oracle_kind):
    """Two shards, engine "bulk"; one voxel of shard 1 is never observed: the group's engine choice is one all-gather of
    every shard's eligibility -- exactly one "no" sends everybody through the frontier rounds."""
    from fiesta_amd.sharded import ShardedESDFMap
    from test_gpu_sharded import compare, drive
    gs, res = (48, 32, 32), 0.1
    sm = ShardedESDFMap((0, 0, 0), res, gs, 2, update_engine="bulk")
    cpu = oracle_libs.OracleMap((0, 0, 0), res, tuple((np.array(gs) - 0.5) * res), kind=best_oracle_kind)
    for m in (sm, cpu):
        m.SetParameters(*P_DEFAULT)
        m.SetOriginalRange()
    allv = all_voxels(gs)
    missing = np.all(allv == np.array([40, 3, 3]), axis=1)
    drive(sm, cpu, [([], allv[~missing], 1)])
    rng = np.random.RandomState(4)
    S = (rng.rand(120, 3) * gs).astype(np.int32)
    S = S[~np.all(S == np.array([40, 3, 3]), axis=1)]
    for occ, free in ((S, []), ((rng.rand(40, 3) * gs).astype(np.int32), S[:40])):
        for _ in range(6 if len(free) else 3):
            sm.SetOccupancy(occ, 1)
            cpu.SetOccupancyVox(occ, 1)
            if len(free):
                sm.SetOccupancy(free, 0)
                cpu.SetOccupancyVox(free, 0)
            assert sm.UpdateOccupancy(True) == cpu.UpdateOccupancy(True)
        st = sm.UpdateESDF()
        cpu.UpdateESDF()
        assert not st["bulk"], st
        f = sm.assemble()
        o = cpu.dump_dense()
        from scenarios import oracle_d2
        od2, _ = oracle_d2(o, gs)
        assert int((f["d2"].astype(np.int64) != od2).sum()) <= 4   # (one unobserved voxel: its neighbourhood is order-dependent)
    sm.close()
