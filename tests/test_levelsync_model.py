"""The GPU level engine's SCHEDULE against the reference's own order spread -- without a GPU.

oracle/esdf_port.cpp carries, next to the restatement of the reference's FIFO (relax), a CPU model of what
fiesta_amd/csrc/level_kernels.hpp does (relax_levels: the FIFO's layers as levels -- every entry of a level pulls from the
field as the level found it, then the ones that did not improve push, a minimum per target; orphans of a delete reset and
pulled; the rule for orphans outside the update window).  Here that model is the "engine under test" and the judge is
scenarios.EnvelopeOracle over the verbatim reference (the restatement where /root/reference is absent): K shuffled replays
of the same observations.  What is asserted is the STRICT contract of scenarios.assert_envelope -- outside the envelope, on
either side, on at most as many voxels as the reference's own runs disagree on; equal to the reference on fully observed
maps -- i.e. the contract the `-m gpu` suite holds the HIP engine to, checked here on the schedule itself so that a change
of the schedule shows up on the CPU-only tier.  The scenarios are those of tests/test_gpu_dense_parity.py,
test_gpu_raycast_parity.py and test_gpu_fuzz.py.
"""
import numpy as np
import pytest

from scenarios import (P_DEFAULT, EnvelopeOracle, all_voxels, assert_envelope, d2_from_dist, depth_to_points, render_depth,
                       yaw_pose)


class Pair:
    """The level-schedule model and the envelope of the reference, driven by identical calls."""

    def __init__(self, oracle_libs, kind, origin, res, size, k, schedule=6):
        # schedule 6: levels + the delete drain's list walk in shells (what the GPU runs on whole-map updates whose level 0
        # holds at most 8192 entries, level_kernels.hpp: k_level_fill); schedule 1: levels only (larger level 0, partial windows)
        self.eng = oracle_libs.OracleMap(origin, res, size, kind="port")
        self.eng.set_schedule(schedule)
        self.env = EnvelopeOracle(lambda: oracle_libs.OracleMap(origin, res, size, kind=kind), k=k)
        self.res = res
        for m in (self.eng, self.env):
            m.SetParameters(*P_DEFAULT)
            m.SetOriginalRange()

    def both(self, f):
        f(self.eng)
        f(self.env)

    def observe(self, vox, occ):
        self.both(lambda m: m.SetOccupancyVox(vox, occ))

    def fuse(self, g=True):
        a, b = self.eng.UpdateOccupancy(g), self.env.UpdateOccupancy(g)
        assert a == b and (self.eng.last_insert, self.eng.last_delete) == (self.env.last_insert, self.env.last_delete)

    def esdf(self):
        self.eng.UpdateESDF()
        self.env.UpdateESDF()

    def cycles(self, occ_vox, free_vox, n, g=True):
        for _ in range(n):
            if len(occ_vox):
                self.observe(occ_vox, 1)
            if len(free_vox):
                self.observe(free_vox, 0)
            self.fuse(g)

    def judge(self, mask=None):
        d2 = d2_from_dist(self.eng.dump_dense(("dist",))["dist"], self.res)
        return self.env.judge(d2, mask=mask)


def size_of(n, res):
    return tuple(np.asarray(n if not np.isscalar(n) else (n, n, n)) * res)


def test_depth_frames(oracle_libs, best_oracle_kind):
    """test_gpu_raycast_parity.py: test_frames_counts_exact_with_reference_dedup, the first frames."""
    origin, size, res = (-6.4, -6.4, -3.2), (12.75, 12.75, 6.35), 0.1
    p = Pair(oracle_libs, best_oracle_kind, origin, res, size, k=4)
    lc, rc = origin, tuple(np.array(origin) + np.array(size))
    spheres = [((1.5, 0.5, 0.0), 0.5), ((-1.0, 2.0, 0.3), 0.7), ((0.5, -2.0, -0.5), 0.4)]
    pos = np.array([0.13, -0.21, 0.05])
    intr = dict(fx=96.1, fy=96.1, cx=80.7, cy=58.9)
    worst = 0
    for f in range(4):
        T = yaw_pose(20.0 * f, pos + 0.05 * f)
        pts = depth_to_points(render_depth(T, rows=120, cols=160, spheres=spheres, intr=intr), intr=intr)
        pts[::501] = np.nan
        o = T[:3, 3]
        p.both(lambda m: m.raycast_frame(pts, T, o, 0.5, 5.0, lc, rc))
        p.fuse()
        p.esdf()
        env = p.judge()
        assert_envelope(env, f"frame {f}", strict=True)
        assert env["inf_where_every_run_is_finite"] == 0 and env["finite_where_every_run_is_inf"] == 0, env
        # like one more run of the reference: not outside the envelope of the others on more voxels than its own runs are
        assert env["outside"] <= max(env["leave_one_out"]) + env["disagree"] // 4, env
        worst = max(worst, env["outside"])
    assert p.eng.levels_run > 20


def test_fragmentary_observation(oracle_libs, best_oracle_kind):
    """test_gpu_dense_parity.py: test_occupancy_fusion_logodds_and_positions (random positions, ~30 % observed)."""
    n = 24
    p = Pair(oracle_libs, best_oracle_kind, (-3.0, -3.0, -1.0), 0.25, size_of(n, 0.25), k=6)
    rng = np.random.RandomState(5)
    for cycle in range(6):
        pos = np.array([-3.0, -3.0, -1.0]) + (rng.rand(5000, 3) * 1.2 - 0.1) * n * 0.25
        occ = (rng.rand(5000) < 0.45).astype(np.int32)
        occ[::97] = 2
        p.both(lambda m: m.SetOccupancyPos(pos, occ))
        p.fuse()
        p.esdf()
        assert_envelope(p.judge(), f"cycle {cycle}", strict=True)


def test_unobserved_blocks_and_late_observation(oracle_libs, best_oracle_kind):
    """test_gpu_dense_parity.py: test_partial_observation_frontier_semantics."""
    n = 40
    p = Pair(oracle_libs, best_oracle_kind, (0, 0, 0), 0.1, size_of(n, 0.1), k=5)
    rng = np.random.RandomState(3)
    g = all_voxels(p.eng.grid_size)
    blocks = rng.rand(n // 4 + 1, n // 4 + 1, n // 4 + 1) > 0.27
    keep = blocks[g[:, 0] // 4, g[:, 1] // 4, g[:, 2] // 4]
    p.observe(g[keep], 0)
    p.fuse()
    p.esdf()
    S = g[keep][rng.choice(keep.sum(), 300, replace=False)]
    p.cycles(S, [], 3)
    p.esdf()
    assert_envelope(p.judge(), "inserts into a partially observed map", strict=True)
    p.observe(g[~keep], 0)
    p.fuse()
    p.esdf()
    assert_envelope(p.judge(), "late observation", strict=True)
    p.cycles(rng.randint(0, n, (50, 3)).astype(np.int32), [], 3)
    p.esdf()
    assert_envelope(p.judge(), "wave through late observations", strict=True)


def test_local_sliding_window(oracle_libs, best_oracle_kind):
    """test_gpu_dense_parity.py: test_local_sliding_window_mode -- inside AND outside the window under the same contract
    (orphans outside the window: the rule of level_kernels.hpp / k_level_outside)."""
    n = 48
    p = Pair(oracle_libs, best_oracle_kind, (0, 0, 0), 0.1, size_of(n, 0.1), k=6)
    gs = p.eng.grid_size
    p.observe(all_voxels(gs), 0)
    p.fuse()
    p.esdf()
    rng = np.random.RandomState(21)
    S = rng.randint(4, n - 4, (250, 3)).astype(np.int32)
    p.cycles(S, [], 3)
    p.esdf()
    assert p.judge()["vs_primary"] == 0          # fully observed: the reference's field
    V = all_voxels(gs)
    for step in range(3):
        c = np.array([1.2 + 0.5 * step, 2.0, 2.4])
        lo, hi = c - [1.5, 1.5, 1.0], c + [1.5, 1.5, 1.0]
        p.both(lambda m: m.SetUpdateRange(lo, hi))
        new = (c / 0.1 + rng.randint(-12, 12, (60, 3))).astype(np.int32)
        gone = S[rng.choice(len(S), 40, replace=False)]
        p.cycles(new, gone, 6, g=False)
        p.esdf()
        wlo, whi = np.floor(lo / 0.1).astype(int), np.floor((hi - 0.05) / 0.1).astype(int)
        inside = np.all((V >= wlo) & (V <= whi), axis=1)
        e_in, e_out = p.judge(mask=inside), p.judge(mask=~inside)
        spread = e_in["disagree"] + e_out["disagree"]
        assert_envelope(e_in, f"step {step} inside", farther_allow=spread, strict=True)
        assert e_in["closer"] <= spread
        assert_envelope(e_out, f"step {step} outside", farther_allow=spread, strict=True)
        # which orphans out there get a value at all follows the reference's list walk closely (the rounds engine: hundreds)
        assert e_out["inf_where_every_run_is_finite"] + e_out["finite_where_every_run_is_inf"] <= spread, e_out


@pytest.mark.parametrize("seed", [41, 44, 47])
def test_random_boxes(oracle_libs, best_oracle_kind, seed):
    """test_gpu_fuzz.py: test_random_sequences_partially_observed."""
    rng = np.random.RandomState(seed)
    dims = tuple(int(v) for v in rng.randint(20, 40, 3))
    p = Pair(oracle_libs, best_oracle_kind, (0.0, 0.0, 0.0), 0.1, tuple((np.array(dims) - 0.5) * 0.1), k=5)
    for step in range(6):
        c0 = np.array([rng.randint(0, d - 8) for d in dims])
        ext = rng.randint(6, 16, 3)
        box = all_voxels(tuple(int(v) for v in ext)) + c0.astype(np.int32)
        box = box[np.all(box < np.array(dims), axis=1)]
        occ = box[rng.rand(len(box)) < 0.02]
        p.cycles(occ, box, 3)
        p.esdf()
        assert_envelope(p.judge(), f"step {step}", strict=True)


@pytest.mark.parametrize("seed", [11, 17, 23])
def test_fully_observed_equals_the_reference(oracle_libs, best_oracle_kind, seed):
    """test_gpu_fuzz.py: test_random_sequences_fully_observed -- mixed inserts and deletes, d^2 equal on every voxel."""
    rng = np.random.RandomState(seed)
    dims = tuple(int(v) for v in rng.randint(9, 44, 3))
    res = float(rng.choice([0.05, 0.1, 0.25]))
    p = Pair(oracle_libs, best_oracle_kind, tuple(float(v) for v in rng.uniform(-3, 3, 3)), res,
             tuple((np.array(dims) - 0.5) * res), k=1)
    p.observe(all_voxels(dims), 0)
    p.fuse()
    p.esdf()
    live = np.zeros((0, 3), np.int32)
    for step in range(5):
        n_new = int(rng.randint(1, 60))
        new = np.stack([rng.randint(-2, dims[k] + 2, n_new) for k in range(3)], -1).astype(np.int32)
        gone = live[rng.rand(len(live)) < 0.3]
        p.cycles(new, gone, int(rng.choice([3, 6])))
        p.esdf()
        env = p.judge()
        assert env["vs_primary"] == 0 and env["disagree"] == 0, env
        ok = np.all((new >= 0) & (new < np.array(dims)), axis=1)
        keep = set(map(tuple, live.tolist())) - set(map(tuple, gone.tolist())) | set(map(tuple, new[ok].tolist()))
        live = np.array(sorted(keep), np.int32).reshape(-1, 3)


@pytest.mark.parametrize("schedule, last", [(1, (44, 0, 0)), (6, (0, 0, 0))], ids=["levels-only", "with-the-list-walk"])
def test_hash_fuzz_seed_63_is_a_property_of_the_schedule(oracle_libs, best_oracle_kind, schedule, last):
    """tests/test_gpu_fuzz.py: test_random_sequences_hash_map[63], on the hash-block flavour of the model and the reference.
    The one state of the GPU suite where the level engine of round 4's first half left the envelope although the reference's
    shuffled runs all agree: the model of that schedule (1: orphans wait in level 0 for their first pull) leaves it on exactly
    the same 44 voxels (all closer, all above the exact distance) -- the schedule, not the device.  With the delete drain's
    list walk ahead of level 0 (6: the dead cells fill from their rims inwards; k_level_fill on the GPU) they
    are gone.  Both pinned so that a change of either shows."""
    from scenarios import D2_INF, hash_key
    kind = best_oracle_kind if oracle_libs.available(best_oracle_kind, "hash") else "port"
    rng = np.random.RandomState(63)
    origin, res = tuple(float(v) for v in rng.uniform(-1, 1, 3)), float(rng.choice([0.05, 0.1]))
    rng.choice([0, 1000, 50000])
    eng = oracle_libs.OracleMap(origin, res, reserve_size=1000, mode="hash", kind="port")
    eng.set_schedule(schedule)
    cpu = EnvelopeOracle(lambda: oracle_libs.OracleMap(origin, res, reserve_size=1000, mode="hash", kind=kind), k=6)
    for m in (eng, cpu):
        m.SetParameters(*P_DEFAULT)
        m.SetOriginalRange()
    centre = rng.randint(-30, 30, 3)
    live = np.zeros((0, 3), np.int32)
    outside = []
    for step in range(5):
        centre = centre + rng.randint(-6, 7, 3)
        ext = rng.randint(8, 22, 3)
        box = (all_voxels(tuple(int(v) for v in ext)) + (centre - ext // 2)).astype(np.int32)
        new = box[rng.rand(len(box)) < 0.01]
        gone = live[rng.rand(len(live)) < 0.4]
        for k in range(3):
            if k == 0:
                eng.SetOccupancyVox(box, 0)
                cpu.SetOccupancyVox(box, 0)
            for vv, o in ((new, 1), (gone, 0)):
                if len(vv):
                    eng.SetOccupancyVox(vv, o)
                    cpu.SetOccupancyVox(vv, o)
            assert eng.UpdateOccupancy(True) == cpu.UpdateOccupancy(True)
        eng.UpdateESDF()
        cpu.UpdateESDF()
        d = eng.dump_hash()
        ok = d["vox"][:, 0] != -10000
        keys, d2 = hash_key(d["vox"][ok]), d2_from_dist(d["dist"][ok], res)
        o = np.argsort(keys)
        env = cpu.judge(d2[o], keys=keys[o])
        outside.append((env["closer"], env["farther"], env["disagree"]))
        if env["closer"]:   # never below the exact distance to the nearest occupied voxel
            D = cpu._fields(keys[o])
            bad = np.flatnonzero(d2[o] < D.min(0))
            pd = cpu.primary.dump_hash()
            occ = pd["vox"][(pd["vox"][:, 0] != -10000) & (pd["occ"] == 1)].astype(np.int64)
            vox = d["vox"][ok][o].astype(np.int64)
            exact = np.array([((occ - vox[i]) ** 2).sum(-1).min() for i in bad])
            assert np.all(d2[o][bad] >= exact) and np.all(D.min(0)[bad] - d2[o][bad] <= 16)   # (by at most ~0.6 voxel in distance)
        live = np.concatenate([live, new])
        rng.uniform(-25, 25, (150, 3))
    assert outside[:4] == [(0, 0, 0)] * 4 and outside[4] == last, outside


def test_wide_mixed_delta_seed_9_is_a_property_of_the_schedule(oracle_libs, best_oracle_kind):
    """tests/test_gpu_level_grid.py: test_grid_and_launch_pairs_run_the_same_schedule -- 400 obstacles into a map with a
    quarter of its 4^3 blocks unobserved, then 200 of them deleted and 200 new ones in ONE update.  The second state is the
    other place of the GPU suite where the level engine is outside the strict contract (42 closer, 4 farther where the
    reference's runs disagree on 13): the model gives exactly those numbers -- the schedule, not the device.
    (Level 0 of that update holds far more than 8192 entries: the GPU runs it without the list walk, schedule 1; with it the
    model says 43 / 0 / 13 -- the far side goes, the near side has another cause.)"""
    n, res = (64, 64, 48), 0.1
    p = Pair(oracle_libs, best_oracle_kind, (0, 0, 0), res, size_of(n, res), k=4, schedule=1)
    rng = np.random.RandomState(9)
    gs = np.array(p.eng.grid_size)
    g = all_voxels(p.eng.grid_size)
    blocks = rng.rand(*(gs // 4 + 1)) > 0.25
    g = g[blocks[g[:, 0] // 4, g[:, 1] // 4, g[:, 2] // 4]]
    p.observe(g, 0)
    p.fuse()
    p.esdf()
    S = g[rng.choice(len(g), 400, replace=False)]
    p.cycles(S, [], 3)
    p.esdf()
    e = p.judge()
    assert_envelope(e, "400 inserts", strict=True)
    T = g[rng.choice(len(g), 200, replace=False)]
    p.cycles(T, S[:200], 6)
    p.esdf()
    e = p.judge()
    assert (e["closer"], e["farther"], e["disagree"]) == (42, 4, 13), e
