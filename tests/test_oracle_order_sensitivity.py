"""How order-dependent is the reference algorithm itself?  (CPU, oracle only.)

The GPU engine cannot reproduce the reference's serial FIFO order, so on PARTIALLY observed maps -- where
the reference's fixed point depends on that order -- the parity tests allow a small d^2
mismatch budget.  This test pins that budget to a measurement instead of a guess: the observations of the
golden raycast fixture are replayed through SetOccupancy in the natural voxel order and in shuffled orders
(identical hit/miss counters, identical occupancy, only the queue order differs) and the distance fields of
the oracle are compared with each other.  On fully observed maps the same experiment changes no distance.
"""
import numpy as np

from golden_programs import SMALL_INTR
from scenarios import P_DEFAULT, all_voxels, depth_to_points, render_depth, yaw_pose


def _mk(oracle_libs, kind, origin, res, size):
    m = oracle_libs.OracleMap(origin, res, size, kind=kind)
    m.SetParameters(*P_DEFAULT)
    m.SetOriginalRange()
    return m


def test_partially_observed_distances_depend_on_queue_order(oracle_libs, best_oracle_kind):
    res, origin, size = 0.1, (-4.0, -4.0, -2.0), (8.0, 8.0, 4.0)
    src = _mk(oracle_libs, best_oracle_kind, origin, res, size)
    maps = [_mk(oracle_libs, best_oracle_kind, origin, res, size) for _ in range(3)]
    rng = np.random.RandomState(0)
    spheres = [((1.5, 0.5, 0.0), 0.6), ((-1.0, -1.5, -0.3), 0.5)]
    gs = src.grid_size
    worst = 0.0
    for f in range(3):
        T = yaw_pose(20.0 * f, (0.1 * f, -0.05 * f, 0.02))
        pts = depth_to_points(render_depth(T, rows=60, cols=80, spheres=spheres, intr=SMALL_INTR), intr=SMALL_INTR)
        src.raycast_frame(pts, T, T[:3, 3], 0.5, 5.0, origin, np.add(origin, size))
        hit, miss = src.dump_counts()
        src.UpdateOccupancy(True)
        src.UpdateESDF()
        idx = np.flatnonzero(miss > 0)
        vox = np.stack([idx // (gs[1] * gs[2]), (idx // gs[2]) % gs[1], idx % gs[2]], -1).astype(np.int32)
        for k, m in enumerate(maps):
            order = np.arange(len(idx)) if k == 0 else rng.permutation(len(idx))
            reps = miss[idx][order]
            v = np.repeat(vox[order], reps, 0)
            occ = np.concatenate([np.r_[np.ones(h, np.int32), np.zeros(t - h, np.int32)]
                                  for h, t in zip(hit[idx][order], reps)])
            m.SetOccupancyVox(v, occ)
            m.UpdateOccupancy(True)
            m.UpdateESDF()
        dumps = [m.dump_dense(("dist", "occ", "logodds")) for m in maps]
        for d in dumps[1:]:  # same counters -> same occupancy, bit for bit
            assert np.array_equal(d["occ"], dumps[0]["occ"]) and np.array_equal(d["logodds"], dumps[0]["logodds"])
        finite = int(((dumps[0]["dist"] >= 0) & (dumps[0]["dist"] < 10000)).sum())
        diffs = [int((d["dist"] != dumps[0]["dist"]).sum()) for d in dumps[1:]]
        diffs.append(int((src.dump_dense(("dist",))["dist"] != dumps[0]["dist"]).sum()))
        if finite:
            worst = max(worst, max(diffs) / finite)
            assert max(diffs) <= 0.03 * finite, (f, finite, diffs)
    assert 0.002 < worst < 0.03, worst  # measured 0.8-1.6 % with the verbatim reference


def test_fully_observed_distances_do_not_depend_on_queue_order(oracle_libs, best_oracle_kind):
    n, res = 32, 0.1
    maps = [_mk(oracle_libs, best_oracle_kind, (0, 0, 0), res, (n * res,) * 3) for _ in range(3)]
    rng = np.random.RandomState(4)
    S = rng.randint(0, n, (400, 3)).astype(np.int32)
    ties = 0
    for k, m in enumerate(maps):
        m.SetOccupancyVox(all_voxels(m.grid_size), 0)
        m.UpdateOccupancy(True)
        m.UpdateESDF()
        order = np.arange(len(S)) if k == 0 else rng.permutation(len(S))
        for _ in range(3):
            m.SetOccupancyVox(S[order], 1)
            m.UpdateOccupancy(True)
        m.UpdateESDF()
    d0 = maps[0].dump_dense(("dist", "coc"))
    for m in maps[1:]:
        d = m.dump_dense(("dist", "coc"))
        assert np.array_equal(d["dist"], d0["dist"])          # distances: order-independent
        ties += int((d["coc"] != d0["coc"]).any(-1).sum())    # ids: NOT order-independent
    assert ties > 0


def test_fully_observed_surface_scene_is_not_an_exact_transform_and_depends_on_order(oracle_libs, best_oracle_kind):
    """... but "order-independent" stops at scatter scenes.  With obstacles on SURFACES (what a depth sensor produces,
    bench.py --scene surfaces) the reference's 24-neighbour vector propagation misses the exact Euclidean transform on a
    few voxels per 10^5 even on a fully observed map -- always on the far side, by up to half a voxel -- and which
    voxels depends on the order of the inserts inside ONE batch.  This is why the GPU parity contract at full size
    (tests/test_gpu_full_size.py, tests/golden/c2_512_*_digest.npz) reads: equal to the reference wherever the reference
    is the exact transform, exact where it is not."""
    import os
    import sys
    from scipy import ndimage
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import Workload
    n, res = 128, 0.1
    V = Workload(n, 3000, scene="surfaces").initial()
    rng = np.random.RandomState(1)
    fields = []
    for k in range(2):
        m = _mk(oracle_libs, best_oracle_kind, (0, 0, 0), res, ((n - 0.5) * res,) * 3)
        m.SetOccupancyVox(all_voxels(m.grid_size), 0)
        m.UpdateOccupancy(True)
        m.UpdateESDF()
        order = np.arange(len(V)) if k == 0 else rng.permutation(len(V))
        for _ in range(3):
            m.SetOccupancyVox(V[order], 1)
            m.UpdateOccupancy(True)
        m.UpdateESDF()
        d = m.dump_dense(("dist", "occ"))
        fields.append(np.rint((d["dist"] / res) ** 2).astype(np.int64))
        occ = d["occ"]
        m.close()
    idx = ndimage.distance_transform_edt(occ.reshape(n, n, n) == 0, return_distances=False, return_indices=True)
    ax = np.arange(n)
    exact = ((idx[0] - ax[:, None, None]) ** 2 + (idx[1] - ax[None, :, None]) ** 2 + (idx[2] - ax[None, None, :]) ** 2).reshape(-1)
    far = [int((f > exact).sum()) for f in fields]
    assert all(int((f < exact).sum()) == 0 for f in fields)      # never closer than the truth
    assert all(0 < x < 2e-4 * n ** 3 for x in far), far           # a few voxels per 10^5 are farther
    assert int((fields[0] != fields[1]).sum()) > 0                # and WHICH ones depends on the insert order
    worst = max(float((np.sqrt(f) - np.sqrt(exact)).max()) for f in fields)
    assert worst <= 0.75, worst                                   # by a fraction of a voxel


def test_order_envelope_of_the_reference_contains_one_more_of_its_runs(oracle_libs, best_oracle_kind):
    """The parity contract of the GPU tests on partially observed maps (scenarios.EnvelopeOracle / assert_envelope),
    exercised on the CPU with the reference itself as the engine under test: K + 1 shuffled runs span an interval per
    voxel; ONE MORE run (another order: what any engine with its own processing order amounts to) lies inside it except
    on a handful of voxels -- far fewer than the runs disagree on among themselves, which is the allowance the contract
    grants.  Also: the committed envelope of the golden raycast program contains the restatement's shuffled run."""
    import os
    from golden_programs import PROGRAMS
    from scenarios import EnvelopeOracle, assert_envelope, d2_from_dist
    res = 0.1
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    fixture, env = np.load(os.path.join(gold, "raycast_frames.npz")), np.load(os.path.join(gold, "raycast_frames_envelope.npz"))
    seen = 0

    def make(o, r, s):   # primary: the recorded order; companions: 5 more orders; the LAST companion plays the engine
        return EnvelopeOracle(lambda: oracle_libs.OracleMap(o, r, s, kind=best_oracle_kind), k=6, seed=99)
    for cp, m, _ in PROGRAMS["raycast_frames"](make):
        engine = m.companions.pop()                       # judged against the other 6 runs
        gd2 = d2_from_dist(engine.dump_dense(("dist",))["dist"], res)
        rep = m.judge(gd2)
        m.companions.append(engine)
        assert rep["runs"] == 6
        assert_envelope(rep, cp)
        if rep["finite"]:
            assert rep["disagree"] > 20 and rep["outside"] * 3 <= rep["disagree"], rep
            assert max(rep["leave_one_out"]) * 3 <= rep["disagree"], rep
            seen += 1
        # ... and against the committed 10-run envelope of the verbatim reference
        lo = d2_from_dist(fixture[f"{cp}/dist"], res)
        hi = lo.copy()
        lo[env[f"{cp}/idx"]], hi[env[f"{cp}/idx"]] = env[f"{cp}/lo"], env[f"{cp}/hi"]
        outside = int(((gd2 < lo) | (gd2 > hi)).sum())
        assert outside <= len(env[f"{cp}/idx"]), (cp, outside)
    assert seen == 2
