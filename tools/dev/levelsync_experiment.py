"""CPU experiment (no GPU): is a level-synchronous schedule inside the reference's own order envelope?

Engine under test = the port with oracle_set_schedule(k) (the model of the GPU's level engine, oracle/esdf_port.cpp:
relax_levels); judge = scenarios.EnvelopeOracle over the verbatim reference (K shuffled replays).  Scenarios are the ones
of tests/test_gpu_dense_parity.py / test_gpu_raycast_parity.py that run on partially observed maps.
usage: python tools/dev/levelsync_experiment.py [schedule=1] [K=8]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle  # noqa: E402
from scenarios import (P_DEFAULT, EnvelopeOracle, all_voxels, d2_from_dist, depth_to_points, render_depth,  # noqa: E402
                       yaw_pose)

SCHED = 1  # (argv[1] is kept for old command lines: the experimental schedules are gone, 1 = the level engine's model)
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
KIND = "ref" if pyoracle.available("ref") else "port"


def short(env):
    return {k: env[k] for k in ("finite", "disagree", "outside", "outside_where_runs_agree", "closer", "farther",
                                "inf_where_every_run_is_finite", "finite_where_every_run_is_inf", "vs_primary")} | {"loo_max": max(env["leave_one_out"])}


class Pair:
    def __init__(self, origin, res, size, k=K):
        self.eng = pyoracle.OracleMap(origin, res, size, kind="port")
        self.eng.set_schedule(SCHED)
        self.env = EnvelopeOracle(lambda: pyoracle.OracleMap(origin, res, size, kind=KIND), k=k)
        self.res = res
        for m in (self.eng, self.env):
            m.SetParameters(*P_DEFAULT)
            m.SetOriginalRange()

    def both(self, f):
        f(self.eng)
        f(self.env)

    def observe(self, vox, occ):
        self.both(lambda m: m.SetOccupancyVox(vox, occ))

    def observe_pos(self, pos, occ):
        self.both(lambda m: m.SetOccupancyPos(pos, occ))

    def fuse(self, g=True):
        a, b = self.eng.UpdateOccupancy(g), self.env.UpdateOccupancy(g)
        assert a == b and (self.eng.last_insert, self.eng.last_delete) == (self.env.last_insert, self.env.last_delete)

    def esdf(self):
        self.eng.UpdateESDF()
        self.env.UpdateESDF()

    def make_occupied(self, v, cycles=3):
        for _ in range(cycles):
            self.observe(v, 1)
            self.fuse()

    def make_free(self, v, cycles=6):
        for _ in range(cycles):
            self.observe(v, 0)
            self.fuse()

    def mixed(self, occ_vox, free_vox, cycles=6, g=True):
        for _ in range(cycles):
            if len(occ_vox):
                self.observe(occ_vox, 1)
            if len(free_vox):
                self.observe(free_vox, 0)
            self.fuse(g)

    def judge(self, what, mask=None):
        d2 = d2_from_dist(self.eng.dump_dense(("dist",))["dist"], self.res)
        env = self.env.judge(d2, mask=mask)
        print(f"  {what}: {short(env)}", flush=True)
        return env


def size_of(n, res):
    return tuple(np.asarray(n if not np.isscalar(n) else (n, n, n)) * res)


def s_frames():
    print("depth frames 128x128x64")
    origin, size, res = (-6.4, -6.4, -3.2), (12.75, 12.75, 6.35), 0.1
    p = Pair(origin, res, size, k=min(K, 5))
    lc, rc = origin, tuple(np.array(origin) + np.array(size))
    spheres = [((1.5, 0.5, 0.0), 0.5), ((-1.0, 2.0, 0.3), 0.7), ((0.5, -2.0, -0.5), 0.4)]
    pos = np.array([0.13, -0.21, 0.05])
    intr = dict(fx=96.1, fy=96.1, cx=80.7, cy=58.9)
    for f in range(6):
        T = yaw_pose(20.0 * f, pos + 0.05 * f)
        pts = depth_to_points(render_depth(T, rows=120, cols=160, spheres=spheres, intr=intr), intr=intr)
        pts[::501] = np.nan
        o = T[:3, 3]
        p.both(lambda m: m.raycast_frame(pts, T, o, 0.5, 5.0, lc, rc))
        p.fuse()
        p.esdf()
        p.judge(f"frame {f}")


def s_fusion():
    print("random positions, 30 % observed (n=24)")
    n = 24
    p = Pair((-3.0, -3.0, -1.0), 0.25, size_of(n, 0.25), k=6)
    rng = np.random.RandomState(5)
    for cycle in range(8):
        pos = np.array([-3.0, -3.0, -1.0]) + (rng.rand(5000, 3) * 1.2 - 0.1) * n * 0.25
        occ = (rng.rand(5000) < 0.45).astype(np.int32)
        occ[::97] = 2
        p.observe_pos(pos, occ)
        p.fuse()
        p.esdf()
        p.judge(f"cycle {cycle}")


def s_partial():
    print("partial observation frontier semantics (n=40)")
    n = 40
    p = Pair((0, 0, 0), 0.1, size_of(n, 0.1), k=6)
    rng = np.random.RandomState(3)
    g = all_voxels((n, n, n))
    blocks = rng.rand(n // 4 + 1, n // 4 + 1, n // 4 + 1) > 0.27
    keep = blocks[g[:, 0] // 4, g[:, 1] // 4, g[:, 2] // 4]
    p.observe(g[keep], 0)
    p.fuse()
    p.esdf()
    S = g[keep][rng.choice(keep.sum(), 300, replace=False)]
    p.make_occupied(S)
    p.esdf()
    p.judge("inserts")
    p.observe(g[~keep], 0)
    p.fuse()
    p.esdf()
    p.judge("late observation")
    p.make_occupied(rng.randint(0, n, (50, 3)).astype(np.int32))
    p.esdf()
    p.judge("wave through late observations")


def s_window():
    print("local sliding window (n=48)")
    n = 48
    p = Pair((0, 0, 0), 0.1, size_of(n, 0.1), k=K)
    p.observe(all_voxels(p.eng.grid_size), 0)
    p.fuse()
    p.esdf()
    rng = np.random.RandomState(21)
    S = rng.randint(4, n - 4, (250, 3)).astype(np.int32)
    p.make_occupied(S)
    p.esdf()
    p.judge("fully observed inserts")
    V = all_voxels(p.eng.grid_size)
    for step in range(4):
        c = np.array([1.2 + 0.5 * step, 2.0, 2.4])
        lo, hi = c - [1.5, 1.5, 1.0], c + [1.5, 1.5, 1.0]
        p.both(lambda m: m.SetUpdateRange(lo, hi))
        new = (c / 0.1 + rng.randint(-12, 12, (60, 3))).astype(np.int32)
        gone = S[rng.choice(len(S), 40, replace=False)]
        p.mixed(new, gone, 6, g=False)
        p.esdf()
        wlo, whi = np.floor(lo / 0.1).astype(int), np.floor((hi - 0.05) / 0.1).astype(int)
        inside = np.all((V >= wlo) & (V <= whi), axis=1)
        p.judge(f"step {step} inside", mask=inside)
        p.judge(f"step {step} outside", mask=~inside)


def s_window_delete():
    print("window then delete with dependents outside (n=40)")
    n = 40
    res = 0.1
    p = Pair((0, 0, 0), res, size_of(n, res), k=6)
    p.observe(all_voxels((n, n, n)), 0)
    p.fuse()
    p.esdf()
    rng = np.random.RandomState(5)
    S = rng.randint(2, n - 2, (120, 3)).astype(np.int32)
    p.make_occupied(S)
    p.esdf()
    p.judge("inserts")
    p.both(lambda m: m.SetUpdateRange((0.0, 0.0, 0.0), (1.8, n * res, n * res)))
    inside = S[S[:, 0] < 17]
    p.make_free(inside[:25])
    p.esdf()
    p.judge("delete with orphans outside the window")
    p.both(lambda m: m.SetUpdateRange((1.0, 0.0, 0.0), (n * res, n * res, n * res)))
    outside = S[S[:, 0] >= 20]
    p.mixed(rng.randint(22, n - 2, (10, 3)).astype(np.int32), outside[:20])
    p.esdf()
    p.judge("window moved over the former outside")


def s_full():
    print("fully observed insert/delete (n=48, (37,50,70))")
    for n in (48, (37, 50, 70)):
        p = Pair((0, 0, 0), 0.1, size_of(n, 0.1), k=3)
        gs = p.eng.grid_size
        p.observe(all_voxels(gs), 0)
        p.fuse()
        p.esdf()
        rng = np.random.RandomState(7)
        dims = np.array(gs)
        S = (rng.randint(0, 1 << 20, (300, 3)) % dims).astype(np.int32)
        p.make_occupied(S)
        p.esdf()
        p.judge("insert")
        p.mixed((rng.randint(0, 1 << 20, (100, 3)) % dims).astype(np.int32), S[:150])
        p.esdf()
        p.judge("mixed")


def s_fuzz_partial():
    print("fuzz: random boxes, partially observed (seeds 41-48)")
    for seed in range(41, 49):
        rng = np.random.RandomState(seed)
        dims = tuple(int(v) for v in rng.randint(20, 40, 3))
        p = Pair((0.0, 0.0, 0.0), 0.1, tuple((np.array(dims) - 0.5) * 0.1), k=6)
        for step in range(6):
            c0 = np.array([rng.randint(0, d - 8) for d in dims])
            ext = rng.randint(6, 16, 3)
            box = all_voxels(tuple(int(v) for v in ext)) + c0.astype(np.int32)
            box = box[np.all(box < np.array(dims), axis=1)]
            occ = box[rng.rand(len(box)) < 0.02]
            for _ in range(3):
                p.observe(box, 0)
                if len(occ):
                    p.observe(occ, 1)
                p.fuse()
            p.esdf()
            p.judge(f"seed {seed} step {step}")


def s_fuzz_full():
    print("fuzz: fully observed (seeds 11-26), must equal the primary")
    for seed in range(11, 27):
        rng = np.random.RandomState(seed)
        dims = tuple(int(v) for v in rng.randint(9, 44, 3))
        res = float(rng.choice([0.05, 0.1, 0.25]))
        origin = tuple(float(v) for v in rng.uniform(-3, 3, 3))
        p = Pair(origin, res, tuple((np.array(dims) - 0.5) * res), k=2)
        p.observe(all_voxels(dims), 0)
        p.fuse()
        p.esdf()
        lo, hi = np.zeros(3, int), np.array(dims)
        live = np.zeros((0, 3), np.int32)
        worst = 0
        for step in range(int(rng.randint(5, 9))):
            n_new = int(rng.randint(1, 60))
            new = np.stack([rng.randint(lo[k] - 2, hi[k] + 2, n_new) for k in range(3)], -1).astype(np.int32)
            gone = live[rng.rand(len(live)) < 0.3]
            cycles = int(rng.choice([1, 3, 3, 6]))
            for _ in range(cycles):
                if rng.rand() < 0.5:
                    p.observe(new, 1)
                else:
                    pos = (new + 0.5 + rng.uniform(-0.3, 0.3, new.shape)) * res + np.array(origin)
                    p.observe_pos(pos, 1)
                if len(gone):
                    p.observe(gone, 0)
                if rng.rand() < 0.3:
                    p.observe(np.concatenate([new[: n_new // 2], new[: n_new // 3]]), int(rng.rand() < 0.5))
                p.fuse()
                if rng.rand() < 0.25:
                    p.esdf()
            p.esdf()
            d2 = d2_from_dist(p.eng.dump_dense(("dist",))["dist"], res)
            env = p.env.judge(d2)
            worst = max(worst, env["vs_primary"])
            inside = np.all((new >= lo) & (new < hi), axis=1)
            keep = set(map(tuple, live.tolist())) - set(map(tuple, gone.tolist())) | set(map(tuple, new[inside].tolist()))
            live = np.array(sorted(keep), np.int32).reshape(-1, 3)
            rng.uniform(-1.0, 1.0, (200, 3))
        print(f"  seed {seed}: worst vs_primary {worst}", flush=True)


if __name__ == "__main__":
    which = sys.argv[3].split(",") if len(sys.argv) > 3 else ["frames", "fusion", "partial", "window", "window_delete", "full"]
    for w in which:
        globals()["s_" + w]()
