"""dev, CPU only: numpy model of the MASKED transform engine judged against the reference's own order spread.

The engine for large deltas on partially observed maps:
  1. T = exact feature transform of the occupied set (what the cell transform / envelope passes write), masked to observed voxels;
  2. certificate: an observed voxel keeps T iff the samples of its straight segment to its winner are all observed
     (n = 2 max|d| + 1 steps, sample i at v + round(d i / n));
  3. every other observed voxel is reset to "no obstacle" and repaired by Jacobi pulls over the 24-stencil
     (src/ESDFMap.cpp:349-367's pull, strict <, stencil order) until nothing changes.
Judged exactly as the GPU tests judge an engine: tests/scenarios.py EnvelopeOracle (K + 1 runs of the verbatim reference in
shuffled first-touch order) -> closer / farther than every run vs the number of voxels the runs disagree on.

    python tools/dev/masked_engine_model.py [grid] [K] [steps]
"""
import json
import os
import sys
import time

import numpy as np
from scipy import ndimage

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle  # noqa: E402
from scenarios import D2_INF, DIRS24, EnvelopeOracle, P_DEFAULT, d2_from_dist  # noqa: E402

sys.path.insert(0, ROOT)


RING = {k: np.array([(x, y, z) for x in range(-2, 3) for y in range(-2, 3) for z in range(-2, 3) if x * x + y * y + z * z == k]) for k in (1, 2, 4)}


def certificate(obs, V, S, idx=None, Sreq=None):
    """V, S: (m, 3) voxels and the segments' far ends.  True where every sample of the segment is observed -- and, with idx
    (the transform's winner per voxel) and Sreq, where every sample's winner is Sreq."""
    d = (S - V).astype(np.int64)
    n = 2 * np.abs(d).max(1) + 1
    cert = np.ones(len(V), bool)
    done = np.zeros(len(V), bool)   # the walk reached a stencil neighbour of the winner: the winner pushes there itself
    relaxed = "relaxcert" in sys.argv
    for i in range(0, int(n.max())):
        act = cert & ~done & (i < n)
        if not act.any():
            break
        ai = np.flatnonzero(act)
        na = n[act][:, None]
        p = V[act] + (2 * d[act] * i + na) // (2 * na)
        r = np.abs(S[act] - p)
        near = (r.sum(1) <= 2) & (r.max(1) <= 2) & ~((r.max(1) == 2) & (r.sum(1) != 2)) if relaxed else (r.sum(1) == 0)
        # (24-stencil: one axis +-1, two axes +-1, one axis +-2; or the winner itself)
        ok = obs[p[:, 0], p[:, 1], p[:, 2]]
        if idx is not None:
            ok &= np.all(np.stack([idx[k][p[:, 0], p[:, 1], p[:, 2]] for k in range(3)], 1) == Sreq[act], axis=1)
        cert[ai[~ok]] = False
        done[ai[near & ok]] = True
    return cert


def effective_sites(occ, obs):
    """obstacles with at least one OBSERVED stencil neighbour: the others can never hand their id to anybody"""
    G = occ.shape
    P = np.pad(obs, 2)
    any_n = np.zeros(G, bool)
    for e in DIRS24:
        any_n |= P[2 + e[0]:2 + e[0] + G[0], 2 + e[1]:2 + e[1] + G[1], 2 + e[2]:2 + e[2] + G[2]]
    return occ & any_n


def masked_engine(occ, obs, W_old=None, keep_old=True, mask_sites=True):
    """occ, obs: bool (G, G, G); W_old: the engine's own previous field (winner coordinates, -1 none).
    Returns d2 (int64; -1 unobserved, D2_INF none), W, stats."""
    G = occ.shape
    eff = effective_sites(occ, obs) if mask_sites else occ
    idx = ndimage.distance_transform_edt(~eff, return_distances=False, return_indices=True)
    g = np.meshgrid(*[np.arange(k) for k in G], indexing="ij")
    V = np.argwhere(obs)
    S = np.stack([idx[k][obs] for k in range(3)], 1)
    cert = certificate(obs, V, S) if eff.any() else np.zeros(len(V), bool)
    n_first = int((~cert).sum())
    if "portal" in sys.argv and eff.any():
        # second chance: the winner s hands its id to an observed stencil neighbour a (a "portal") whose own winner is s, and
        # the segment v -> a is observed
        Gs = np.array(G)
        for e in DIRS24:
            todo = np.flatnonzero(~cert)
            if not len(todo):
                break
            Sv = S[todo]
            A = Sv + e
            ok = np.all((A >= 0) & (A < Gs), axis=1)
            Ac = np.where(ok[:, None], A, 0)
            ok &= obs[Ac[:, 0], Ac[:, 1], Ac[:, 2]]
            ok &= np.all(np.stack([idx[k][Ac[:, 0], Ac[:, 1], Ac[:, 2]] for k in range(3)], 1) == Sv, axis=1)
            # ... strictly: no other effective site as near to a as s is (a tie at the only way out: the reference may hold the other)
            de = int((e ** 2).sum())
            Pe = np.pad(eff, 2)
            rivals = np.zeros(len(todo), np.int64)
            for r in RING[de]:
                rivals += Pe[Ac[:, 0] + 2 + r[0], Ac[:, 1] + 2 + r[1], Ac[:, 2] + 2 + r[2]]
            ok &= rivals == 1
            # a must not be farther from v than s is (the id travels towards v)
            ok &= ((A - V[todo]) ** 2).sum(1) < ((Sv - V[todo]) ** 2).sum(1)
            sub = todo[ok]
            if len(sub):
                c2 = certificate(obs, V[sub], Ac[ok], idx, Sv[ok]) if "loosepath" not in sys.argv else certificate(obs, V[sub], Ac[ok])
                cert[sub[c2]] = True
    # field: winner coordinates per voxel; -1 = none
    W = np.full(G + (3,), -1, np.int64)
    W[V[cert, 0], V[cert, 1], V[cert, 2]] = S[cert]
    U = V[~cert]
    if keep_old and W_old is not None:  # an uncertified voxel keeps what it held if that obstacle is still there
        o = W_old[U[:, 0], U[:, 1], U[:, 2]]
        oc = np.where(o >= 0, o, 0)
        alive = (o[:, 0] >= 0) & occ[oc[:, 0], oc[:, 1], oc[:, 2]]
        W[U[alive, 0], U[alive, 1], U[alive, 2]] = o[alive]
    O = np.argwhere(occ)
    W[O[:, 0], O[:, 1], O[:, 2]] = O  # an obstacle is its own closest obstacle
    U = U[~occ[U[:, 0], U[:, 1], U[:, 2]]]
    iters = 0
    nx, ny, nz = G
    while len(U):
        iters += 1
        cur = W[U[:, 0], U[:, 1], U[:, 2]]
        best = np.where(cur[:, 0] >= 0, ((U - cur) ** 2).sum(1), D2_INF)
        bw = cur.copy()
        for e in DIRS24:
            N = U + e
            ok = np.all((N >= 0) & (N < np.array(G)), axis=1)
            Nc = np.where(ok[:, None], N, 0)
            w = W[Nc[:, 0], Nc[:, 1], Nc[:, 2]]
            has = ok & (w[:, 0] >= 0) & obs[Nc[:, 0], Nc[:, 1], Nc[:, 2]]
            cand = np.where(has, ((U - w) ** 2).sum(1), D2_INF)
            take = cand < best
            best = np.where(take, cand, best)
            bw[take] = w[take]
        changed = (bw != cur).any(1)
        if not changed.any():
            break
        W[U[:, 0], U[:, 1], U[:, 2]] = bw
    d2 = np.where(W[..., 0] >= 0, ((np.stack(g, -1) - W) ** 2).sum(-1), D2_INF)
    d2 = np.where(obs, d2, -1)
    return d2.astype(np.int64), W, {"observed": int(obs.sum()), "uncertified": int((~cert).sum()), "uncertified_first_test": n_first, "jacobi_iterations": iters,
                                    "isolated_obstacles": int((occ & ~eff).sum())}


def run(G, K, steps, unobserved=0.27):
    sys.path.insert(0, ROOT)
    import bench
    res = 0.1
    keep = np.random.RandomState(2718).rand(G // 32, G // 32, G // 32) >= unobserved
    obs0 = np.repeat(np.repeat(np.repeat(keep, 32, 0), 32, 1), 32, 2)
    kind = "ref" if pyoracle.available("ref", "array") else "port"
    env = EnvelopeOracle(lambda: pyoracle.OracleMap((0, 0, 0), res, (G * res,) * 3, kind=kind), k=K)
    env.SetParameters(*P_DEFAULT)
    env.SetOriginalRange()
    V = np.argwhere(obs0).astype(np.int32)
    env.primary.SetOccupancyVox(V, 0)
    env.UpdateOccupancy(True)
    env.UpdateESDF()
    n_obs = max(2, int(round(50000 * (G / 512.0) ** 3)))
    w = bench.Workload(G, n_obs, seed=12345)
    for _ in range(3):
        env.primary.SetOccupancyVox(w.initial(), 1)
        env.UpdateOccupancy(True)
    env.UpdateESDF()
    rows = []
    state = {"W": None}

    def judge(what):
        d = env.primary.dump_dense(("dist", "occ"))
        occ = d["occ"].reshape(G, G, G) != 0
        obs = d["dist"].reshape(G, G, G) >= 0
        t0 = time.time()
        d2, state["W"], st = masked_engine(occ, obs, state["W"], keep_old="nokeep" not in sys.argv, mask_sites="nomask" not in sys.argv)
        rep = env.judge(d2.reshape(-1))
        rep.pop("outside_idx")
        rep.update(st, what=what, model_s=round(time.time() - t0, 1), obstacles=int(occ.sum()))
        rows.append(rep)
        print(json.dumps(rep), flush=True)

    judge("scatter insert")
    for s in range(steps):
        new, old = w.next_step()
        for c in range(3):
            env.primary.SetOccupancyVox(new, 1)
            if c == 2:
                env.primary.SetOccupancyVox(old, 0)
            env.UpdateOccupancy(True)
        env.UpdateESDF()
        judge(f"step {s + 1}")
    env.close()
    return rows


if __name__ == "__main__":
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    run(G, K, steps)
