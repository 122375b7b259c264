"""tests only: numpy model of the MASKED transform (fiesta_amd/csrc/mask_kernels.hpp) -- the engine for large
deltas on partially observed maps.  Same arithmetic, same schedule: the GPU field equals this model voxel for voxel
(tests/test_gpu_masked.py), and the model is what was judged against the verbatim reference's order spread on the CPU before
any kernel was written (tools/dev/masked_engine_model.py).

  1. sites = the obstacles with at least one OBSERVED stencil neighbour (the others can hand their id to nobody,
     src/ESDFMap.cpp:375-391); T = exact feature transform of the sites;
  2. an observed voxel keeps T iff every voxel of its discrete segment to the winner is observed (certificate) -- or, second
     chance, iff the winner has a PORTAL (portal_certificate) the voxel's way to is clear;
  3. every other observed voxel keeps what it held before the update if that obstacle still exists, else "no obstacle", and is
     repaired by 24-neighbour pulls (:349-367: stencil order, strict <) -- block Jacobi: per global iteration every 8^3 cell
     runs up to BLOCK_SUBITERS steps on its own voxels against the other cells as the iteration found them.
"""
import numpy as np
from scipy import ndimage

from scenarios import D2_INF, DIRS24

BLOCK_SUBITERS = 6   # cell-local Jacobi steps per global iteration (mask_kernels.hpp: kMaskSub); 1 = plain Jacobi
def sensor_mask(G, poses=6, seed=31415):
    """A sensor-shaped observed set for the parity fixtures (tests/golden/make_golden_c2_partial.py --pattern sensor): the union of
    `poses` view cones -- apex and axis random, half-angle 33 degrees, range 0.62 G -- so that the boundary of the observed space
    cuts through cells and words everywhere (the 32^3-block pattern of bench.py --unobserved never does)."""
    rng = np.random.RandomState(seed)
    ax = np.arange(G, dtype=np.float32)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    obs = np.zeros((G, G, G), bool)
    for _ in range(poses):
        apex = rng.uniform(0.15 * G, 0.85 * G, 3).astype(np.float32)
        d = rng.normal(size=3)
        d = (d / np.linalg.norm(d)).astype(np.float32)
        rx, ry, rz = X - apex[0], Y - apex[1], Z - apex[2]
        along = rx * d[0] + ry * d[1] + rz * d[2]
        r2 = rx * rx + ry * ry + rz * rz
        obs |= (along > 0) & (along * along >= np.float32(np.cos(np.deg2rad(33.0)) ** 2) * r2) & (r2 <= np.float32((0.62 * G) ** 2))
    return obs


# the six voxels between two voxels a (1, 1, 1)-diagonal apart, as choices of the axes that have already moved
HALFWAY = np.array([(1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1)])


def certificate(obs, V, S):
    """V, S: (m, 3) voxels and their winners.  True where every sample of the discrete segment is observed: n = 2 max|d| + 1
    steps, sample i at v + round(d i / n), i = 1 .. n - 1 (mask_kernels.hpp: mask_segment_observed) -- and where two consecutive
    samples differ along ALL THREE axes, one of the six voxels between them is observed as well: the stencil the reference
    propagates along (24 directions, src/ESDFMap.cpp:33-60) has no (1, 1, 1) step, an id crosses such a diagonal in two hops."""
    d = (S - V).astype(np.int64)
    n = 2 * np.abs(d).max(1) + 1
    cert = np.ones(len(V), bool)
    prev = V.astype(np.int64).copy()
    for i in range(1, int(n.max()) + 1):
        act = cert & (i <= n)   # (sample n is the winner itself: the last step may be a diagonal too)
        if not act.any():
            break
        ai = np.flatnonzero(act)
        na = n[act][:, None]
        p = V[act] + (2 * d[act] * i + na) // (2 * na)
        ok = obs[p[:, 0], p[:, 1], p[:, 2]]
        step = p - prev[act]
        tri = np.flatnonzero(ok & (step != 0).all(1))
        if len(tri):
            q = prev[act][tri][:, None, :] + HALFWAY[None] * step[tri][:, None, :]
            ok[tri] = obs[q[..., 0], q[..., 1], q[..., 2]].any(1)
        cert[ai[~ok]] = False
        prev[act] = p
    return cert


def certificate_by_cells(cellobs, V, S):
    """The same answer from the 8^3 cells a segment crosses (mask_kernels.hpp: mask_segment_cells), for maps whose cells are all
    either fully observed (cellobs 1) or never observed (0): axis a of sample i is v + sign floor((2 |d_a| i + n) / 2n), so the
    walk enters its k-th voxel along a at sample ceil(n (2k - 1) / (2 |d_a|)); the cell boundaries are taken in the order of those
    samples, the axes that cross at the same sample together.  One voxel pair at a time (a model of the kernel's loop, not fast)."""
    out = np.ones(len(V), bool)
    for j, (v, s) in enumerate(zip(np.asarray(V, np.int64), np.asarray(S, np.int64))):
        d = s - v
        a = np.abs(d)
        n = 2 * int(a.max()) + 1
        sg = np.where(d < 0, -1, 1)
        k = np.where(d < 0, (v & 7) + 1, 8 - (v & 7))
        INF = 1 << 60
        t = [(-(-(n * (2 * int(k[x]) - 1)) // (2 * int(a[x]))) if k[x] <= a[x] else INF) for x in range(3)]
        c = v >> 3
        assert cellobs[c[0], c[1], c[2]] == 1
        while True:
            tm = min(t)
            if tm == INF:
                break
            assert tm <= n - 1
            if t[0] == tm and t[1] == tm and t[2] == tm:   # through a cell's corner: one of the six other cells around it observed
                if not any(cellobs[c[0] + sg[0] * h[0], c[1] + sg[1] * h[1], c[2] + sg[2] * h[2]] for h in HALFWAY):
                    out[j] = False
                    break
            for x in range(3):
                if t[x] == tm:
                    c[x] += sg[x]
                    k[x] += 8
                    t[x] = -(-(n * (2 * int(k[x]) - 1)) // (2 * int(a[x]))) if k[x] <= a[x] else INF
            if cellobs[c[0], c[1], c[2]] == 0:
                out[j] = False
                break
    return out


RING = {k: np.array([(x, y, z) for x in range(-2, 3) for y in range(-2, 3) for z in range(-2, 3) if 0 < x * x + y * y + z * z <= k]) for k in (1, 2, 4)}


def path_in_cell(occ, obs, idx, V, P, S):
    """every sample of the discrete segment V -> P observed, free, and with S as its own winner; where two consecutive samples differ
    along all three axes, one of the six voxels between them as well, or that voxel is S itself (the stencil has no (1, 1, 1) step:
    the id crosses in two hops -- mask_kernels.hpp: mask_path_diagonals)"""
    d = (P - V).astype(np.int64)
    n = 2 * np.abs(d).max(1) + 1
    ok = np.ones(len(V), bool)
    prev = V.astype(np.int64).copy()

    def good_at(p, Sa):
        g = obs[p[..., 0], p[..., 1], p[..., 2]] & ~occ[p[..., 0], p[..., 1], p[..., 2]]
        return g & np.all(np.stack([idx[k][p[..., 0], p[..., 1], p[..., 2]] for k in range(3)], -1) == Sa, axis=-1)

    for i in range(1, int(n.max())):
        act = ok & (i < n)
        if not act.any():
            break
        p = V[act] + (2 * d[act] * i + n[act][:, None]) // (2 * n[act][:, None])
        good = good_at(p, S[act])
        step = p - prev[act]
        tri = np.flatnonzero(good & (step != 0).all(1))
        if len(tri):
            q = prev[act][tri][:, None, :] + HALFWAY[None] * step[tri][:, None, :]
            Sq = S[act][tri][:, None, :]
            good[tri] = (good_at(q, Sq) | (q == Sq).all(-1)).any(1)
        ok[np.flatnonzero(act)[~good]] = False
        prev[act] = p
    return ok


def portal_certificate(occ, obs, eff, idx, V, S):
    """The second certificate (mask_kernels.hpp: k_mask_walk): a winner s hidden behind an unobserved voxel hands its id to an
    observed stencil neighbour p -- a portal -- and through it to everybody whose way to p is clear.  Only a HIDDEN winner has
    portals (one of its 26 neighbours inside the grid was never observed).  v keeps T(v) = s if, for some
    stencil direction e (in stencil order), p = s + e is inside the grid, observed, free, nearer to v than s, has NO other site
    within |e| of it (then p can only hold s: s pushes it there itself, src/ESDFMap.cpp:375-391), and every voxel of the discrete
    segment v -> p is observed, free and has s as its own winner."""
    G = np.array(occ.shape)
    cert = np.zeros(len(V), bool)
    Pe = np.pad(eff, 2)
    # only a HIDDEN winner has portals: one with a never-observed voxel among its 26 neighbours (inside the grid)
    Po = np.pad(obs, 1, constant_values=True)
    hidden = np.zeros(occ.shape, bool)
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                hidden |= ~Po[1 + dx:1 + dx + G[0], 1 + dy:1 + dy + G[1], 1 + dz:1 + dz + G[2]]
    cand = hidden[S[:, 0], S[:, 1], S[:, 2]]
    for e in DIRS24:
        todo = np.flatnonzero(~cert & cand)
        if not len(todo):
            break
        A = S[todo] + e
        ok = np.all((A >= 0) & (A < G), axis=1)
        Ac = np.where(ok[:, None], A, 0)
        ok &= obs[Ac[:, 0], Ac[:, 1], Ac[:, 2]] & ~occ[Ac[:, 0], Ac[:, 1], Ac[:, 2]]
        ok &= ((A - V[todo]) ** 2).sum(1) < ((S[todo] - V[todo]) ** 2).sum(1)
        sub = np.flatnonzero(ok)
        if not len(sub):
            continue
        riv = np.zeros(len(sub), np.int64)   # sites within |e|^2 of the portal (the winner itself is one of them)
        for r in RING[int((e ** 2).sum())]:
            riv += Pe[Ac[sub, 0] + 2 + r[0], Ac[sub, 1] + 2 + r[1], Ac[sub, 2] + 2 + r[2]]
        sub = sub[riv == 1]
        if not len(sub):
            continue
        good = path_in_cell(occ, obs, idx, V[todo][sub], Ac[sub], S[todo][sub])
        cert[todo[sub[good]]] = True
    return cert


def effective_sites(occ, obs):
    """obstacles with at least one OBSERVED stencil neighbour: the others can never hand their id to anybody"""
    G = occ.shape
    P = np.pad(obs, 2)
    any_n = np.zeros(G, bool)
    for e in DIRS24:
        any_n |= P[2 + e[0]:2 + e[0] + G[0], 2 + e[1]:2 + e[1] + G[1], 2 + e[2]:2 + e[2] + G[2]]
    return occ & any_n


def masked_engine(occ, obs, W_old=None, keep_old=True, mask_sites=True, subiters=None, portals=True):
    """occ, obs: bool (G, G, G); W_old: the engine's own previous field (winner coordinates, -1 none).
    Returns d2 (int64; -1 unobserved, D2_INF none), W, stats."""
    G = occ.shape
    eff = effective_sites(occ, obs) if mask_sites else occ
    idx = ndimage.distance_transform_edt(~eff, return_distances=False, return_indices=True)
    g = np.meshgrid(*[np.arange(k) for k in G], indexing="ij")
    V = np.argwhere(obs)
    S = np.stack([idx[k][obs] for k in range(3)], 1)
    cert = certificate(obs, V, S) if eff.any() else np.zeros(len(V), bool)
    n_straight = int((~cert).sum())
    if portals and eff.any():
        u = np.flatnonzero(~cert & ~occ[V[:, 0], V[:, 1], V[:, 2]])
        cert[u[portal_certificate(occ, obs, eff, idx, V[u], S[u])]] = True
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
    Gs = np.array(G)
    L = BLOCK_SUBITERS if subiters is None else subiters  # cell-local Jacobi steps per global iteration (1: plain Jacobi); halos frozen at the iteration's start
    qid = lambda P: ((P[:, 0] >> 3) * 4096 + (P[:, 1] >> 3)) * 4096 + (P[:, 2] >> 3)   # noqa: E731  (the block: a cell of 8^3 voxels)
    qU = qid(U) if len(U) else None
    while len(U):
        iters += 1
        snap = W.copy() if L > 1 else W
        any_change = False
        for sub in range(L):
            cur = W[U[:, 0], U[:, 1], U[:, 2]]
            best = np.where(cur[:, 0] >= 0, ((U - cur) ** 2).sum(1), D2_INF)
            bw = cur.copy()
            for e in DIRS24:
                N = U + e
                ok = np.all((N >= 0) & (N < Gs), axis=1)
                Nc = np.where(ok[:, None], N, 0)
                w = W[Nc[:, 0], Nc[:, 1], Nc[:, 2]]
                if L > 1:
                    same = qid(Nc) == qU
                    w = np.where(same[:, None], w, snap[Nc[:, 0], Nc[:, 1], Nc[:, 2]])
                has = ok & (w[:, 0] >= 0) & obs[Nc[:, 0], Nc[:, 1], Nc[:, 2]]
                cand = np.where(has, ((U - w) ** 2).sum(1), D2_INF)
                take = cand < best
                best = np.where(take, cand, best)
                bw[take] = w[take]
            changed = (bw != cur).any(1)
            if not changed.any():
                break
            any_change = True
            W[U[:, 0], U[:, 1], U[:, 2]] = bw
        if not any_change:
            break
    d2 = np.where(W[..., 0] >= 0, ((np.stack(g, -1) - W) ** 2).sum(-1), D2_INF)
    d2 = np.where(obs, d2, -1)
    return d2.astype(np.int64), W, {"observed": int(obs.sum()), "uncertified": int((~cert).sum()), "uncertified_by_the_straight_segment": n_straight, "jacobi_iterations": iters,
                                    "isolated_obstacles": int((occ & ~eff).sum())}


