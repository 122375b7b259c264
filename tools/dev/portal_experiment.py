"""dev, CPU only: would a second certificate through PORTALS pay?  (modelled in r06, not shipped.)

A winner s hidden behind one unobserved voxel hands its id to an observed stencil neighbour p (a portal) and to nobody else; every
voxel that prefers s then fails the straight-segment certificate and is repaired by propagation from p -- on config 2's partially
observed scene four fifths of the repair set.  Portal certificate: v keeps T(v) = s if some stencil neighbour p of s is observed,
nearer to v than s, is the only site's... (strict: no other site within |p - s| of p), the discrete segment v -> p is observed AND
every voxel on it has s as its own winner.  This script runs tests/masked_model.py with and without it on the dumps
tests/golden/make_golden_c2_partial.py leaves in --tmp, against the committed envelope of the reference's runs.
    python tools/dev/portal_experiment.py /tmp/c2p256 256
"""
import os
import sys
import time

import numpy as np
from scipy import ndimage

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import masked_model as mm  # noqa: E402
from scenarios import D2_INF, DIRS24  # noqa: E402

RING = {k: np.array([(x, y, z) for x in range(-2, 3) for y in range(-2, 3) for z in range(-2, 3) if 0 < x * x + y * y + z * z <= k]) for k in (1, 2, 4)}


def path_ok(obs, idx, V, P, S):
    """segment V -> P observed and every sample's winner == S"""
    d = (P - V).astype(np.int64)
    n = 2 * np.abs(d).max(1) + 1
    ok = np.ones(len(V), bool)
    for i in range(1, int(n.max())):
        act = ok & (i < n)
        if not act.any():
            break
        na = n[act][:, None]
        p = V[act] + (2 * d[act] * i + na) // (2 * na)
        good = obs[p[:, 0], p[:, 1], p[:, 2]]
        good &= np.all(np.stack([idx[k][p[:, 0], p[:, 1], p[:, 2]] for k in range(3)], 1) == S[act], axis=1)
        ok[np.flatnonzero(act)[~good]] = False
    return ok


def portal_certificate(occ, obs, eff, idx, V, S):
    G = np.array(occ.shape)
    cert = np.zeros(len(V), bool)
    Pe = np.pad(eff, 2)
    for e in DIRS24:
        todo = np.flatnonzero(~cert)
        A = S[todo] + e
        ok = np.all((A >= 0) & (A < G), axis=1)
        Ac = np.where(ok[:, None], A, 0)
        ok &= obs[Ac[:, 0], Ac[:, 1], Ac[:, 2]] & ~occ[Ac[:, 0], Ac[:, 1], Ac[:, 2]]
        ok &= ((A - V[todo]) ** 2).sum(1) < ((S[todo] - V[todo]) ** 2).sum(1)
        de = int((e ** 2).sum())
        sub = np.flatnonzero(ok)
        if not len(sub):
            continue
        riv = np.zeros(len(sub), np.int64)   # sites within |e|^2 of the portal (the winner itself is one of them)
        for r in RING[de]:
            riv += Pe[Ac[sub, 0] + 2 + r[0], Ac[sub, 1] + 2 + r[1], Ac[sub, 2] + 2 + r[2]]
        sub = sub[riv == 1]
        if not len(sub):
            continue
        good = path_ok(obs, idx, V[todo][sub], Ac[sub], S[todo][sub])
        cert[todo[sub[good]]] = True
    return cert


def engine(occ, obs, W_old, portals):
    """masked_model.masked_engine with the optional second certificate (a copy of its first half)"""
    G = occ.shape
    eff = mm.effective_sites(occ, obs)
    idx = ndimage.distance_transform_edt(~eff, return_distances=False, return_indices=True)
    V = np.argwhere(obs)
    S = np.stack([idx[k][obs] for k in range(3)], 1)
    cert = mm.certificate(obs, V, S)
    n1 = int((~cert).sum())
    if portals:
        u = np.flatnonzero(~cert)
        cert[u[portal_certificate(occ, obs, eff, idx, V[u], S[u])]] = True
    return cert, V, S, n1, eff, idx


if __name__ == "__main__":
    tmp, G = sys.argv[1], int(sys.argv[2])
    gold = np.load(os.path.join(ROOT, "tests", "golden", f"c2_partial_{G}_envelope.npz"))
    for portals in (False, True):
        W = None
        for cp in ("scatter", "step"):
            occ = np.load(os.path.join(tmp, f"{cp}_occ.npy")).reshape(G, G, G) != 0
            obs = (np.load(os.path.join(tmp, f"r0_{cp}_d2.npy")) >= 0).reshape(G, G, G)
            t0 = time.time()
            if not portals:
                d2, W, st = mm.masked_engine(occ, obs, W)
            else:
                # monkey-patch the certificate for this call
                orig = mm.certificate
                eff = mm.effective_sites(occ, obs)
                idx = ndimage.distance_transform_edt(~eff, return_distances=False, return_indices=True)

                def both(obs_, V, S, _o=orig, occ_=occ, eff_=eff, idx_=idx):
                    c = _o(obs_, V, S)
                    u = np.flatnonzero(~c)
                    c[u[portal_certificate(occ_, obs_, eff_, idx_, V[u], S[u])]] = True
                    return c
                mm.certificate = both
                d2, W, st = mm.masked_engine(occ, obs, W)
                mm.certificate = orig
            g = d2.reshape(-1)
            T = np.full(G ** 3, 0, np.int32)
            # the envelope: T_eff except on the listed voxels -- judged on the listed voxels and, elsewhere, against T itself
            eff = mm.effective_sites(occ, obs)
            idx = ndimage.distance_transform_edt(~eff, return_distances=False, return_indices=True)
            gg = np.meshgrid(*[np.arange(G, dtype=np.int32)] * 3, indexing="ij")
            T = np.where(obs, sum((idx[k].astype(np.int64) - gg[k]) ** 2 for k in range(3)), -1).astype(np.int32).reshape(-1)
            lo, hi = T.copy(), T.copy()
            e = gold[f"{cp}/exc_idx"].astype(np.int64)
            lo[e], hi[e] = gold[f"{cp}/exc_lo"], gold[f"{cp}/exc_hi"]
            print({"portals": portals, "cp": cp, "marked": st["uncertified"], "iterations": st["jacobi_iterations"], "closer": int((g < lo).sum()),
                   "farther": int((g > hi).sum()), "disagree": int(gold[f"{cp}/disagree"]), "seconds": round(time.time() - t0)}, flush=True)
