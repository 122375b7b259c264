"""dev, CPU only: a design study for the general engine on partially observed maps.

Question.  On a partially observed map the reference's field is not the exact transform T of the occupied set (propagation
only passes through observed voxels) -- but where IS it?  Candidate rule: a voxel v whose straight segment to its nearest
obstacle s(v) runs through observed voxels only ("certified") holds exactly T(v) in the reference, whatever the queue order.
If that holds and most voxels are certified, a large delta on such a map could be served by the cell transform (0.3 ms) with
the frontier rounds repairing only the shadows -- instead of the rounds recomputing everything (8 ms on C2-partial).

What this script measures, on the verbatim reference (oracle/_ref) at C2's obstacle density with 27 % of the map in
never-observed 32^3 blocks: the fraction of observed voxels that is certified, how many certified voxels differ from T in
the reference (the rule's error), and how many uncertified ones equal T anyway (the rule's pessimism).
    python tools/dev/masked_transform_study.py [grid] [seeds]
"""
import json
import os
import sys

import numpy as np
from scipy import ndimage

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402

P_DEFAULT = (0.70, 0.35, 0.12, 0.97, 0.80)


def study(G, seed, kind, delta=False):
    res = 0.1
    rng = np.random.RandomState(seed)
    keep = rng.rand(G // 32, G // 32, G // 32) >= 0.27
    obs = np.repeat(np.repeat(np.repeat(keep, 32, 0), 32, 1), 32, 2)
    m = pyoracle.OracleMap((0, 0, 0), res, (G * res,) * 3, kind=kind)
    m.SetParameters(*P_DEFAULT)
    m.SetOriginalRange()
    V = np.argwhere(obs).astype(np.int32)
    for s in range(0, len(V), 1 << 20):
        m.SetOccupancyVox(V[s:s + (1 << 20)], 0)
    m.UpdateOccupancy(True)
    m.UpdateESDF()
    n_obs = int(round(3.7e-4 * G ** 3))
    S = V[rng.choice(len(V), n_obs, replace=False)]
    for _ in range(3):
        m.SetOccupancyVox(S, 1)
        m.UpdateOccupancy(True)
    m.UpdateESDF()
    if delta:  # config 2's steady-state step: half of the obstacles replaced in ONE UpdateESDF
        new = V[rng.choice(len(V), n_obs // 2, replace=False)]
        for c in range(6):
            m.SetOccupancyVox(new, 1)
            m.SetOccupancyVox(S[: n_obs // 2], 0)
            m.UpdateOccupancy(True)
        m.UpdateESDF()
    d = m.dump_dense(("dist", "occ"))
    dist = d["dist"].reshape(G, G, G)
    occ = d["occ"].reshape(G, G, G) != 0
    m.close()
    # the reference's squared voxel distance (dist is in metres; unobserved -10000, no obstacle +10000)
    finite = obs & (dist >= 0) & (dist < 9999)
    F = np.where(finite, np.rint((dist / res) ** 2), -1).astype(np.int64)
    idx = ndimage.distance_transform_edt(~occ, return_distances=False, return_indices=True)
    g = np.meshgrid(*[np.arange(G)] * 3, indexing="ij")
    T = sum((idx[k] - g[k]) ** 2 for k in range(3)).astype(np.int64)
    # certification: samples along the segment v -> s(v), two per voxel of its longest axis
    sel = np.argwhere(obs)
    v = sel.astype(np.float64)
    s = np.stack([idx[k][obs] for k in range(3)], 1).astype(np.float64)
    n = (2 * np.abs(s - v).max(1)).astype(np.int64) + 1
    cert = np.ones(len(sel), bool)
    for i in range(1, int(n.max()) + 1):
        act = cert & (i <= n)
        if not act.any():
            break
        t = (i / n[act])[:, None]
        p = np.rint(v[act] + (s[act] - v[act]) * t).astype(np.int64)
        ok = obs[p[:, 0], p[:, 1], p[:, 2]]
        cert[np.flatnonzero(act)[~ok]] = False
    Fo, To = F[obs], T[obs]
    out = {
        "grid": G, "seed": seed, "after": "insert + mixed delta" if delta else "insert", "observed_voxels": int(obs.sum()), "obstacles": int(occ.sum()),
        "reference_equals_T": float((Fo == To).mean()),
        "certified": float(cert.mean()),
        "certified_but_reference_differs": int((cert & (Fo != To)).sum()),
        "uncertified": int((~cert).sum()),
        "uncertified_yet_reference_equals_T": float(((~cert) & (Fo == To)).sum() / max(1, (~cert).sum())),
        "reference_never_reached": int((obs & ~finite).sum()),
    }
    return out


if __name__ == "__main__":
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    kind = "ref" if pyoracle.available("ref", "array") else "port"
    rows = [study(G, 100 + k, kind, delta) for k in range(seeds) for delta in (False, True)]
    print(json.dumps({"oracle": kind, "runs": rows}, indent=1))
