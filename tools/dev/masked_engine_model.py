"""dev, CPU only: numpy model of the MASKED transform engine judged against the reference's own order spread.

The engine itself is tests/masked_model.py (what the GPU tests compare the kernels with); this script drives it on bench.py's
C2-partial scenario next to K + 1 runs of the verbatim reference.  Variants tried on the way are switches of the model
("nokeep": uncertified voxels start from "no obstacle" instead of their old value; "nomask": every obstacle is a site; block=1:
plain Jacobi) -- a second certificate through "portals" was modelled here too and dropped (wrong on 70 of 11.8 M voxels).
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


from masked_model import certificate, effective_sites, masked_engine  # noqa: E402,F401
import masked_model  # noqa: E402


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
    for a in sys.argv:
        if a.startswith("block="):
            masked_model.BLOCK_SUBITERS = int(a[6:])
    run(G, K, steps)
