"""dev, GPU: the masked transform (update_engine "masked") on bench.py's C2-partial scenario at a size the verbatim reference
handles in seconds, judged against the K-run envelope of the reference (tests/scenarios.py) and, voxel by voxel, against the
numpy model of the same engine (tools/dev/masked_engine_model.py).

    python tools/dev/masked_gpu_check.py [grid] [K] [steps] [engine]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools", "dev"))
import fiesta_amd  # noqa: E402
from oracle import pyoracle  # noqa: E402
from scenarios import Both, EnvelopeOracle, P_DEFAULT, compare_dense  # noqa: E402
import bench  # noqa: E402
import masked_model as model  # noqa: E402


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    engine = sys.argv[4] if len(sys.argv) > 4 else "masked"
    with_model = "nomodel" not in sys.argv
    res = 0.1
    keep = np.random.RandomState(2718).rand(G // 32, G // 32, G // 32) >= 0.27
    kind = "ref" if pyoracle.available("ref", "array") else "port"
    env = EnvelopeOracle(lambda: pyoracle.OracleMap((0, 0, 0), res, (G * res,) * 3, kind=kind), k=K)
    gpu = fiesta_amd.ESDFMap((0, 0, 0), res, (G * res,) * 3, update_engine=engine)
    b = Both(gpu, env)
    b.params()
    gpu.SetOriginalRange()
    env.SetOriginalRange()
    obs0 = np.repeat(np.repeat(np.repeat(keep, 32, 0), 32, 1), 32, 2)
    b.observe(np.argwhere(obs0).astype(np.int32), 0)
    b.fuse()
    b.esdf()
    n_obs = max(2, int(round(50000 * (G / 512.0) ** 3)))
    w = bench.Workload(G, n_obs, seed=12345)
    for _ in range(3):
        b.observe(w.initial(), 1)
        b.fuse()
    state = {"W": None}

    def judge(what, sg):
        rep = compare_dense(gpu, env, check_logodds=False)
        e = rep["envelope"]
        out = {"what": what, "engine": {k: sg[k] for k in ("masked", "bulk", "cells", "levels", "rounds", "mask_uncertified", "mask_iterations", "mask_walks", "mask_quads", "nn_failed")}, "changed": sg["prof"],
               "ms": {k: round(sg[k], 4) for k in ("host_ms", "device_ms", "nn_cells_ms", "nn_lists_ms", "nn_fill_ms", "mask_certify_ms", "mask_repair_ms")},
               "finite": e["finite"], "disagree": e["disagree"], "closer": e["closer"], "farther": e["farther"], "vs_primary": e["vs_primary"],
               "pair_violations": rep["pair_violations"], "leave_one_out": e["leave_one_out"]}
        if with_model:
            d = env.primary.dump_dense(("dist", "occ"))
            occ = d["occ"].reshape(G, G, G) != 0
            obs = d["dist"].reshape(G, G, G) >= 0
            d2m, state["W"], st = model.masked_engine(occ, obs, state["W"])
            g = gpu.download_field()["d2"].astype(np.int64).reshape(G, G, G)
            out["gpu_vs_model"] = int((g != d2m).sum())
            out["model"] = st
        print(json.dumps(out), flush=True)

    sg, _ = b.esdf()
    judge("scatter insert", sg)
    for s in range(steps):
        new, old = w.next_step()
        for c in range(3):
            b.observe(new, 1)
            if c == 2:
                b.observe(old, 0)
            b.fuse()
        sg, _ = b.esdf()
        judge(f"step {s + 1}", sg)
    gpu.close()
    env.close()


if __name__ == "__main__":
    main()
