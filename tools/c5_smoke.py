#!/usr/bin/env python3
"""Config-5-sized shards on ONE GPU: a 2048 x 1024 x 1024 grid (2.1 G voxels) as 2 shards of 1024^3 multiplexed on the one
visible MI355X through the native shard group (local transport) -- the shard size, id encoding (coordinates modulo 1024),
wide site packing and margins of BASELINE config 5, at half its shard count because eight such shards (8 x ~45 GB) do not
fit one GPU.  Scatter scene at C2's density; checks a random sample of voxels against a k-d tree over the obstacles.

    python tools/c5_smoke.py [--engine auto|rounds]  ->  one JSON line (also: profiles/r02b_c5_two_shards.json)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--engine", default="auto")
    ap.add_argument("--shard", type=int, default=1024)
    a = ap.parse_args()
    from scipy.spatial import cKDTree
    from fiesta_amd.sharded import ShardedESDFMap
    S = a.shard
    gs = (2 * S, S, S)
    n_obs = int(round(50000 * (gs[0] * gs[1] * gs[2]) / 512 ** 3))
    t0 = time.time()
    sm = ShardedESDFMap((0, 0, 0), 0.1, gs, 2, update_engine=a.engine)
    sm.SetParameters(0.70, 0.35, 0.12, 0.97, 0.80)
    sm.SetOriginalRange()
    sm.SetOccupancyBox((0, 0, 0), tuple(np.array(gs) - 1), 0)
    sm.UpdateOccupancy(True)
    sm.UpdateESDF()
    t_prologue = time.time() - t0
    rng = np.random.RandomState(12345)
    A = np.unique((rng.rand(n_obs, 3) * gs).astype(np.int32), axis=0)
    for _ in range(3):
        sm.SetOccupancy(A, 1)
        sm.UpdateOccupancy(True)
    t1 = time.time()
    st1 = sm.UpdateESDF()
    t_insert = time.time() - t1
    B = (rng.rand(n_obs // 2, 3) * gs).astype(np.int32)
    old = A[: len(A) // 2]
    for c in range(3):
        sm.SetOccupancy(B, 1)
        if c == 2:
            sm.SetOccupancy(old, 0)
        sm.UpdateOccupancy(True)
    ni, nd = sm.last_insert, sm.last_delete
    t2 = time.time()
    st2 = sm.UpdateESDF()
    t_step = time.time() - t2
    live = np.unique(np.concatenate([A[len(A) // 2:], B]), axis=0)
    # a sample of voxels, incl. a band around the cut x = S and the id wrap at multiples of 1024
    q = (rng.rand(40000, 3) * gs).astype(np.int64)
    q[:8000, 0] = rng.randint(S - 40, S + 40, 8000)
    tree = cKDTree(live.astype(np.float64))
    dist, _ = tree.query(q.astype(np.float64))
    want = np.rint(dist ** 2).astype(np.int64)
    got = (np.asarray(sm.GetDistance(q.astype(np.int32))) / 0.1) ** 2
    wrong = np.flatnonzero(np.abs(got - want) > 1e-6 * np.maximum(want, 1))
    bad = int(len(wrong))
    for i in wrong[:12]:
        print("MISMATCH voxel", q[i].tolist(), "got d2", float(got[i]), "want", int(want[i]), file=sys.stderr)
    out = {"grid": list(gs), "shards": 2, "engine": a.engine, "obstacles": int(len(A)), "prologue_s": t_prologue,
           "insert": {"bulk": st1["bulk"], "rounds": st1["rounds"], "sweeps": st1.get("sweeps"), "wall_ms": t_insert * 1e3,
                      "ft_max_d2": st1.get("ft_max_d2")},
           "step": {"inserted": ni, "deleted": nd, "bulk": st2["bulk"], "rounds": st2["rounds"], "sweeps": st2.get("sweeps"),
                    "wall_ms": t_step * 1e3, "halo_entries_sent": st2.get("halo_entries_sent"),
                    "cells": st2.get("cells"),   # 1: every shard ran the cell transform (nn_kernels.hpp), else the envelope passes
                    "nn_ms_max_over_shards": [st2.get("nn_cells_ms"), st2.get("nn_lists_ms"), st2.get("nn_fill_ms")],
                    "ft_ms_max_over_shards": [st2.get("ft_rows_ms"), st2.get("ft_plane_ms"), st2.get("ft_x_ms")]},
           "sample": {"voxels": len(q), "mismatch_vs_kdtree": bad}}
    print(json.dumps(out))
    sm.close()
    assert bad == 0


if __name__ == "__main__":
    main()
