#!/usr/bin/env python3
"""Pins BASELINE config 2 at FULL size (512^3) against the REFERENCE ITSELF (parity gap 1).

Runs the verbatim-compiled reference (oracle/_ref/libfiesta_ref_array.so, i.e. /root/reference/src/ESDFMap.cpp
unmodified) once on exactly the inputs bench.py generates for C2 -- observe-all prologue, scatter insert of the 50 000
obstacles (scene A), then one steady-state step of 25 000 inserts + 25 000 deletes landing in one UpdateESDF (scene B) --
and writes a compact digest to tests/golden/c2_512_<scene>_digest.npz:

    <cp>/crc      per-x-slab CRC32 of the slab's squared voxel distances (int32 LE; 0x7FFFFFFF = observed/no obstacle,
                  -1 = never observed), 512 values
    <cp>/sum_d2   sum of all finite d^2 (int64)
    <cp>/n_occ    Exist() count
    <cp>/stats    inserted, deleted, expanded, change_num exactly as UpdateESDF printed them (src/ESDFMap.cpp:277,394)
    <cp>/crc_exact   the same CRCs for the EXACT Euclidean feature transform of the occupied set (scipy)
    <cp>/exc_idx, <cp>/exc_ref_d2   the voxels where the reference is NOT the exact transform (linear index, the
                  reference's d^2 there).  The reference's 24-neighbour vector propagation is exact almost everywhere
                  on a fully observed map, but not everywhere: a few voxels per 10^8 on the scatter scenes, ~3 per 10^5
                  on the surface scene keep a slightly larger distance -- and WHICH voxels depends on the order of the
                  inserts inside one batch (tests/test_oracle_order_sensitivity.py), so no parallel engine can
                  reproduce them.  The GPU contract: equal to the reference everywhere else, exact on these.
    step/updated  the benchmark's unit of work between the two checkpoints

    python tests/golden/make_golden_c2.py [--grid 512] [--scene scatter|surfaces]

Needs /root/reference (build container only), ~8 GB of RAM and ~3-4 minutes per scene on one core.
"""
import argparse
import os
import sys
import time
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import pyoracle  # noqa: E402

P_DEFAULT = (0.70, 0.35, 0.12, 0.97, 0.80)
D2_INF = 0x7FFFFFFF


def field_d2(dump, g):
    """int32 d^2 per voxel from the reference's closest_obstacle_ (the distance is sqrt(d2)*res exactly)."""
    coc = dump["coc"]
    d2 = np.empty(g ** 3, np.int32)
    step = g * g * 16
    for s in range(0, g ** 3, step):
        idx = np.arange(s, min(s + step, g ** 3), dtype=np.int64)
        c = coc[s:s + len(idx)].astype(np.int64)
        d = (idx // (g * g) - c[:, 0]) ** 2 + ((idx // g) % g - c[:, 1]) ** 2 + (idx % g - c[:, 2]) ** 2
        d = np.where(c[:, 0] == -10000, D2_INF, d)
        d = np.where(dump["dist"][s:s + len(idx)] < 0, -1, d)
        d2[s:s + len(idx)] = d.astype(np.int32)
    return d2


def digest(d2, g):
    slab = g * g
    crc = np.array([zlib.crc32(d2[x * slab:(x + 1) * slab].tobytes()) for x in range(g)], np.uint32)
    fin = (d2 >= 0) & (d2 != D2_INF)
    return crc, int(d2[fin].astype(np.int64).sum())


def exact_edt_d2(occ, g):
    from scipy import ndimage
    idx = ndimage.distance_transform_edt(occ.reshape(g, g, g) == 0, return_distances=False, return_indices=True)
    ax = np.arange(g, dtype=np.int32)
    d2 = (idx[0] - ax[:, None, None]).astype(np.int64) ** 2
    d2 += (idx[1] - ax[None, :, None]).astype(np.int64) ** 2
    d2 += (idx[2] - ax[None, None, :]).astype(np.int64) ** 2
    return d2.reshape(-1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--obstacles", type=int, default=50000)
    ap.add_argument("--scene", default="scatter")
    ap.add_argument("--no-edt", action="store_true")
    a = ap.parse_args()
    sys.path.insert(0, ROOT)
    from bench import Workload  # the very generator the benchmark uses (pinned by tests/test_bench_inputs.py)

    pyoracle.build("ref")
    assert pyoracle.available("ref", "array"), "the verbatim reference build is required"
    g, res = a.grid, 0.1
    m = pyoracle.OracleMap((0, 0, 0), res, ((g - 0.5) * res,) * 3, kind="ref")
    assert m.describe.startswith("reference") and m.grid_total_size == g ** 3
    m.SetParameters(*P_DEFAULT)
    m.SetOriginalRange()
    t0 = time.time()
    idx = np.arange(g ** 3, dtype=np.int64)
    allv = np.stack([idx // (g * g), (idx // g) % g, idx % g], -1).astype(np.int32)
    del idx
    m.SetOccupancyVox(allv, 0)
    del allv
    m.UpdateOccupancy(True)
    m.UpdateESDF()
    print(f"prologue {time.time() - t0:.1f} s", flush=True)
    w = Workload(g, a.obstacles, seed=12345, scene=a.scene)
    out = {"grid": np.array([g, g, g]), "obstacles": np.array(a.obstacles)}

    def checkpoint(cp, st):
        d = m.dump_dense(("dist", "coc", "occ"))
        d2 = field_d2(d, g)
        crc, s = digest(d2, g)
        out[f"{cp}/crc"], out[f"{cp}/sum_d2"], out[f"{cp}/n_occ"] = crc, np.array(s), np.array(int(d["occ"].sum()))
        out[f"{cp}/stats"] = np.array([st["inserted"], st["deleted"], st["expanded"], st["change_num"]])
        out[f"{cp}/seconds"] = np.array(st["seconds"])
        if not a.no_edt:
            e = exact_edt_d2(d["occ"], g).astype(np.int32)
            exc = np.flatnonzero(e != d2)
            assert np.all(d2[exc] > e[exc]), "the reference can only be farther than the exact transform"
            out[f"{cp}/crc_exact"] = digest(e, g)[0]
            out[f"{cp}/exc_idx"], out[f"{cp}/exc_ref_d2"] = exc.astype(np.int64), d2[exc]
            print(f"{cp}: reference vs exact EDT: {len(exc)} voxels differ", flush=True)
        print(f"{cp}: {st}", flush=True)
        return d, d2

    for _ in range(3):
        m.SetOccupancyVox(w.initial(), 1)
        m.UpdateOccupancy(True)
    st = m.UpdateESDF()
    before, bd2 = checkpoint("scatter", st)
    new, old = w.next_step()
    for c in range(3):
        m.SetOccupancyVox(new, 1)
        if c == 2:
            m.SetOccupancyVox(old, 0)
        m.UpdateOccupancy(True)
    assert (m.last_insert, m.last_delete) == (len(new), len(old))
    st = m.UpdateESDF()
    after, ad2 = checkpoint("step", st)
    changed = bd2 != ad2
    bc = before["coc"].astype(np.int64)
    has = bc[:, 0] >= 0
    lin = (bc[:, 0] * g + bc[:, 1]) * g + bc[:, 2]
    gone = np.zeros(len(lin), bool)
    gone[has] = after["occ"][lin[has]] == 0
    out["step/updated"] = np.array(int((changed | gone).sum()))
    print("updated voxels of the step:", int(out["step/updated"]), flush=True)
    name = f"c2_{g}_{a.scene}_digest.npz"
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name)


if __name__ == "__main__":
    main()
