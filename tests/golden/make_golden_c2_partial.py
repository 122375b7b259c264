#!/usr/bin/env python3
"""Pins "C2-partial" -- BASELINE config 2's workload on a PARTIALLY OBSERVED map (bench.py --unobserved 0.27: 27 % of the
32^3-voxel blocks never observed) -- against the REFERENCE ITSELF, at the size the benchmark runs it.

On such a map the reference's distances depend on the order of its queues (tests/test_oracle_order_sensitivity.py), so the pin is
not one field but the ENVELOPE of K + 1 runs of the verbatim-compiled reference (oracle/_ref, /root/reference/src/ESDFMap.cpp
unmodified) on exactly bench.py's inputs, each run receiving every SetOccupancy batch in its own shuffled order (run 0: bench.py's
order) -- what tests/scenarios.py: EnvelopeOracle does at test sizes.  Written to tests/golden/c2_partial_<grid>_envelope.npz,
for the two checkpoints "scatter" (all obstacles inserted into the empty map) and "step" (one steady-state step: half of the
obstacles replaced in ONE UpdateESDF):

    <cp>/exc_idx, exc_lo, exc_hi   the voxels where some run differs from T, the exact transform of the EFFECTIVE sites masked to
                  the observed voxels (tests/masked_model.py: effective_sites) -- everywhere else every run equals T, which a test
                  recomputes with scipy --, with the smallest and largest squared distance over the runs (0x7FFFFFFF = no obstacle)
    <cp>/disagree, <cp>/leave_one_out   voxels on which the runs disagree; per run, voxels where it leaves the others' envelope
    <cp>/occ_crc, obs_crc, n_occ, n_obs, finite    CRC32 of the packed occupancy / observed bitmaps (identical in every run), counts
    <cp>/model_idx, model_d2      (with --model) the voxels where the numpy model of the masked transform differs from T, and its value

    python tests/golden/make_golden_c2_partial.py --grid 256 [--runs 4] [--model] [--pattern sensor]

--pattern sensor: the same workload on a map observed through a handful of view cones (c2_sensor_<grid>_envelope.npz): the boundary
of the observed space cuts cells and bitmap words, which the block pattern never does.

Needs /root/reference (build container only); at 512^3 7.4 GB and ~3 min per run (the runs go side by side: --jobs).
"""
import argparse
import multiprocessing as mp
import os
import sys
import time
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, ROOT)

P_DEFAULT = (0.70, 0.35, 0.12, 0.97, 0.80)
D2_INF = 0x7FFFFFFF
UNOBSERVED = 0.27


def d2_of(dist, res):
    d2 = np.rint((dist / res) ** 2)
    d2[dist >= 10000] = D2_INF
    d2[dist < 0] = -1
    return d2.astype(np.int32)


def one_run(args):
    """One run of the verbatim reference; run 0 in bench.py's order, run r > 0 with every batch shuffled (seed r)."""
    G, r, tmp, pattern = args
    from oracle import pyoracle
    import bench
    res = 0.1
    rng = np.random.RandomState(90210 + r)
    shuf = (lambda v: v) if r == 0 else (lambda v: v[rng.permutation(len(v))])
    m = pyoracle.OracleMap((0, 0, 0), res, (G * res,) * 3, kind="ref")
    assert m.grid_total_size == G ** 3
    m.SetParameters(*P_DEFAULT)
    m.SetOriginalRange()
    if pattern == "sensor":   # a sensor-shaped observed set (tests/masked_model.py: sensor_mask), voxel by voxel
        import masked_model
        v = np.argwhere(masked_model.sensor_mask(G)).astype(np.int32)
        if r:
            v = v[rng.permutation(len(v))]
        for s in range(0, len(v), 1 << 21):
            m.SetOccupancyVox(v[s:s + (1 << 21)], 0)
    else:
        keep = np.random.RandomState(2718).rand(G // 32, G // 32, G // 32) >= UNOBSERVED
        blocks = np.argwhere(keep)
        if r:
            blocks = blocks[rng.permutation(len(blocks))]
        cube = np.stack(np.meshgrid(np.arange(32), np.arange(32), np.arange(32), indexing="ij"), -1).reshape(-1, 3)
        for s in range(0, len(blocks), 64):   # the prologue, block by block as bench.py observes it (SetOccupancyBox per block)
            v = (blocks[s:s + 64, None, :] * 32 + cube[None]).reshape(-1, 3).astype(np.int32)
            m.SetOccupancyVox(shuf(v), 0)
    m.UpdateOccupancy(True)
    m.UpdateESDF()
    n_obs = max(2, int(round(50000 * (G / 512.0) ** 3)))
    w = bench.Workload(G, n_obs, seed=12345)
    for _ in range(3):
        m.SetOccupancyVox(shuf(w.initial()), 1)
        m.UpdateOccupancy(True)
    t0 = time.time()
    st = m.UpdateESDF()
    out = {}
    d = m.dump_dense(("dist", "occ"))
    np.save(os.path.join(tmp, f"r{r}_scatter_d2.npy"), d2_of(d["dist"], res))
    if r == 0:
        np.save(os.path.join(tmp, "scatter_occ.npy"), d["occ"].astype(np.uint8))
    out["scatter_s"] = st["seconds"]
    new, old = w.next_step()
    both = np.concatenate([new, old])
    occ = np.concatenate([np.ones(len(new), np.int32), np.zeros(len(old), np.int32)])
    for c in range(3):
        if c < 2:
            m.SetOccupancyVox(shuf(new), 1)
        else:
            p = np.arange(len(both)) if r == 0 else rng.permutation(len(both))
            m.SetOccupancyVox(both[p], occ[p])
        m.UpdateOccupancy(True)
    st = m.UpdateESDF()
    d = m.dump_dense(("dist", "occ"))
    np.save(os.path.join(tmp, f"r{r}_step_d2.npy"), d2_of(d["dist"], res))
    if r == 0:
        np.save(os.path.join(tmp, "step_occ.npy"), d["occ"].astype(np.uint8))
    out["step_s"] = st["seconds"]
    out["wall_s"] = time.time() - t0
    m.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--runs", type=int, default=4, help="runs of the reference (K + 1)")
    ap.add_argument("--jobs", type=int, default=4)
    ap.add_argument("--tmp", default="/tmp/c2_partial")
    ap.add_argument("--model", action="store_true", help="also store the numpy model's field (tests/masked_model.py)")
    ap.add_argument("--reuse", action="store_true", help="use the runs' dumps already in --tmp")
    ap.add_argument("--pattern", default="blocks", choices=["blocks", "sensor"],
                    help="the never-observed space: bench.py's 32^3 blocks, or outside a union of view cones (masked_model.sensor_mask)")
    a = ap.parse_args()
    G = a.grid
    os.makedirs(a.tmp, exist_ok=True)
    if not a.reuse:
        with mp.get_context("spawn").Pool(a.jobs) as pool:
            info = pool.map(one_run, [(G, r, a.tmp, a.pattern) for r in range(a.runs)])
        print(info, flush=True)
    import masked_model
    from scipy import ndimage
    out = {"grid": np.array([G, G, G]), "runs": np.array(a.runs), "unobserved": np.array(UNOBSERVED)}
    W = None
    for cp in ("scatter", "step"):
        D = np.stack([np.load(os.path.join(a.tmp, f"r{r}_{cp}_d2.npy")) for r in range(a.runs)])
        occ = np.load(os.path.join(a.tmp, f"{cp}_occ.npy")).reshape(G, G, G) != 0
        obs = (D[0] >= 0).reshape(G, G, G)
        for r in range(1, a.runs):
            assert np.array_equal(D[r] >= 0, D[0] >= 0), "observed sets differ between the runs"
        eff = masked_model.effective_sites(occ, obs)
        idx = ndimage.distance_transform_edt(~eff, return_distances=False, return_indices=True)
        g = np.meshgrid(*[np.arange(G, dtype=np.int32)] * 3, indexing="ij")
        T = sum((idx[k].astype(np.int64) - g[k]) ** 2 for k in range(3))
        T = np.where(obs, T, -1).astype(np.int32).reshape(-1)
        del idx, g
        lo, hi = D.min(0), D.max(0)
        exc = np.flatnonzero((lo != T) | (hi != T))
        loo = []
        for k in range(a.runs):
            rest = np.delete(D, k, 0)
            loo.append(int(((D[k] < rest.min(0)) | (D[k] > rest.max(0))).sum()))
        fin = (D[0] >= 0) & (D[0] != D2_INF)
        out[f"{cp}/exc_idx"] = exc.astype(np.uint32)
        out[f"{cp}/exc_lo"] = lo[exc]
        out[f"{cp}/exc_hi"] = hi[exc]
        out[f"{cp}/disagree"] = np.array(int((lo != hi).sum()))
        out[f"{cp}/leave_one_out"] = np.array(loo)
        out[f"{cp}/occ_crc"] = np.array(zlib.crc32(np.packbits(occ.reshape(-1)).tobytes()))
        out[f"{cp}/obs_crc"] = np.array(zlib.crc32(np.packbits(obs.reshape(-1)).tobytes()))
        out[f"{cp}/n_occ"] = np.array(int(occ.sum()))
        out[f"{cp}/n_obs"] = np.array(int(obs.sum()))
        out[f"{cp}/finite"] = np.array(int(fin.sum()))
        print(cp, {k.split("/")[1]: (v.tolist() if v.size < 8 else v.shape) for k, v in out.items() if k.startswith(cp)}, flush=True)
        if a.model:
            t0 = time.time()
            d2m, W, st = masked_model.masked_engine(occ, obs, W)
            d2m = d2m.reshape(-1).astype(np.int32)
            mi = np.flatnonzero(d2m != T)
            out[f"{cp}/model_idx"] = mi.astype(np.uint32)
            out[f"{cp}/model_d2"] = d2m[mi]
            print(cp, "model", st, "closer", int((d2m < lo).sum()), "farther", int((d2m > hi).sum()), f"{time.time() - t0:.0f} s", flush=True)
    out["pattern"] = np.array(a.pattern)
    path = os.path.join(HERE, f"c2_{'partial' if a.pattern == 'blocks' else a.pattern}_{G}_envelope.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
