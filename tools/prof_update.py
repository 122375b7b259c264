#!/usr/bin/env python3
"""Engine diagnostics: runs the C2 scatter insert and one steady-state step on a G^3 grid and prints the
UpdateESDF counters (rounds, tile visits, levels, writes, per-phase cycles with FIESTA_HIP_PROF=1).
With --compare E the same workload runs on a second engine (update_engine: 0 auto, 1 frontier rounds, 2 bulk transform) and the d^2 fields are compared.
    FIESTA_HIP_PROF=1 python tools/prof_update.py --grid 512 --engine 1 --compare 2
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import fiesta_amd  # noqa: E402
from bench import P_DEFAULT, Workload  # noqa: E402


def run(G, n_obs, ts, hash_mode=False):
    dev = torch.device("cuda", 0)
    res = 0.1
    if hash_mode:  # same workload on the paged ("bricked") map: grid placed at voxels [-G/2, G/2)
        m = fiesta_amd.ESDFMap((0, 0, 0), res, reserve_size=G ** 3, mode="hash")
        m.SetParameters(*P_DEFAULT)
        m.SetOriginalRange()
        off = -G // 2
        for x0 in range(0, G, 32):
            g = np.stack(np.meshgrid(np.arange(x0, x0 + 32), np.arange(G), np.arange(G), indexing="ij"), -1).reshape(-1, 3)
            m.SetOccupancy((g + off).astype(np.int32), 0, want_ret=False)
        m.UpdateOccupancy(True)
        m.UpdateESDF()
        w = Workload(G, n_obs)
        out = {}
        init = w.initial() + off
        for _ in range(3):
            m.SetOccupancy(init, 1, want_ret=False)
            m.UpdateOccupancy(True)
        st = m.UpdateESDF()
        st["updated"] = 0
        out["scatter"] = st
        for k in range(2):
            new, old = w.next_step()
            for c in range(3):
                m.SetOccupancy(new + off, 1, want_ret=False)
                if c == 2:
                    m.SetOccupancy(old + off, 0, want_ret=False)
                m.UpdateOccupancy(True)
            st = m.UpdateESDF()
            st["updated"] = 0
            out[f"steady{k}"] = st
        m.close()
        return out, None, None
    m = fiesta_amd.ESDFMap((0, 0, 0), res, (G * res,) * 3, update_engine=ts)
    m.SetParameters(*P_DEFAULT)
    m.SetOriginalRange()
    for x0 in range(0, G, 64):
        xs = torch.arange(x0, min(G, x0 + 64), device=dev, dtype=torch.int32)
        ys = torch.arange(G, device=dev, dtype=torch.int32)
        v = torch.stack(torch.meshgrid(xs, ys, ys, indexing="ij"), -1).reshape(-1, 3).contiguous()
        o = torch.zeros(v.shape[0], dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        m.SetOccupancyDevice(v.data_ptr(), o.data_ptr(), v.shape[0])
        m.synchronize()
    m.UpdateOccupancy(True)
    m.UpdateESDF()
    w = Workload(G, n_obs)
    init = w.initial()
    out = {}
    for _ in range(3):
        m.SetOccupancy(init, 1, want_ret=False)
        m.UpdateOccupancy(True)
    m.snapshot_save(0)
    st = m.UpdateESDF()
    st["updated"] = m.snapshot_count_updated(0)
    out["scatter"] = st
    for k in range(2):
        new, old = w.next_step()
        for c in range(3):
            m.SetOccupancy(new, 1, want_ret=False)
            if c == 2:
                m.SetOccupancy(old, 0, want_ret=False)
            m.UpdateOccupancy(True)
        m.snapshot_save(0)
        st = m.UpdateESDF()
        st["updated"] = m.snapshot_count_updated(0)
        out[f"steady{k}"] = st
    f = m.download_field(("d2", "occ"))
    m.close()
    return out, f["d2"], f["occ"]


def show(tag, st):
    p = st["prof"]
    tot = max(1, p[0] + p[1] + p[2])
    print(f"{tag:9s} dev {st['device_ms']:8.3f} ms relax {st['relax_ms']:8.3f} ms rounds {st['rounds']:3d} visits {st['tile_visits']:8d} "
          f"levels {st['sweeps']:9d} writes {st['voxel_writes']:10d} updated {st['updated']:10d} inval {st['invalidated']:9d} | "
          f"cycles stage {p[0] / tot:.2f} propagate {p[1] / tot:.2f} writeback {p[2] / tot:.2f} items {p[3]} "
          f"cyc/visit {tot / max(1, st['tile_visits']):.0f} | compact {p[4] / tot:.2f} process {p[5] / tot:.2f} pulls {p[6]} push-ok {p[7]}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--obstacles", type=int, default=None)
    ap.add_argument("--engine", type=int, default=0)
    ap.add_argument("--compare", type=int, default=None)
    ap.add_argument("--hash", action="store_true", help="run the workload on the paged hash-block map")
    a = ap.parse_args()
    n_obs = a.obstacles or int(round(50000 * (a.grid / 512) ** 3))
    o1, d1, occ1 = run(a.grid, n_obs, a.engine, a.hash)
    for k, st in o1.items():
        show(f"ts{a.engine}:{k}", st)
    if a.compare is not None:
        o2, d2, occ2 = run(a.grid, n_obs, a.compare)
        for k, st in o2.items():
            show(f"ts{a.compare}:{k}", st)
        assert np.array_equal(occ1, occ2)
        idx = np.flatnonzero(d1 != d2)
        bad = len(idx)
        print(f"d2 fields differ at {bad} of {d1.size} voxels")
        if bad:  # which one is the exact Euclidean distance transform there?
            G = a.grid
            obs = np.flatnonzero(occ1).astype(np.int64)
            O = np.stack([obs // (G * G), (obs // G) % G, obs % G], -1)
            for i in idx[:20]:
                v = np.array([i // (G * G), (i // G) % G, i % G])
                exact = int(((O - v) ** 2).sum(-1).min())
                print(f"  voxel {tuple(v)}: ts{a.engine} d2={d1[i]} ts{a.compare} d2={d2[i]} exact EDT d2={exact}")
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
