#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REFERENCE ITSELF: the reference's src/ESDFMap.cpp + src/raycast.cpp
compiled verbatim (oracle/Makefile target `ref` -> oracle/_ref/libfiesta_ref_array.so).  Needs /root/reference,
i.e. it only runs in the build container; the fixtures it writes travel to machines that have no reference.

    python tests/golden/make_golden.py

Stored per checkpoint: distance_buffer_ (f64), closest_obstacle_ (int16), Exist() (uint8), occupancy_buffer_
(f64) exactly as the reference held them, the queue sizes UpdateOccupancy produced, the counters UpdateESDF
prints (src/ESDFMap.cpp:277,394) and query results.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from golden_programs import PROGRAMS, golden_rays  # noqa: E402
from oracle import pyoracle  # noqa: E402


def main():
    pyoracle.build("ref")
    assert pyoracle.available("ref", "array"), "the verbatim reference build is required"

    def make(origin, res, size):
        return pyoracle.OracleMap(origin, res, size, kind="ref")

    for name, prog in PROGRAMS.items():
        out = {}
        for cp, m, extra in prog(make):
            assert m.describe.startswith("reference"), m.describe
            d = m.dump_dense()
            out[f"{cp}/dist"] = d["dist"]
            out[f"{cp}/coc"] = d["coc"].astype(np.int16)
            out[f"{cp}/occ"] = d["occ"]
            out[f"{cp}/logodds"] = d["logodds"]
            out[f"{cp}/grid_size"] = np.array(m.grid_size)
            for k, v in extra.items():
                if k == "stats":
                    out[f"{cp}/stats"] = np.array([v["inserted"], v["deleted"], v["expanded"], v["change_num"]])
                elif k == "pos":
                    out[f"{cp}/pos"] = v
                    out[f"{cp}/GetDistance"] = m.GetDistancePos(v)
                    dist, grad = m.GetDistWithGradTrilinear(v)
                    out[f"{cp}/TrilinearDist"], out[f"{cp}/TrilinearGrad"] = dist, grad
                    out[f"{cp}/GetOccupancy"] = m.GetOccupancyPos(v)
                else:
                    out[f"{cp}/{k}"] = v
        path = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(path, **out)
        print(f"{path}: {os.path.getsize(path) / 1024:.1f} KiB, {len(out)} arrays")
    rays, lo, hi = golden_rays()
    out = {"lo": lo, "hi": hi}
    for i, (a, b) in enumerate(rays):
        out[f"a{i}"], out[f"b{i}"] = a, b
        out[f"v{i}"] = pyoracle.raycast(a, b, lo, hi, kind="ref").astype(np.int16)
    path = os.path.join(HERE, "raycast_kat.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
