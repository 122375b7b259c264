#!/usr/bin/env python3
"""Generates tests/golden/raycast_frames_envelope.npz: the ORDER ENVELOPE of the reference on the `raycast_frames` golden
program (a partially observed map, where the reference's distances depend on its queue order).

The verbatim-compiled reference (oracle/_ref) runs the program K + 1 times: once as recorded in raycast_frames.npz and K
times with every frame's observations replayed in a shuffled first-touch order (tests/scenarios.py: EnvelopeOracle --
same counters, same occupancy, only the queue order differs).  Stored per checkpoint, sparsely: the voxels on which the
runs disagree and the smallest / largest squared distance any run holds there.  Everywhere else all runs equal the
fixture.  Needs /root/reference (build container only); the fixture travels.

    python tests/golden/make_golden_envelope.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from golden_programs import PROGRAMS  # noqa: E402
from oracle import pyoracle  # noqa: E402
from scenarios import EnvelopeOracle, d2_from_dist  # noqa: E402

K = 9


def main():
    pyoracle.build("ref")
    assert pyoracle.available("ref", "array"), "the verbatim reference build is required"
    gold = np.load(os.path.join(HERE, "raycast_frames.npz"))

    def make(origin, res, size):
        return EnvelopeOracle(lambda: pyoracle.OracleMap(origin, res, size, kind="ref"), k=K)

    out = {"runs": np.array(K + 1)}
    for cp, m, _ in PROGRAMS["raycast_frames"](make):
        assert np.array_equal(m.primary.dump_dense(("dist",))["dist"], gold[f"{cp}/dist"]), "the primary run is the fixture"
        D = m._fields()
        lo, hi = D.min(0), D.max(0)
        idx = np.flatnonzero(lo != hi)
        out[f"{cp}/idx"], out[f"{cp}/lo"], out[f"{cp}/hi"] = idx.astype(np.int32), lo[idx], hi[idx]
        rep = m.judge(d2_from_dist(gold[f"{cp}/dist"], m.resolution))
        out[f"{cp}/leave_one_out"] = np.array(rep["leave_one_out"])
        print(cp, "finite", rep["finite"], "disagree", len(idx), "leave-one-out", rep["leave_one_out"])
    path = os.path.join(HERE, "raycast_frames_envelope.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
