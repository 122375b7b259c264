#!/usr/bin/env python3
"""Diagnostic: d^2 mismatch of every engine variant against the golden raycast fixture (partially observed)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_programs import PROGRAMS
from scenarios import GpuAsOracle, compare_gpu_to_golden
gold = np.load(os.path.join(ROOT, "tests", "golden", "raycast_frames.npz"))
for ts in (1, 2, 0, 11, 12):
    out = []
    for cp, m, extra in PROGRAMS["raycast_frames"](lambda o, r, s: GpuAsOracle(o, r, s, tile_shape=ts)):
        rep = compare_gpu_to_golden(m.m, gold, cp)
        out.append((rep["d2_mismatch"], rep["finite"], rep["gpu_finite_cpu_inf"], rep["cpu_finite_gpu_inf"]))
    print("tile_shape", ts, out)
