"""dev: where does the local-window scenario leave the reference's envelope?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import pyoracle
import fiesta_amd
from scenarios import *
from test_gpu_dense_parity import make_pair, observe_all
n = 48
b = make_pair(pyoracle, "ref", n, envelope=6)
observe_all(b, n)
rng = np.random.RandomState(21)
S = rng.randint(4, n - 4, (250, 3)).astype(np.int32)
b.make_occupied(S)
b.esdf()
for step in range(4):
    c = np.array([1.2 + 0.5 * step, 2.0, 2.4])
    lo, hi = c - [1.5, 1.5, 1.0], c + [1.5, 1.5, 1.0]
    for m in (b.gpu, b.cpu):
        m.SetUpdateRange(lo, hi)
    new = (c / 0.1 + rng.randint(-12, 12, (60, 3))).astype(np.int32)
    gone = S[rng.choice(len(S), 40, replace=False)]
    for _ in range(6):
        b.observe(new, 1)
        b.observe(gone, 0)
        b.fuse(global_map=False)
    b.esdf()
    f = b.gpu.download_field()
    gd2 = f["d2"].astype(np.int64)
    D = b.cpu._fields()
    lo2, hi2 = D.min(0), D.max(0)
    wlo = np.floor(lo / 0.1).astype(int); whi = np.floor((hi - 0.05) / 0.1).astype(int)
    gs = b.gpu.grid_size; idx = np.arange(gs[0]*gs[1]*gs[2]); V = np.stack([idx // (gs[1] * gs[2]), (idx // gs[2]) % gs[1], idx % gs[2]], -1)
    inwin = np.all((V >= wlo) & (V <= whi), axis=1)
    far = gd2 > hi2; close = gd2 < lo2
    print("step", step, "window", wlo, whi, "farther", far.sum(), "in-window", (far & inwin).sum(), "gpu inf", (far & (gd2 == D2_INF)).sum(),
          "closer", close.sum(), "closer in-window", (close & inwin).sum(), "disagree", (lo2 != hi2).sum(), "disagree in-window", ((lo2 != hi2) & inwin).sum())
    # farther & finite: how much
    ff = far & (gd2 < D2_INF)
    if ff.any():
        print("   finite farther: d2 gpu/hi", list(zip(gd2[ff][:8], hi2[ff][:8])), "in-window", (ff & inwin).sum())
    fi = far & (gd2 == D2_INF)
    if fi.any():
        print("   inf farther: ref hi", hi2[fi][:10], "in-window", (fi & inwin).sum())
