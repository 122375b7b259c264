"""Randomised pin of the CPU restatement against the verbatim-compiled reference (CPU only, both map flavours).

Same generator idea as tests/test_gpu_fuzz.py, but here BOTH sides follow the reference's FIFO order, so everything must
be identical including closest-obstacle ids, the expansion counters the reference prints and (hash flavour) the internal
slot order.  Skipped where oracle/_ref is not built; tests/golden/ carries the pin there.
"""
import numpy as np
import pytest

from scenarios import P_DEFAULT, all_voxels
from test_oracle_port_vs_ref import both, esdf_same, pair_factory, same_dense  # noqa: F401  (fixture + helpers)


@pytest.mark.parametrize("seed", list(range(100, 112)))
def test_dense_random_sequences(pair_factory, seed):
    rng = np.random.RandomState(seed)
    dims = np.array([int(v) for v in rng.randint(8, 30, 3)])
    res = float(rng.choice([0.05, 0.1, 0.2]))
    origin = tuple(float(v) for v in rng.uniform(-2, 2, 3))
    ms = pair_factory(origin, res, tuple((dims - 0.5) * res))
    assert ms[0].grid_size == tuple(dims) == ms[1].grid_size
    if rng.rand() < 0.7:                                   # fully observed, else only what the batches touch
        for m in ms:
            m.SetOccupancyVox(all_voxels(tuple(dims)), 0)
        both(ms, lambda m: m.UpdateOccupancy(True))
        esdf_same(ms)
    live = np.zeros((0, 3), np.int32)
    for step in range(int(rng.randint(4, 8))):
        n_new = int(rng.randint(1, 40))
        new = np.stack([rng.randint(-2, dims[k] + 2, n_new) for k in range(3)], -1).astype(np.int32)
        gone = live[rng.rand(len(live)) < 0.35]
        if rng.rand() < 0.3:                               # local-window update (SetUpdateRange + global_map=False)
            c = (rng.rand(3) * dims * res) + np.array(origin)
            wlo, whi = c - rng.uniform(0.3, 1.5, 3), c + rng.uniform(0.3, 1.5, 3)
            for m in ms:
                m.SetUpdateRange(wlo, whi)
            glob = False
        else:
            for m in ms:
                m.SetOriginalRange()
            glob = True
        for _ in range(int(rng.choice([1, 3, 6]))):
            by_pos = rng.rand() < 0.5                      # (one draw for both sides)
            for m in ms:
                if by_pos:
                    m.SetOccupancyPos((new + 0.5) * res + np.array(origin), 1)
                else:
                    m.SetOccupancyVox(new, 1)
                if len(gone):
                    m.SetOccupancyVox(gone, 0)
            r = both(ms, lambda m: (m.UpdateOccupancy(glob), m.last_insert, m.last_delete))
            assert r[0] == r[1]
        esdf_same(ms)
        same_dense(*ms)
        inside = np.all((new >= 0) & (new < dims), axis=1)
        keep = set(map(tuple, live.tolist())) - set(map(tuple, gone.tolist())) | set(map(tuple, new[inside].tolist()))
        live = np.array(sorted(keep), np.int32).reshape(-1, 3)
        q = rng.uniform(0.15, 0.85, (100, 3)) * dims * res + np.array(origin)
        for fn in ("GetDistancePos", "GetOccupancyPos"):
            a, b = both(ms, lambda m: getattr(m, fn)(q))
            assert np.array_equal(a, b), fn
        (da, ga), (db, gb) = both(ms, lambda m: m.GetDistWithGradTrilinear(q))
        assert np.array_equal(da, db) and np.array_equal(ga, gb)


@pytest.mark.parametrize("seed", [200, 201, 202, 203])
def test_hash_random_sequences(pair_factory, seed):
    rng = np.random.RandomState(seed)
    res = float(rng.choice([0.05, 0.1]))
    ms = pair_factory(tuple(float(v) for v in rng.uniform(-1, 1, 3)), res, mode="hash", reserve=int(rng.choice([0, 500, 20000])))
    centre = rng.randint(-20, 20, 3)
    live = np.zeros((0, 3), np.int32)
    for step in range(5):
        centre = centre + rng.randint(-5, 6, 3)
        ext = rng.randint(6, 16, 3)
        box = (all_voxels(tuple(int(v) for v in ext)) + (centre - ext // 2)).astype(np.int32)
        new = box[rng.rand(len(box)) < 0.015]
        gone = live[rng.rand(len(live)) < 0.4]
        for k in range(3):
            for m in ms:
                if k == 0:
                    m.SetOccupancyVox(box, 0)
                if len(new):
                    m.SetOccupancyVox(new, 1)
                if len(gone):
                    m.SetOccupancyVox(gone, 0)
            r = both(ms, lambda m: (m.UpdateOccupancy(True), m.last_insert, m.last_delete))
            assert r[0] == r[1]
        esdf_same(ms)
        a, b = both(ms, lambda m: m.dump_hash())
        for kk in ("vox", "dist", "coc", "occ"):
            assert np.array_equal(a[kk], b[kk]), kk
        ha, hb = both(ms, lambda m: m.dump_counts())
        assert np.array_equal(ha[0], hb[0]) and np.array_equal(ha[1], hb[1])
        live = np.concatenate([live, new])
