"""The synthetic inputs of bench.py are part of the measurement contract (fixed seeds): pin them."""
import importlib.util
import os

import numpy as np

from scenarios import c4_frame

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)       # (imports nothing heavy at module level: torch / the HIP library load in main())
    return m


def test_c2_scatter_workload_is_deterministic_and_unique():
    b = _bench()
    w1, w2 = b.Workload(512, 50000), b.Workload(512, 50000)
    a = w1.initial()
    assert np.array_equal(a, w2.initial())
    assert a.shape == (50000, 3) and a.min() >= 0 and a.max() < 512
    assert len(np.unique(a, axis=0)) == 50000
    assert a[:2].tolist() == [[482, 485, 285], [129, 420, 425]]        # seed 12345 (the first draws of the stream)
    new, old = w1.next_step()
    assert new.shape == old.shape == (25000, 3)
    assert np.array_equal(old, a[:25000])                               # the oldest half leaves
    live = set(map(tuple, a.tolist()))
    assert not (set(map(tuple, new.tolist())) & live)                   # fresh voxels only
    assert np.array_equal(w2.next_step()[0], new)


def test_c2_surfaces_scene_lies_on_its_planes_and_spheres():
    b = _bench()
    w = b.Workload(256, 4000, scene="surfaces")
    a = w.initial().astype(np.int64)
    assert a.min() >= 0 and a.max() < 256 and len(np.unique(a, axis=0)) == 4000
    on_plane = np.zeros(len(a), bool)
    for ax, pos in w.planes:
        on_plane |= a[:, ax] == pos
    near_sphere = np.zeros(len(a), bool)
    for c, r in w.spheres:
        near_sphere |= np.abs(np.sqrt(((a - c) ** 2).sum(-1)) - r) < 1.0
    clipped = np.any((a == 0) | (a == 255), axis=1)                     # samples clipped to the grid faces
    assert np.all(on_plane | near_sphere | clipped)
    assert on_plane.sum() > 100 and near_sphere.sum() > 1000


def test_c4_stream_frames_are_deterministic_and_inside_their_window():
    for k in (0, 7, 50, 199):
        lo, hi, occ = c4_frame(k)
        lo2, hi2, occ2 = c4_frame(k)
        assert np.array_equal(lo, lo2) and np.array_equal(hi, hi2) and np.array_equal(occ, occ2)
        assert tuple(hi - lo + 1) == (120, 120, 60)
        assert np.all((occ >= lo) & (occ <= hi)) and len(np.unique(occ, axis=0)) == len(occ)
        assert np.all(np.abs(np.concatenate([lo, hi])) < 512)           # inside the paged map's virtual window
        assert 10000 < len(occ) < 25000
    a, b = c4_frame(10)[2], c4_frame(11)[2]
    sa, sb = set(map(tuple, a.tolist())), set(map(tuple, b.tolist()))
    assert 0.7 < len(sa & sb) / len(sa) < 1.0                           # overlapping windows, moving content
