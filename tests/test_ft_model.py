"""The bulk path's integer machinery (fiesta_amd/csrc/ft_core.hpp: streaming lower envelope, nearest set bit of a row)
checked on the CPU: tests/cpp/ft_model.cpp drives the very header the HIP kernels instantiate -- 64-lane waves,
lock-step emission, a deque that outgrows its ring and goes on in the backing store (the wave's spill mode: eviction of the
oldest ring entry, pops and the emission point reaching into the store) -- and the result must be the exact Euclidean feature transform
(squared distances equal to scipy's EDT; every closest site occupied).  No GPU, no oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
from scipy import ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "cpp", "ft_model.cpp")
LIB = os.path.join(HERE, "cpp", "libft_model.so")
CORE = os.path.join(os.path.dirname(HERE), "fiesta_amd", "csrc", "ft_core.hpp")


@pytest.fixture(scope="module")
def model():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(CORE)):
        subprocess.run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", LIB, SRC], check=True)
    lib = C.CDLL(LIB)
    lib.ft_model_run.restype = C.c_int
    lib.ft_model_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ft_model_run_wide.restype = C.c_int
    lib.ft_model_run_wide.argtypes = lib.ft_model_run.argtypes
    return lib


def run(lib, occ, S0):
    nx, ny, nz = occ.shape
    occ = np.ascontiguousarray(occ, dtype=np.uint8)
    out = np.empty(occ.shape, np.uint32)
    stats = np.zeros(3, np.int32)
    rc = lib.ft_model_run(occ.ctypes.data, nx, ny, nz, S0, out.ctypes.data, stats.ctypes.data)
    assert rc == 0, rc
    return out, stats


def check_exact(occ, out):
    nx, ny, nz = occ.shape
    if not occ.any():
        assert np.all(out == 0x80000000)
        return
    idx = ndimage.distance_transform_edt(occ == 0, return_distances=False, return_indices=True)
    gx, gy, gz = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    want = (idx[0] - gx) ** 2 + (idx[1] - gy) ** 2 + (idx[2] - gz) ** 2
    cx, cy, cz = (out >> 20).astype(np.int64), ((out >> 10) & 1023).astype(np.int64), (out & 1023).astype(np.int64)
    assert np.all(out < 0x40000000)
    assert cx.max() < nx and cy.max() < ny and cz.max() < nz
    assert np.all(occ[cx, cy, cz] == 1), "closest site is not occupied"
    got = (cx - gx) ** 2 + (cy - gy) ** 2 + (cz - gz) ** 2
    bad = np.flatnonzero(got != want)
    assert len(bad) == 0, f"{len(bad)} voxels differ from the exact transform, first {np.unravel_index(bad[:3], occ.shape)}"


CASES = [
    ((20, 24, 70), 0.002, 11), ((33, 17, 130), 0.0005, 12), ((40, 40, 64), 0.01, 13), ((7, 50, 65), 0.05, 14),
    ((48, 48, 96), 0.0001, 15), ((16, 16, 200), 0.3, 16), ((5, 5, 5), 0.1, 17), ((64, 3, 129), 0.003, 18),
]


@pytest.mark.parametrize("shape,density,seed", CASES)
@pytest.mark.parametrize("S0", [64, 4])
def test_random_scatter_is_exact(model, shape, density, seed, S0):
    rng = np.random.RandomState(seed)
    occ = (rng.rand(*shape) < density).astype(np.uint8)
    if not occ.any():
        occ[tuple(rng.randint(0, s) for s in shape)] = 1
    out, stats = run(model, occ, S0)
    check_exact(occ, out)
    if S0 == 4 and density <= 0.01 and min(shape) > 8:
        assert stats[1] > 0 and stats[2] > 0, "a 4-entry ring must spill somewhere on sparse fields (spill mode was not exercised)"


def test_no_site_and_single_site(model):
    occ = np.zeros((9, 10, 70), np.uint8)
    out, _ = run(model, occ, 64)
    check_exact(occ, out)
    occ[8, 0, 69] = 1
    out, stats = run(model, occ, 64)
    check_exact(occ, out)


def test_walls_and_shells_need_deep_rings(model):
    """A wall parallel to a column makes every position a different winner: the deque is as deep as twice the
    distance; rings of 32, 16, 8 and 4 entries spill more and more of it into the backing store."""
    occ = np.zeros((70, 40, 66), np.uint8)
    occ[:, 2, :] = 1          # wall y = 2
    occ[35, 30:40, 10:60] = 1  # plate
    g = np.stack(np.meshgrid(np.arange(70), np.arange(40), np.arange(66), indexing="ij"), -1)
    r = np.sqrt(((g - np.array([30, 20, 30])) ** 2).sum(-1))
    occ[np.abs(r - 14) < 0.6] = 1
    out, stats = run(model, occ, 32)
    check_exact(occ, out)
    seen = []
    for S0 in (16, 8, 4):
        out2, stats2 = run(model, occ, S0)
        check_exact(occ, out2)
        assert stats2[1] > 0 and stats2[2] > 0
        seen.append(int(stats2[2]))
    assert seen[0] < seen[1] < seen[2], seen      # a smaller ring evicts more


def test_far_field_of_a_wall(model):
    occ = np.zeros((150, 12, 64), np.uint8)
    occ[:, 0, :] = 1  # wall; the far rows are 11 voxels away, columns along x see one new winner per position
    occ[:, 11, 5] = 0
    out, stats = run(model, occ, 8)
    check_exact(occ, out)
    assert stats[0] >= 8 and stats[1] > 0


def test_column_as_long_as_the_deepest_ring_allows(model):
    """Every site of a 1024-long column equally good and 1023 voxels away: nothing is final before the last site has
    arrived, so the deque holds one entry per position -- all but the ring's S - 1 newest in the backing store, which
    has one slot per counter value (a column's positions + 2)."""
    occ = np.zeros((1024, 1024, 1), np.uint8)
    occ[:, 0, 0] = 1
    out, stats = run(model, occ, 64)
    y = np.arange(1024)
    got_x, got_y = (out >> 20).astype(np.int64)[:, :, 0], ((out >> 10) & 1023).astype(np.int64)[:, :, 0]
    assert np.array_equal(got_y, np.zeros((1024, 1024), np.int64))                       # the obstacle straight "below"
    assert np.array_equal(got_x, np.broadcast_to(np.arange(1024)[:, None], (1024, 1024)))
    assert stats[0] >= 1022 and stats[2] > 900 * 1024   # deque as deep as the column is long: nearly every entry of every column was evicted


def test_wide_packing_reach_of_an_id(model):
    """The kernels' WIDE variant (regions beyond 1024 voxels: ids are stored modulo 1024 and reach 512 voxels): offsets
    instead of absolute sites, candidates out of reach dropped on arrival, lanes that never see a site in reach, and the
    cap d^2 < 2^18 -- against brute force on a 1100 x 24 x 700 slab (the first and the last axis exceed the reach)."""
    from scipy.spatial import cKDTree
    rng = np.random.RandomState(11)
    shape = (1100, 24, 700)
    occ = np.zeros(shape, np.uint8)
    pts = np.stack([rng.randint(0, 1100, 40), rng.randint(0, 24, 40), rng.randint(0, 40, 40)], -1)   # all near z = 0
    pts = np.concatenate([pts, [[5, 3, 699]]])                                                        # and one far corner
    occ[tuple(pts.T)] = 1
    out = np.empty(shape, np.uint32)
    stats = np.zeros(3, np.int32)
    o = np.ascontiguousarray(occ)
    rc = model.ft_model_run_wide(o.ctypes.data, *shape, 64, out.ctypes.data, stats.ctypes.data)
    assert rc == 0, rc
    g = np.stack(np.meshgrid(*[np.arange(n) for n in shape], indexing="ij"), -1).reshape(-1, 3)
    d, _ = cKDTree(np.argwhere(occ)).query(g)
    d2 = np.rint(d ** 2).astype(np.int64)
    want = np.where(d2 < (1 << 18), d2, 0x7FFFFFFF).reshape(shape)
    assert np.array_equal(out.astype(np.int64), want)
    assert (want == 0x7FFFFFFF).any() and (want < 0x7FFFFFFF).any()
