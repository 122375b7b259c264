"""The cell transform's arithmetic (fiesta_amd/csrc/nn_core.hpp: which obstacles a cell's list must hold, the search window
that finds them, the keys a voxel minimises) checked on the CPU: tests/cpp/nn_model.cpp drives the very header the HIP
kernels of nn_kernels.hpp are built on, and wherever every cell got its list the result must be the exact Euclidean
feature transform (squared distances equal to scipy's EDT; every closest site occupied).  Cells that cannot be served --
no obstacle within reach, too many candidates -- must say so (the GPU path then runs the envelope passes instead).
No GPU, no oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
from scipy import ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "cpp", "nn_model.cpp")
LIB = os.path.join(HERE, "cpp", "libnn_model.so")
CORE = os.path.join(os.path.dirname(HERE), "fiesta_amd", "csrc", "nn_core.hpp")


@pytest.fixture(scope="module")
def model():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(CORE)):
        subprocess.run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", LIB, SRC], check=True)
    lib = C.CDLL(LIB)
    lib.nn_model_run.restype = C.c_int
    lib.nn_model_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    return lib


def run(lib, occ):
    nx, ny, nz = occ.shape
    occ = np.ascontiguousarray(occ, dtype=np.uint8)
    out = np.full(occ.shape, 0xDEADBEEF, np.uint32)
    stats = np.zeros(4, np.int64)
    rc = lib.nn_model_run(occ.ctypes.data, nx, ny, nz, out.ctypes.data, stats.ctypes.data)
    assert rc == 0, "a key's distance part disagrees with the site it names (1) / the record's count with the return value (2)"
    return out, dict(failed=int(stats[0]), entries=int(stats[1]), longest=int(stats[2]), sites=int(stats[3]))


def check_exact(occ, out, served=None):
    nx, ny, nz = occ.shape
    idx = ndimage.distance_transform_edt(occ == 0, return_distances=False, return_indices=True)
    gx, gy, gz = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    want = (idx[0] - gx) ** 2 + (idx[1] - gy) ** 2 + (idx[2] - gz) ** 2
    sel = np.ones(occ.shape, bool) if served is None else served
    o = out[sel]
    assert np.all(o < 0x40000000)
    cx, cy, cz = (o >> 20).astype(np.int64), ((o >> 10) & 1023).astype(np.int64), (o & 1023).astype(np.int64)
    assert cx.max() < nx and cy.max() < ny and cz.max() < nz
    assert np.all(occ[cx, cy, cz] == 1), "closest site is not occupied"
    got = (cx - gx[sel]) ** 2 + (cy - gy[sel]) ** 2 + (cz - gz[sel]) ** 2
    bad = np.flatnonzero(got != want[sel])
    assert len(bad) == 0, f"{len(bad)} voxels differ from the exact transform"


def scatter(shape, density, seed):
    rng = np.random.RandomState(seed)
    return (rng.rand(*shape) < density).astype(np.uint8)


# (shape, density, seed): ragged extents, the benchmark's density (3.7e-4), denser and sparser neighbours of it
CASES = [((64, 64, 64), 3.7e-4, 1), ((96, 72, 80), 3.7e-4, 2), ((61, 45, 83), 1e-3, 3), ((40, 40, 40), 5e-3, 4),
         ((33, 17, 130), 2e-3, 5), ((128, 128, 128), 3.7e-4, 6), ((24, 100, 9), 3e-3, 7), ((70, 70, 70), 2e-2, 8)]


@pytest.mark.parametrize("shape,density,seed", CASES)
def test_cell_lists_give_the_exact_transform(model, shape, density, seed):
    occ = scatter(shape, density, seed)
    if not occ.any():
        occ[tuple(s // 2 for s in shape)] = 1
    out, st = run(model, occ)
    assert st["sites"] == int(occ.sum())
    ncell = np.prod([(s + 7) // 8 for s in shape])
    if st["failed"] == 0:
        check_exact(occ, out)
    else:  # the served cells are still exact; the others say "no list"
        served = out != 0x80000000
        assert (~served).sum() > 0
        check_exact(occ, out, served)
    print(shape, density, "cells", ncell, st, "mean list", st["entries"] / max(1, ncell - st["failed"]))


def test_benchmark_density_needs_no_fallback(model):
    """config 2's scene at a size the model runs in seconds: every cell must be served, lists short"""
    rng = np.random.RandomState(12345)
    n = 160
    occ = np.zeros((n, n, n), np.uint8)
    v = rng.randint(0, n, (int(50000 * (n / 512.0) ** 3), 3))
    occ[v[:, 0], v[:, 1], v[:, 2]] = 1
    out, st = run(model, occ)
    assert st["failed"] == 0, st
    assert st["longest"] <= 30 and st["entries"] / (n // 8) ** 3 < 9.0, st
    check_exact(occ, out)


def test_cells_that_cannot_be_served_say_so(model):
    """one obstacle in a 96^3 map: cells farther than the window's reach have no list; a wall: more candidates than a list
    holds.  Both must be reported, never answered wrongly."""
    occ = np.zeros((96, 96, 96), np.uint8)
    occ[3, 4, 5] = 1
    out, st = run(model, occ)
    assert st["failed"] > 0
    served = out != 0x80000000
    assert served.sum() > 0
    check_exact(occ, out, served)
    wall = np.zeros((48, 48, 48), np.uint8)
    wall[:, :, 20] = 1
    out, st = run(model, wall)
    assert st["failed"] > 0
    served = out != 0x80000000
    if served.any():
        check_exact(wall, out, served)


def test_ties_and_clusters(model):
    """obstacles on a regular lattice (every voxel between them ties) and a dense cluster inside an empty region"""
    occ = np.zeros((64, 64, 64), np.uint8)
    occ[4::12, 4::12, 4::12] = 1
    out, st = run(model, occ)
    assert st["failed"] == 0
    check_exact(occ, out)
    occ = scatter((64, 64, 64), 4e-4, 21)
    occ[28:33, 30:34, 29:31] = 1
    out, st = run(model, occ)
    served = out != 0x80000000
    check_exact(occ, out, served)


# ---- a shard's region: the array somewhere inside a larger grid, lists from the global bitmap ------------------------------
def run_shard(lib, occ, l0, ln, mc):
    lib.nn_model_run_shard.restype = C.c_int
    lib.nn_model_run_shard.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_void_p, C.c_void_p]
    occ = np.ascontiguousarray(occ, dtype=np.uint8)
    G = np.array(occ.shape, np.int32)
    l0, ln = np.array(l0, np.int32), np.array(ln, np.int32)
    out = np.full(tuple(ln), 0xDEADBEEF, np.uint32)
    stats = np.zeros(4, np.int64)
    rc = lib.nn_model_run_shard(occ.ctypes.data, G.ctypes.data, l0.ctypes.data, ln.ctypes.data, int(mc), out.ctypes.data, stats.ctypes.data)
    assert rc == 0, rc
    return out, dict(failed=int(stats[0]), entries=int(stats[1]), longest=int(stats[2]), sites=int(stats[3]))


def check_shard_exact(occ, out, l0, ln):
    """the array's words against the exact transform of the WHOLE grid's obstacles"""
    idx = ndimage.distance_transform_edt(occ == 0, return_distances=False, return_indices=True)
    sl = tuple(slice(a, a + n) for a, n in zip(l0, ln))
    g = np.meshgrid(*[np.arange(a, a + n) for a, n in zip(l0, ln)], indexing="ij")
    want = sum((idx[k][sl] - g[k]) ** 2 for k in range(3))
    assert np.all(out < 0x40000000)
    # the word names the site modulo 1024 per axis; it is decoded relative to the voxel that holds it (fiesta_amd/csrc/common.hpp:
    # coc_offset) -- the plain coordinate on a grid of at most 1024 voxels
    w = [(out >> 20).astype(np.int64), ((out >> 10) & 1023).astype(np.int64), (out & 1023).astype(np.int64)]
    d = [((g[k] - w[k] + 512) & 1023) - 512 for k in range(3)]
    c = [g[k] - d[k] for k in range(3)]
    assert all((c[k] >= 0).all() and (c[k] < occ.shape[k]).all() for k in range(3))
    assert np.all(occ[c[0], c[1], c[2]] == 1), "closest site is not occupied"
    got = sum(d[k] ** 2 for k in range(3))
    assert int((got != want).sum()) == 0, f"{int((got != want).sum())} voxels differ from the exact transform of the global grid"


@pytest.mark.parametrize("l0,ln", [((0, 0, 0), (66, 66, 66)), ((62, 62, 62), (66, 66, 66)), ((62, 0, 30), (66, 64, 50)),
                                   ((0, 62, 0), (128, 66, 128)), ((29, 35, 41), (37, 30, 51))])
def test_a_shard_inside_a_larger_grid_is_exact_when_the_margin_suffices(model, l0, ln):
    """arrays with ghost layers at odd offsets of a 128^3 grid; obstacles on both sides of every face; a margin that holds
    every search window: no failed cell, and the array equals the transform of the global obstacle set"""
    occ = scatter((128, 128, 128), 1.2e-3, 17)
    out, st = run_shard(model, occ, l0, ln, 40)
    assert st["failed"] == 0, st
    check_shard_exact(occ, out, l0, ln)


def test_a_margin_too_small_fails_cells_instead_of_answering_wrong(model):
    """with 8 voxels of margin the windows of the cells at the open faces leave the region: they must fail (the shard then
    takes the envelope passes, which test their own margin) -- and a face that lies on the grid's own boundary is not open"""
    occ = scatter((128, 128, 128), 1.2e-3, 17)
    out, st = run_shard(model, occ, (62, 62, 62), (66, 66, 66), 8)
    assert st["failed"] > 0
    out, st = run_shard(model, occ, (0, 0, 0), (128, 128, 128), 0)   # the whole grid as ONE shard: nothing is open
    assert st["failed"] == 0
    check_shard_exact(occ, out, (0, 0, 0), (128, 128, 128))


def test_the_nearest_obstacle_beyond_an_open_face(model):
    """an empty half: every obstacle of the array's voxels lies across the face, farther than the margin -> cells fail; with
    the margin grown past them the array is exact"""
    occ = np.zeros((128, 64, 64), np.uint8)
    occ[100:, :, :] = scatter((28, 64, 64), 2e-3, 3)
    l0, ln = (0, 0, 0), (66, 64, 64)
    out, st = run_shard(model, occ, l0, ln, 16)
    assert st["failed"] > 0
    occ[70:, :, :] = scatter((58, 64, 64), 2e-3, 4)
    out, st = run_shard(model, occ, l0, ln, 48)
    if st["failed"] == 0:
        check_shard_exact(occ, out, l0, ln)


def test_a_region_of_more_than_1024_voxels_stores_its_sites_modulo_1024(model):
    """a config-5-shaped shard along x: 1026 voxels of a 2048-long grid from an odd offset, margin on the open side -> a region
    of 1072 voxels; sites and words modulo 1024, decoded relative to the cell / the voxel"""
    occ = scatter((2048, 24, 40), 1.0e-3, 23)
    for l0, ln in (((1022, 0, 0), (1026, 24, 40)), ((0, 0, 0), (1026, 24, 40)), ((500, 0, 0), (1100, 24, 40))):
        out, st = run_shard(model, occ, l0, ln, 40)
        assert st["failed"] == 0, (l0, st)
        check_shard_exact(occ, out, l0, ln)


def test_sparse_scenes_within_seven_cells_of_reach(model):
    """search windows reach 7 cells (nn_core.hpp: kKmax): scenes down to ~8e-5 of the voxels (where dense_map.hip starts to try
    the transform) are served without a failed cell -- the nearest obstacle of a cell's centre may be 50 voxels away -- and
    are exact; at 6e-5 the first cells fail (the GPU path then takes the envelope passes), every served voxel still exact"""
    for dens, seed in ((9e-5, 8), (1.3e-4, 9)):
        occ = scatter((160, 152, 168), dens, seed)
        out, st = run(model, occ)
        assert st["failed"] == 0, (dens, st)
        check_exact(occ, out)
    occ = scatter((160, 152, 168), 6e-5, 7)
    out, st = run(model, occ)
    assert 0 < st["failed"] < 20, st
    check_exact(occ, out, served=out != 0x80000000)


def test_a_shards_region_starts_on_whole_bitmap_words_along_z(model):
    """the region's z-origin is a multiple of 32 (k_nn_cells reads the rows of the replica with dword-aligned wide loads), its
    x / y origins multiples of 8; the array -- here at z = 45 -- is exact all the same"""
    occ = scatter((96, 96, 200), 1.0e-3, 31)
    l0, ln = (30, 22, 45), (40, 50, 90)
    out, st = run_shard(model, occ, l0, ln, 40)
    assert st["failed"] == 0, st
    check_shard_exact(occ, out, l0, ln)
