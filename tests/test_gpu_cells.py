"""The cell transform (fiesta_amd/csrc/nn_kernels.hpp) on the GPU: UpdateESDF on fully observed maps with a
sparse obstacle set, served by per-cell obstacle lists instead of the envelope passes.  Same bar as every fully observed
test: squared distances equal to the reference's / the exact transform's on every voxel, ids tie-equivalent (occupied, at
exactly that distance).  Also: which transform the library picks, that a scene the cell transform cannot serve (a cell
with no obstacle in reach, a wall) is handed to the envelope passes within the same call, and that the tracked distance
bound is maintained.  The CPU model of the same arithmetic is tests/test_nn_model.py."""
import numpy as np
import pytest
from scipy import ndimage

from scenarios import P_DEFAULT, Both, all_voxels, assert_exact, compare_dense

pytestmark = pytest.mark.gpu


def make_map(shape, engine, res=0.1):
    import fiesta_amd
    m = fiesta_amd.ESDFMap((0, 0, 0), res, tuple((s - 0.5) * res for s in shape), update_engine=engine)   # (ceil(size / res) voxels)
    assert tuple(m.grid_size) == tuple(shape)
    m.SetParameters(*P_DEFAULT)
    m.SetOriginalRange()
    m.SetOccupancyBox((0, 0, 0), tuple(s - 1 for s in shape), 0)
    m.UpdateOccupancy(True)
    m.UpdateESDF()
    return m


def occupy(m, vox, cycles=3):
    for _ in range(cycles):
        m.SetOccupancy(np.ascontiguousarray(vox, np.int32), 1, want_ret=False)
        m.UpdateOccupancy(True)


def free(m, vox, cycles=6):
    for _ in range(cycles):
        m.SetOccupancy(np.ascontiguousarray(vox, np.int32), 0, want_ret=False)
        m.UpdateOccupancy(True)


def check_exact(m, shape):
    f = m.download_field(("d2", "coc", "occ"))
    occ = f["occ"].reshape(shape)
    d2 = f["d2"].reshape(shape).astype(np.int64)
    assert occ.any()
    idx = ndimage.distance_transform_edt(occ == 0, return_distances=False, return_indices=True)
    gx, gy, gz = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
    want = (idx[0] - gx) ** 2 + (idx[1] - gy) ** 2 + (idx[2] - gz) ** 2
    bad = int((d2 != want).sum())
    assert bad == 0, f"{bad} voxels differ from the exact transform"
    c = f["coc"].reshape(shape + (3,)).astype(np.int64)
    assert np.all(occ[c[..., 0], c[..., 1], c[..., 2]] == 1), "a closest obstacle is not occupied"
    own = (c[..., 0] - gx) ** 2 + (c[..., 1] - gy) ** 2 + (c[..., 2] - gz) ** 2
    assert np.array_equal(own, want), "an id is not at the stored distance"


SCENES = [((64, 64, 64), 200, 1), ((96, 72, 80), 300, 2), ((61, 45, 83), 230, 3), ((128, 128, 128), 1000, 12345),
          ((33, 17, 130), 140, 5), ((24, 100, 9), 60, 7)]


@pytest.mark.parametrize("shape,k,seed", SCENES)
def test_cell_transform_is_the_exact_transform(hip_lib, shape, k, seed):
    """scatter scenes, ragged extents (cells cut by the array's faces), insert then a mixed insert + delete update"""
    m = make_map(shape, "cells")
    rng = np.random.RandomState(seed)
    S = (rng.randint(0, 1 << 20, (k, 3)) % np.array(shape)).astype(np.int32)
    occupy(m, S)
    st = m.UpdateESDF()
    assert st["bulk"] == 1 and st["cells"] == 1 and st["nn_failed"] == 0, st
    assert st["nn_entries"] > 0
    check_exact(m, shape)
    new = (rng.randint(0, 1 << 20, (k // 2, 3)) % np.array(shape)).astype(np.int32)
    for _ in range(6):
        m.SetOccupancy(new, 1, want_ret=False)
        m.SetOccupancy(S[: k // 2], 0, want_ret=False)
        m.UpdateOccupancy(True)
    st = m.UpdateESDF()
    assert st["deleted"] > 0 and st["inserted"] > 0
    assert st["bulk"] == 1 and st["cells"] == 1, st
    check_exact(m, shape)
    m.close()


def test_cells_against_the_reference_with_queries(hip_lib, oracle_libs, best_oracle_kind):
    """the same flow against the verbatim reference (ids tie-equivalent, log-odds, queues) -- config 1's size and scene"""
    import fiesta_amd
    n, res = 128, 0.1
    gpu = fiesta_amd.ESDFMap((0, 0, 0), res, (n * res,) * 3, update_engine="cells")
    cpu = oracle_libs.OracleMap((0, 0, 0), res, (n * res,) * 3, kind=best_oracle_kind)
    b = Both(gpu, cpu)
    b.params()
    gpu.SetOriginalRange()
    cpu.SetOriginalRange()
    b.observe(all_voxels(n), 0)
    b.fuse()
    b.esdf()
    S = np.random.RandomState(12345).randint(0, n, (1000, 3)).astype(np.int32)
    b.make_occupied(S)
    sg, _ = b.esdf()
    assert sg["cells"] == 1
    assert_exact(compare_dense(gpu, cpu))
    b.make_free(S[:500])
    sg, _ = b.esdf()
    assert sg["cells"] == 1
    assert_exact(compare_dense(gpu, cpu))
    pos = np.random.RandomState(3).uniform(0.2, n * res - 0.2, (2000, 3))
    dg, gg = gpu.GetDistWithGradTrilinear(pos)
    dc, gc = cpu.GetDistWithGradTrilinear(pos)
    assert np.array_equal(dg, dc) and np.array_equal(gg, gc)
    gpu.close()
    cpu.close()


def test_scenes_the_cell_transform_cannot_serve_go_to_the_envelope_passes(hip_lib):
    """one obstacle in a 96^3 map (cells with nothing in reach) and a wall (more candidates than a list holds): the cell
    transform reports failed cells, the SAME UpdateESDF call finishes on the envelope passes, the result is exact; the
    library does not try again at that obstacle count unless pinned"""
    shape = (96, 96, 96)
    m = make_map(shape, "cells")
    occupy(m, np.array([[3, 4, 5]], np.int32))
    st = m.UpdateESDF()
    assert st["bulk"] == 1 and st["cells"] == 0 and st["nn_failed"] > 0, st
    check_exact(m, shape)
    m.close()
    shape = (48, 48, 48)
    m = make_map(shape, "bulk")
    wall = np.array([(x, y, 20) for x in range(48) for y in range(48)], np.int32)[::23]  # 101 voxels of a plane: density in range
    occupy(m, wall)
    st = m.UpdateESDF()
    check_exact(m, shape)
    first_cells = st["cells"]
    occupy(m, np.array([(x, y, 20) for x in range(48) for y in range(48)], np.int32))
    st = m.UpdateESDF()
    assert st["bulk"] == 1
    check_exact(m, shape)
    m.set_update_engine("cells")
    free(m, np.array([[0, 0, 20]], np.int32))
    st = m.UpdateESDF()
    assert st["bulk"] == 1 and st["cells"] == 0 and st["nn_failed"] > 0, (st, first_cells)
    check_exact(m, shape)
    m.close()


def test_which_transform_the_library_picks(hip_lib):
    """density inside the cell transform's range -> cells; outside (a handful of obstacles, or one in forty voxels) -> the
    envelope passes; engine "envelope" never runs the cell transform; all three fields identical"""
    shape = (80, 80, 80)
    rng = np.random.RandomState(9)
    S = rng.randint(0, 80, (300, 3)).astype(np.int32)   # 5.9e-4 of the voxels
    fields = []
    for engine, want_cells in (("bulk", 1), ("envelope", 0), ("cells", 1), ("auto", 1)):
        m = make_map(shape, engine)
        occupy(m, S)
        st = m.UpdateESDF()
        assert st["bulk"] == 1 and st["cells"] == want_cells, (engine, st)
        fields.append(m.download_field(("d2",))["d2"].copy())
        m.close()
    for f in fields[1:]:
        assert np.array_equal(f, fields[0])
    m = make_map(shape, "bulk")
    occupy(m, S[:5])                                   # 1e-5: nothing within a cell's reach almost everywhere
    st = m.UpdateESDF()
    assert st["bulk"] == 1 and st["cells"] == 0 and st["nn_failed"] == 0, st   # (not even tried)
    check_exact(m, shape)
    dense = rng.randint(0, 80, (20000, 3)).astype(np.int32)  # 3.8e-2
    occupy(m, dense)
    st = m.UpdateESDF()
    assert st["bulk"] == 1 and st["cells"] == 0 and st["nn_failed"] == 0, st
    check_exact(m, shape)
    m.close()


def test_cell_transform_keeps_the_tracked_distance_bound(hip_lib):
    """a map that has seen a ray-cast frame tracks the largest stored distance (the delete scan of the other engines is
    bounded by it): after a cell transform the bound must cover the field, or a later small delete misses orphans"""
    import fiesta_amd
    shape = (64, 64, 64)
    m = make_map(shape, "cells")
    # one ray-cast frame far from everything switches the tracking on (raycast.hip: enable_distance_tracking)
    T = np.eye(4)
    T[:3, 3] = (3.2, 3.2, 3.2)
    pts = np.array([[0.3, 0.0, 0.0]], np.float32)
    m.RaycastFrame(pts, T, (3.2, 3.2, 3.2), 0.05, 5.0, (-100.0,) * 3, (100.0,) * 3)
    m.UpdateOccupancy(True)
    m.UpdateESDF()
    rng = np.random.RandomState(4)
    S = rng.randint(0, 64, (200, 3)).astype(np.int32)
    occupy(m, S)
    st = m.UpdateESDF()
    assert st["cells"] == 1, st
    check_exact(m, shape)
    # a small delete on the rounds / level engine: its scan must reach every voxel that pointed at the deleted obstacles
    m.set_update_engine("rounds")
    free(m, S[:3])
    st = m.UpdateESDF()
    assert st["bulk"] == 0
    check_exact(m, shape)
    m.close()


@pytest.mark.parametrize("n_shards", [1, 2, 4, 8])
def test_cell_transform_on_shards(hip_lib, oracle_libs, best_oracle_kind, n_shards):
    """One grid cut into shards (multiplexed on the one GPU): every shard runs the cell transform on its own array -- owned
    box + ghost layers, at odd offsets of the global grid -- reading the replica of the global bitmap, its region grown by the
    group's margin; no exchange.  Obstacles hug the cuts, so nearest obstacles lie across them.  Same bar as unsharded: the
    assembled field equals the reference's on every voxel; and the shards did run the cell transform."""
    from fiesta_amd.sharded import ShardedESDFMap
    from test_gpu_sharded import compare, drive
    gs, res = (136, 128, 144), 0.1
    sm = ShardedESDFMap((0, 0, 0), res, gs, n_shards, native=True, update_engine="cells")
    cpu = oracle_libs.OracleMap((0, 0, 0), res, tuple((np.array(gs) - 0.5) * res), kind=best_oracle_kind)
    assert cpu.grid_size == gs
    for m in (sm, cpu):
        m.SetParameters(*P_DEFAULT)
        m.SetOriginalRange()
    allv = all_voxels(gs)
    rng = np.random.RandomState(31)
    S = (rng.rand(2400, 3) * gs).astype(np.int32)
    S[:200, 0] = rng.randint(66, 70, 200)   # shard faces at 68 / 64 / 72
    S[200:400, 1] = rng.randint(62, 66, 200)
    S[400:600, 2] = rng.randint(70, 74, 200)
    drive(sm, cpu, [([], allv, 1)])
    for occ, free, cycles in ((S, [], 3), ((rng.rand(700, 3) * gs).astype(np.int32), S[:1200], 6)):
        for _ in range(cycles):
            for v, o in ((occ, 1), (free, 0)):
                if len(v):
                    sm.SetOccupancy(v, o)
                    cpu.SetOccupancyVox(v, o)
            assert sm.UpdateOccupancy(True) == cpu.UpdateOccupancy(True)
        sg, sc = sm.UpdateESDF(), cpu.UpdateESDF()
        assert (sg["inserted"], sg["deleted"]) == (sc["inserted"], sc["deleted"])
        assert sg["bulk"] == 1 and sg["cells"] == 1 and sg["nn_failed"] == 0, sg
        compare(sm, cpu, gs)
    sm.close()


def test_cell_transform_on_config5_shaped_shards(hip_lib):
    """a 2048-long grid cut into two shards of 1024 (+ ghost layers): ids modulo 1024 in the voxel words (common.hpp: pack_coc),
    and a REGION of more than 1024 voxels (array + margin), whose sites the cell transform stores modulo 1024 too
    (nn_core.hpp: site_offset).  Obstacles along the whole length, around the cut / the wrap of the ids at x = 1024 and at both
    ends; insert, then a mixed update -- squared distances and decoded obstacles of 60 000 voxels against brute force."""
    from fiesta_amd.sharded import ShardedESDFMap
    from test_gpu_sharded import _brute_d2
    gs, res = (2048, 40, 48), 0.1
    sm = ShardedESDFMap((0, 0, 0), res, gs, 2, native=True, update_engine="cells")
    sm.SetParameters(*P_DEFAULT)
    sm.SetOriginalRange()
    sm.SetOccupancyBox((0, 0, 0), tuple(np.array(gs) - 1), 0)
    sm.UpdateOccupancy(True)
    sm.UpdateESDF()
    rng = np.random.RandomState(19)
    S = (rng.rand(3000, 3) * gs).astype(np.int32)
    S[:200, 0] = rng.randint(1010, 1040, 200)
    S[200:240, 0] = rng.randint(0, 6, 40)
    S[240:280, 0] = rng.randint(2042, 2048, 40)
    S = np.unique(S, axis=0)

    def check(obs):
        f = sm.assemble()
        assert int(f["occ"].sum()) == len(obs)
        idx = np.random.RandomState(2).randint(0, gs[0] * gs[1] * gs[2], 60000).astype(np.int64)
        idx = np.concatenate([idx, (np.arange(1000, 1048)[:, None] * gs[1] * gs[2] + np.arange(0, gs[1] * gs[2], 37)[None, :]).reshape(-1)])
        V = np.stack([idx // (gs[1] * gs[2]), (idx // gs[2]) % gs[1], idx % gs[2]], -1)
        want = _brute_d2(V, obs.astype(np.int64))
        got = f["d2"][idx].astype(np.int64)
        assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]
        c = f["coc"][idx].astype(np.int64)
        assert np.array_equal(((V - c) ** 2).sum(-1), want)
        assert np.all(f["occ"][(c[:, 0] * gs[1] + c[:, 1]) * gs[2] + c[:, 2]] == 1)

    for _ in range(3):
        sm.SetOccupancy(S, 1)
        sm.UpdateOccupancy(True)
    st = sm.UpdateESDF()
    assert st["inserted"] == len(S) and st["bulk"] == 1 and st["cells"] == 1 and st["nn_failed"] == 0, st
    check(S)
    gone, new = S[::2], (rng.rand(800, 3) * gs).astype(np.int32)
    for _ in range(6):
        sm.SetOccupancy(new, 1)
        sm.SetOccupancy(gone, 0)
        sm.UpdateOccupancy(True)
    st = sm.UpdateESDF()
    assert st["bulk"] == 1 and st["cells"] == 1, st
    check(np.unique(np.concatenate([S[1::2], new]), axis=0))
    sm.close()


def test_a_shard_whose_cells_fail_takes_the_envelope_passes(hip_lib, oracle_libs, best_oracle_kind):
    """two shards, every obstacle in the far half of ONE of them: the other shard's nearest obstacles lie across the cut, farther
    than its region's margin -- its cells find nothing in reach or a window that touches the open face, the cell transform
    fails there (nn_failed), that shard runs the envelope passes (which grow their margin until it suffices), the other one
    may keep the cell transform; the assembled field equals the reference's"""
    from fiesta_amd.sharded import ShardedESDFMap
    from test_gpu_sharded import compare, drive
    gs, res = (256, 48, 56), 0.1
    sm = ShardedESDFMap((0, 0, 0), res, gs, 2, native=True, update_engine="cells")
    cpu = oracle_libs.OracleMap((0, 0, 0), res, tuple((np.array(gs) - 0.5) * res), kind=best_oracle_kind)
    for m in (sm, cpu):
        m.SetParameters(*P_DEFAULT)
        m.SetOriginalRange()
    rng = np.random.RandomState(8)
    S = (rng.rand(900, 3) * gs).astype(np.int32)
    S[:, 0] = 200 + S[:, 0] % 56            # x in [200, 256): the left shard (x < 128) is empty, 72+ voxels from the first obstacle
    drive(sm, cpu, [([], all_voxels(gs), 1)])
    for _ in range(3):
        sm.SetOccupancy(S, 1)
        cpu.SetOccupancyVox(S, 1)
        assert sm.UpdateOccupancy(True) == cpu.UpdateOccupancy(True)
    sg, sc = sm.UpdateESDF(), cpu.UpdateESDF()
    assert sg["inserted"] == sc["inserted"] and sg["bulk"] == 1 and sg["cells"] == 0, sg   # (not EVERY shard ran the cell transform)
    compare(sm, cpu, gs)
    sm.close()


def test_cell_transform_on_an_unsharded_map_beyond_1024_voxels(hip_lib):
    """one map of 1400 x 40 x 48 voxels: ids modulo 1024, decoded relative to the voxel (common.hpp), and the cell transform's
    sites likewise relative to the asking cell -- every voxel against scipy's exact transform, insert and mixed update"""
    shape = (1400, 40, 48)
    m = make_map(shape, "cells")
    rng = np.random.RandomState(41)
    S = (rng.randint(0, 1 << 20, (2600, 3)) % np.array(shape)).astype(np.int32)
    S[:150, 0] = rng.randint(1000, 1050, 150)    # around the wrap of the ids
    occupy(m, S)
    st = m.UpdateESDF()
    assert st["bulk"] == 1 and st["cells"] == 1 and st["nn_failed"] == 0, st
    check_exact(m, shape)
    new = (rng.randint(0, 1 << 20, (900, 3)) % np.array(shape)).astype(np.int32)
    for _ in range(6):
        m.SetOccupancy(new, 1, want_ret=False)
        m.SetOccupancy(S[:1300], 0, want_ret=False)
        m.UpdateOccupancy(True)
    st = m.UpdateESDF()
    assert st["bulk"] == 1 and st["cells"] == 1, st
    check_exact(m, shape)
    m.close()


def test_a_scene_the_cell_transform_cannot_serve_is_not_retried_at_once(hip_lib):
    """the library's own choice (`bulk`: a transform whenever the gate is open): 216 obstacles in one corner of a 96^3 map are in
    the density range of the cell transform, but most cells have nothing within reach -- the first update tries it, counts
    the failed cells and finishes on the envelope passes; the next 8 eligible updates do not try (no failed cell reported,
    no time lost), the one after them does again"""
    shape = (96, 96, 96)
    m = make_map(shape, "bulk")
    cube = np.array([(x, y, z) for x in range(6) for y in range(6) for z in range(6)], np.int32) + 3
    occupy(m, cube)
    st = m.UpdateESDF()
    assert st["bulk"] == 1 and st["cells"] == 0 and st["nn_failed"] > 0, st
    check_exact(m, shape)
    flip = np.array([[80, 80, 80]], np.int32)
    tried = []
    for i in range(10):
        (occupy if i % 2 == 0 else free)(m, flip, 6)
        st = m.UpdateESDF()
        assert st["bulk"] == 1 and st["cells"] == 0 and st["inserted"] + st["deleted"] == 1, st
        tried.append(st["nn_failed"] > 0)
    assert tried == [False] * 8 + [True] + [False], tried
    check_exact(m, shape)
    m.close()


def test_a_few_cells_without_a_list_are_served_one_by_one(hip_lib):
    """r06: a cell that cannot be listed costs THAT cell, not the transform.  A scatter scene with (a) an
    empty corner -- the corner's cells find nothing within their widest window -- and (b) a solid block of obstacles -- the cells
    next to it have more survivors than a list holds: the cell transform serves the update, those few cells by brute force
    (against every site / against their window), and the field is the exact transform on every voxel."""
    shape = (128, 128, 128)
    rng = np.random.RandomState(31)
    S = rng.randint(0, 128, (700, 3)).astype(np.int32)
    S = S[~np.all(S < 70, axis=1)]                      # (a) nothing within 70 voxels of the corner (0, 0, 0)
    m = make_map(shape, "cells")
    occupy(m, S)
    st = m.UpdateESDF()
    assert st["bulk"] == 1 and st["cells"] == 1 and st["nn_failed"] == 0 and 0 < st["nn_brute_cells"] <= 64, st
    check_exact(m, shape)
    sparse_cells = st["nn_brute_cells"]
    cube = np.array([(x, y, z) for x in range(7) for y in range(7) for z in range(7)], np.int32) + 90   # (b)
    occupy(m, cube)
    st = m.UpdateESDF()
    assert st["bulk"] == 1 and st["cells"] == 1 and st["nn_failed"] == 0 and st["nn_brute_cells"] > sparse_cells, st
    check_exact(m, shape)
    # a small delta next: the incremental transform meets the cells without a list again -- it fails on them, the same call runs
    # in full (brute force included), and the one after that is incremental again if nothing of it touches such a cell
    free(m, S[:1])
    st = m.UpdateESDF()
    assert st["bulk"] == 1 and st["cells"] == 1 and st["nn_failed"] == 0, st
    check_exact(m, shape)
    m.close()
