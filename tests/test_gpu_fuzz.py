"""Randomised differential test: random operation sequences against the CPU oracle (verbatim reference when built).

Every seed draws a grid shape (ragged sizes included), a sequence of observation batches (hits, misses, duplicates,
voxels outside the map, position- and voxel-addressed), UpdateOccupancy / UpdateESDF calls in varying rhythm, and
queries.  On a fully observed grid the distance field must be bit-identical after every UpdateESDF; return values,
queue sizes, log-odds and batched queries always.
"""
import numpy as np
import pytest

from scenarios import P_DEFAULT, Both, EnvelopeOracle, all_voxels, assert_envelope, assert_exact, compare_dense

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("engine")]


def _make(oracle_libs, kind, size_vox, res, origin, envelope=0):
    import fiesta_amd
    size = tuple((np.array(size_vox) - 0.5) * res)          # ceil(size/res) == size_vox (src/ESDFMap.cpp:175-176)
    gpu = fiesta_amd.ESDFMap(origin, res, size)
    mk = lambda: oracle_libs.OracleMap(origin, res, size, kind=kind)   # noqa: E731
    cpu = EnvelopeOracle(mk, k=envelope) if envelope else mk()        # (K shuffled-order replays of the reference)
    assert gpu.grid_size == tuple(size_vox) == cpu.grid_size
    b = Both(gpu, cpu)
    b.params(P_DEFAULT)
    gpu.SetOriginalRange()
    cpu.SetOriginalRange()
    return b


@pytest.mark.parametrize("seed", list(range(11, 27)))
def test_random_sequences_fully_observed(hip_lib, oracle_libs, best_oracle_kind, seed):
    rng = np.random.RandomState(seed)
    dims = tuple(int(v) for v in rng.randint(9, 44, 3))
    res = float(rng.choice([0.05, 0.1, 0.25]))
    origin = tuple(float(v) for v in rng.uniform(-3, 3, 3))
    b = _make(oracle_libs, best_oracle_kind, dims, res, origin)
    b.observe(all_voxels(dims), 0)
    b.fuse()
    b.esdf()
    lo, hi = np.zeros(3, int), np.array(dims)
    live = np.zeros((0, 3), np.int32)
    for step in range(int(rng.randint(5, 9))):
        n_new = int(rng.randint(1, 60))
        new = np.stack([rng.randint(lo[k] - 2, hi[k] + 2, n_new) for k in range(3)], -1).astype(np.int32)  # some outside
        gone = live[rng.rand(len(live)) < 0.3]
        cycles = int(rng.choice([1, 3, 3, 6]))
        for _ in range(cycles):
            if rng.rand() < 0.5:
                b.observe(new, 1)
            else:                                            # the same batch by position (voxel centres, jittered)
                pos = (new + 0.5 + rng.uniform(-0.3, 0.3, new.shape)) * res + np.array(origin)
                b.observe_pos(pos, 1)
            if len(gone):
                b.observe(gone, 0)
            if rng.rand() < 0.3:
                b.observe(np.concatenate([new[: n_new // 2], new[: n_new // 3]]), int(rng.rand() < 0.5))  # duplicates
            b.fuse()
            if rng.rand() < 0.25:
                b.esdf()                                     # UpdateESDF between ingest cycles as well
        sg, sc = b.esdf()
        assert (sg["inserted"], sg["deleted"]) == (sc["inserted"], sc["deleted"])
        assert_exact(compare_dense(b.gpu, b.cpu))
        inside = np.all((new >= lo) & (new < hi), axis=1)
        keep = set(map(tuple, live.tolist())) - set(map(tuple, gone.tolist())) | set(map(tuple, new[inside].tolist()))
        live = np.array(sorted(keep), np.int32).reshape(-1, 3)
        # batched queries, bit-exact (incl. positions outside the map)
        q = rng.uniform(-1.0, 1.0, (200, 3)) * (np.array(dims) * res * 0.7) + np.array(origin) + np.array(dims) * res / 2
        assert np.array_equal(b.gpu.GetDistance(q), b.cpu.GetDistancePos(q))
        # trilinear: inside the map at least one voxel away from the faces, or outside the map (-1). In the face layer the
        # reference reads past its arrays, which nothing can be compared with.
        ext = np.array(dims) * res
        rel = q - np.array(origin)
        safe = np.all((rel > 1.0 * res) & (rel < ext - 1.0 * res), axis=1) | np.any((rel < -res) | (rel > ext + res), axis=1)
        dg, gg = b.gpu.GetDistWithGradTrilinear(q[safe])
        dc, gc = b.cpu.GetDistWithGradTrilinear(q[safe])
        assert safe.sum() > 50 and np.array_equal(dg, dc) and np.array_equal(gg, gc)
        assert np.array_equal(b.gpu.GetOccupancy(q), b.cpu.GetOccupancyPos(q))


@pytest.mark.parametrize("seed", list(range(41, 49)))
def test_random_sequences_partially_observed(hip_lib, oracle_libs, best_oracle_kind, seed):
    """Only random boxes are ever observed: the reference's result depends on its queue order there,
    so distances are judged against the envelope of the reference's own shuffled runs of the same sequence
    (scenarios.EnvelopeOracle); occupancy, log-odds, queue sizes and the observed set stay exact."""
    rng = np.random.RandomState(seed)
    dims = tuple(int(v) for v in rng.randint(20, 40, 3))
    b = _make(oracle_libs, best_oracle_kind, dims, 0.1, (0.0, 0.0, 0.0), envelope=6)
    for step in range(6):
        c0 = np.array([rng.randint(0, d - 8) for d in dims])
        ext = rng.randint(6, 16, 3)
        box = all_voxels(tuple(int(v) for v in ext)) + c0.astype(np.int32)
        box = box[np.all(box < np.array(dims), axis=1)]
        occ = box[rng.rand(len(box)) < 0.02]
        for _ in range(3):
            b.observe(box, 0)
            if len(occ):
                b.observe(occ, 1)
            b.fuse()
        b.esdf()
        rep = compare_dense(b.gpu, b.cpu)
        assert_envelope(rep, f"step {step}", strict=b.only_levels)
        assert rep["pair_violations"] == 0, rep


@pytest.mark.parametrize("seed", [61, 62, 63, 64])
def test_random_sequences_hash_map(hip_lib, oracle_libs, best_oracle_kind, seed):
    """The same on the paged (hash-block) map against the -DHASH_TABLE reference: random boxes around a wandering centre
    (negative coordinates, page growth from a tiny reserve), obstacles that come and go."""
    import fiesta_amd
    from test_gpu_hash_parity import compare as compare_hash
    kind = best_oracle_kind if oracle_libs.available(best_oracle_kind, "hash") else "port"
    rng = np.random.RandomState(seed)
    origin, res = tuple(float(v) for v in rng.uniform(-1, 1, 3)), float(rng.choice([0.05, 0.1]))
    gpu = fiesta_amd.ESDFMap(origin, res, reserve_size=int(rng.choice([0, 1000, 50000])), mode="hash")
    cpu = EnvelopeOracle(lambda: oracle_libs.OracleMap(origin, res, reserve_size=1000, mode="hash", kind=kind), k=6)
    for m in (gpu, cpu):
        m.SetParameters(*P_DEFAULT)
        m.SetOriginalRange()
    centre = rng.randint(-30, 30, 3)
    live = np.zeros((0, 3), np.int32)
    for step in range(5):
        centre = centre + rng.randint(-6, 7, 3)
        ext = rng.randint(8, 22, 3)
        box = (all_voxels(tuple(int(v) for v in ext)) + (centre - ext // 2)).astype(np.int32)
        new = box[rng.rand(len(box)) < 0.01]
        gone = live[rng.rand(len(live)) < 0.4]
        for k in range(3):
            if k == 0:
                gpu.SetOccupancy(box, 0)
                cpu.SetOccupancyVox(box, 0)
            for vv, o in ((new, 1), (gone, 0)):
                if len(vv):
                    gpu.SetOccupancy(vv, o)
                    cpu.SetOccupancyVox(vv, o)
            a, c = gpu.UpdateOccupancy(True), cpu.UpdateOccupancy(True)
            assert a == c and (gpu.last_insert, gpu.last_delete) == (cpu.last_insert, cpu.last_delete)
        sg, sc = gpu.UpdateESDF(), cpu.UpdateESDF()
        assert (sg["inserted"], sg["deleted"]) == (sc["inserted"], sc["deleted"])
        rep = compare_hash(gpu, cpu)
        # (not strict, whatever the engine: on seed 63, step 4 -- 4 % observed, propagation through channels a voxel wide -- the
        #  frontier rounds, and the level engine as long as its orphans waited in level 0 for their first pull, end 44 voxels
        #  CLOSER than seven shuffled runs of the reference that agree with each other, by 1-11 in d^2 at distances of 7-14
        #  voxels, both sides above the exact distance: the reference floods the dead cells during its list walk, a layered
        #  schedule lets another obstacle through the channel first.  With the list walk ahead of level 0 (k_level_fill) the
        #  pinned level engine is at 0 here; `auto` and the rounds serve this 59-delete update with the frontier rounds and stay
        #  at 44.  tests/test_levelsync_model.py pins both numbers on the CPU model.)
        assert_envelope(rep, f"step {step}")
        live = np.concatenate([live, new])
        q = (rng.uniform(-25, 25, (150, 3)) + centre) * res + np.array(origin)
        assert np.array_equal(gpu.GetOccupancy(q), cpu.GetOccupancyPos(q))
