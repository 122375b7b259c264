"""GPU: the INCREMENTAL cell transform (nn_kernels.hpp: k_nn_mark / k_nn_lists_dirty / k_nn_fill_dirty) -- on a fully observed map
whose last UpdateESDF was a cell transform, a small delta redoes only the cells whose search window holds a changed voxel
(the reference's own cost follows the delta's Voronoi cells, src/ESDFMap.cpp:273-337).  Exactness is the transform's: squared
distances equal to the reference's on every voxel, whatever mix of incremental and full updates produced the field."""
import numpy as np
import pytest

from scenarios import P_DEFAULT, Both, all_voxels, assert_exact, compare_dense

pytestmark = pytest.mark.gpu


def _both(oracle_libs, kind, n, engine="cells"):
    import fiesta_amd
    res = 0.1
    gpu = fiesta_amd.ESDFMap((0, 0, 0), res, ((n - 0.5) * res,) * 3, update_engine=engine)
    cpu = oracle_libs.OracleMap((0, 0, 0), res, ((n - 0.5) * res,) * 3, kind=kind)
    b = Both(gpu, cpu)
    b.params()
    b.observe(all_voxels(n), 0)
    b.fuse()
    b.esdf()
    return b


def test_incremental_and_full_updates_interleaved_match_the_reference(hip_lib, oracle_libs, best_oracle_kind):
    n = 160
    b = _both(oracle_libs, best_oracle_kind, n)
    rng = np.random.RandomState(7)
    V = all_voxels(n)
    live = V[rng.choice(len(V), 400, replace=False)]
    b.make_occupied(live)
    sg, _ = b.esdf()
    assert sg["cells"] == 1 and sg["nn_incremental"] == 0, sg
    assert_exact(compare_dense(b.gpu, b.cpu))
    seen_inc = seen_full = 0
    for step in range(10):
        k = [2, 5, 1, 40, 3, 7, 1, 200, 4, 2][step]   # small deltas run incrementally, the large ones dirty too much and run in full
        new = V[rng.choice(len(V), k, replace=False)]
        old, live = live[:k], np.concatenate([live[k:], new])
        b.mixed(new, old)
        sg, _ = b.esdf()
        assert sg["cells"] == 1 and sg["nn_failed"] == 0, sg
        seen_inc += int(sg["nn_incremental"])
        seen_full += int(not sg["nn_incremental"])
        if sg["nn_incremental"]:
            assert 0 < sg["nn_dirty_cells"] < (n // 8) ** 3, sg
        assert_exact(compare_dense(b.gpu, b.cpu))
    assert seen_inc >= 6 and seen_full >= 1, (seen_inc, seen_full)
    b.gpu.close()
    b.cpu.close()


def test_incremental_delete_of_the_only_obstacle_in_reach_falls_back(hip_lib, oracle_libs, best_oracle_kind):
    """Deleting obstacles until some cell finds nothing within its widest window: the incremental attempt fails (a cell without a
    list), the same call is served in full -- by the envelope passes, since the cell transform cannot serve such a scene either."""
    n = 128
    b = _both(oracle_libs, best_oracle_kind, n)
    rng = np.random.RandomState(3)
    V = all_voxels(n)
    S = V[rng.choice(len(V), 500, replace=False)]
    b.make_occupied(S)
    sg, _ = b.esdf()
    assert sg["cells"] == 1, sg
    far = S[(S[:, 0] > 40)]          # free one side of the map completely
    for s in range(0, len(far), 60):
        b.make_free(far[s:s + 60])
        sg, _ = b.esdf()
        assert sg["bulk"] == 1, sg
        assert_exact(compare_dense(b.gpu, b.cpu))
    b.gpu.close()
    b.cpu.close()


def test_other_engines_invalidate_the_lists(hip_lib, oracle_libs, best_oracle_kind):
    n = 128
    b = _both(oracle_libs, best_oracle_kind, n, engine="auto")
    rng = np.random.RandomState(9)
    V = all_voxels(n)
    S = V[rng.choice(len(V), 800, replace=False)]
    b.make_occupied(S)
    sg, _ = b.esdf()
    assert sg["cells"] == 1, sg
    b.gpu.set_update_engine("rounds")
    b.mixed(V[rng.choice(len(V), 3, replace=False)], S[:3])
    sg, _ = b.esdf()
    assert sg["bulk"] == 0, sg
    b.gpu.set_update_engine("cells")
    b.mixed(V[rng.choice(len(V), 3, replace=False)], S[3:6])
    sg, _ = b.esdf()
    assert sg["cells"] == 1 and sg["nn_incremental"] == 0, sg   # (the rounds changed the field behind the lists' back)
    assert_exact(compare_dense(b.gpu, b.cpu))
    b.mixed(V[rng.choice(len(V), 3, replace=False)], S[6:9])
    b.gpu.snapshot_save(0)
    sg, _ = b.esdf()
    assert sg["cells"] == 1 and sg["nn_incremental"] == 1, sg
    assert_exact(compare_dense(b.gpu, b.cpu))
    b.gpu.snapshot_restore(0)
    sg = b.gpu.UpdateESDF()
    assert sg["cells"] == 1 and sg["nn_incremental"] == 0, sg   # (a restored map: whatever the lists describe, it is not this field)
    assert_exact(compare_dense(b.gpu, b.cpu))
    b.gpu.close()
    b.cpu.close()


@pytest.mark.parametrize("shape,seed", [((161, 45, 83), 1), ((72, 200, 40), 2), ((130, 66, 97), 3)])
def test_incremental_updates_on_ragged_maps_are_exact(hip_lib, shape, seed):
    """grids that are no multiple of the cell edge: a run of small deltas, each served incrementally where the lists allow, every
    field the exact transform of what is occupied (scipy)"""
    from test_gpu_cells import check_exact, free, make_map, occupy
    rng = np.random.RandomState(seed)
    m = make_map(shape, "cells")
    V = all_voxels(shape)
    live = V[rng.choice(len(V), len(V) // 2500, replace=False)]
    occupy(m, live)
    st = m.UpdateESDF()
    assert st["cells"] == 1, st
    check_exact(m, shape)
    inc = 0
    for step in range(6):
        k = 1   # (one insert + one delete: these maps have ~10^3 cells, a voxel dirties ~10^2 of them)
        new = V[rng.choice(len(V), k, replace=False)]
        occupy(m, new)
        free(m, live[:k])
        live = np.concatenate([live[k:], new])
        st = m.UpdateESDF()
        assert st["bulk"] == 1, st
        inc += int(st["nn_incremental"])
        check_exact(m, shape)
    assert inc >= 3, inc
    m.close()


def test_incremental_update_with_many_dirty_cells_is_exact(hip_lib):
    """a map large enough for a three-digit delta to stay incremental: tens of thousands of dirty cells, k_nn_mark's work-groups
    flush their LDS queues more than once (one atomic on the list's cursor per flush)"""
    from test_gpu_cells import check_exact, free, make_map, occupy
    shape = (320, 320, 320)
    rng = np.random.RandomState(11)
    m = make_map(shape, "cells")
    n = shape[0] * shape[1] * shape[2]
    pick = lambda k: np.stack(np.unravel_index(rng.choice(n, k, replace=False), shape), 1).astype(np.int32)  # noqa: E731
    live = pick(n // 2700)
    occupy(m, live)
    st = m.UpdateESDF()
    assert st["cells"] == 1, st
    for step in range(2):
        k = 60
        new = pick(k)
        occupy(m, new)
        free(m, live[:k])
        live = np.concatenate([live[k:], new])
        st = m.UpdateESDF()
        assert st["bulk"] == 1 and st["nn_incremental"] == 1, st
        assert st["nn_dirty_cells"] > 5000, st
        check_exact(m, shape)
    m.close()
