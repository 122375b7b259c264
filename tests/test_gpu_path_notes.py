"""fiesta_hip_stats.path_notes (include/fiesta_hip.h: FIESTA_HIP_NOTE_*, `why` in the Python statistics): UpdateESDF says WHY it
took the path it took -- the gates and back-offs behind the engine flags (every cliff is a place where the
latency of a call changes by a factor, and nothing told the caller which one it stood at)."""
import numpy as np
import pytest

from scenarios import P_DEFAULT
from test_gpu_cells import free, make_map, occupy

pytestmark = pytest.mark.gpu


def scatter(shape, k, seed):
    rng = np.random.RandomState(seed)
    n = shape[0] * shape[1] * shape[2]
    return np.stack(np.unravel_index(rng.choice(n, k, replace=False), shape), 1).astype(np.int32)


def test_a_fully_observed_sparse_map_has_nothing_to_report(hip_lib):
    shape = (64, 64, 64)
    m = make_map(shape, "auto")
    occupy(m, scatter(shape, 200, 1))
    st = m.UpdateESDF()
    assert st["cells"] == 1 and st["why"] == [] and st["path_notes"] == 0, st
    m.close()


def test_density_outside_the_cell_transforms_range(hip_lib):
    shape = (64, 64, 64)
    m = make_map(shape, "auto")
    occupy(m, scatter(shape, 3000, 2))   # more than a 400th of the voxels
    st = m.UpdateESDF()
    assert st["bulk"] == 1 and st["cells"] == 0 and "density" in st["why"], st
    m.close()


def test_a_pinned_engine_and_a_small_delta(hip_lib):
    shape = (64, 64, 64)
    m = make_map(shape, "rounds")
    occupy(m, scatter(shape, 200, 3))
    st = m.UpdateESDF()
    assert st["bulk"] == 0 and "engine_pinned" in st["why"], st
    m.close()
    m = make_map((128, 128, 128), "auto")
    occupy(m, scatter((128, 128, 128), 1000, 4))
    m.UpdateESDF()
    occupy(m, np.array([[5, 5, 5]], np.int32))
    st = m.UpdateESDF()
    assert ("small_delta" in st["why"]) == (st["bulk"] == 0), st
    m.close()


def test_a_partially_observed_map_and_a_window(hip_lib):
    import fiesta_amd
    shape = (96, 96, 96)
    m = fiesta_amd.ESDFMap((0, 0, 0), 0.1, tuple((s - 0.5) * 0.1 for s in shape), update_engine="auto")
    m.SetParameters(*P_DEFAULT)
    m.SetOriginalRange()
    m.SetOccupancyBox((0, 0, 0), (95, 95, 63), 0)   # a third of the map is never observed
    m.UpdateOccupancy(True)
    m.UpdateESDF()
    S = scatter((96, 96, 64), 700, 5)
    occupy(m, S)
    st = m.UpdateESDF()
    assert "partly_observed" in st["why"] and st["masked"] == 1, st
    # a small delta on the same map: the level engine, and the note says a transform would not have paid
    occupy(m, np.array([[40, 40, 10]], np.int32))
    st = m.UpdateESDF()
    assert "partly_observed" in st["why"] and st["masked"] == 0, st
    # under a partial window; then the window's history stays on record
    m.SetUpdateRange((0.0, 0.0, 0.0), (4.0, 4.0, 4.0))
    occupy(m, np.array([[20, 20, 20]], np.int32))
    st = m.UpdateESDF()
    assert "partial_window" in st["why"], st
    m.SetOriginalRange()
    free(m, S[:300])
    occupy(m, scatter((96, 96, 64), 300, 6))
    st = m.UpdateESDF()
    assert "window_history" in st["why"] and st["masked"] == 0, st
    m.close()
