"""Edge cases of the boundary, through the C ABI on the GPU: empty batches, the largest axis the 30-bit obstacle id
allows, rejected configurations, out-of-map input, repeated / no-op calls (the reference's conventions: -10000, -1,
+10000 sentinels; errors as status codes, never exceptions across the ABI)."""
import ctypes as C

import numpy as np
import pytest

from scenarios import P_DEFAULT

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("engine")]


def test_empty_batches_and_noop_updates(hip_lib):
    import fiesta_amd
    m = fiesta_amd.ESDFMap((0, 0, 0), 0.1, (1.6, 1.6, 1.6))
    m.SetParameters(*P_DEFAULT)
    assert m.CheckUpdate() is False
    assert m.UpdateOccupancy(True) is False and (m.last_insert, m.last_delete) == (0, 0)
    st = m.UpdateESDF()
    assert st["inserted"] == st["deleted"] == st["rounds"] == 0
    e3i, e3d = np.empty((0, 3), np.int32), np.empty((0, 3), np.float64)
    assert len(m.SetOccupancy(e3i, 1)) == 0 and len(m.SetOccupancy(e3d, 0)) == 0
    assert len(m.GetDistance(e3i)) == 0 and len(m.GetDistance(e3d)) == 0 and len(m.GetOccupancy(e3i)) == 0
    d, g = m.GetDistWithGradTrilinear(e3d)
    assert len(d) == 0 and g.shape == (0, 3)
    m.RaycastFrame(np.empty((0, 3), np.float32), np.eye(4), (0, 0, 0), 0.5, 5.0, (0, 0, 0), (1.6, 1.6, 1.6))
    assert m.CheckUpdate() is False
    assert len(m.GetOccupiedVoxels()) == 0
    # never observed: distance reads +10000 (GetDistance(Vector3i), src/ESDFMap.cpp:477-479), occupancy 0
    assert m.GetDistance(np.array([[3, 3, 3]], np.int32))[0] == 10000.0
    assert m.GetOccupancy(np.array([[3, 3, 3]], np.int32))[0] == 0
    # everything outside the map / invalid occ: -10000, nothing queued
    r = m.SetOccupancy(np.array([[9.0, 0.5, 0.5], [-1.0, 0.5, 0.5]]), 1)
    assert list(r) == [-10000, -10000] and m.CheckUpdate() is False
    assert m.SetOccupancy(np.array([[0.5, 0.5, 0.5]]), 7)[0] == -10000 and m.CheckUpdate() is False
    m.close()
    m.close()  # idempotent


def test_largest_axis_and_rejected_configs(hip_lib):
    import fiesta_amd
    m = fiesta_amd.ESDFMap((0, 0, 0), 0.1, (102.35, 0.75, 0.75))  # 1024 x 8 x 8: the largest axis with plain 10-bit ids
    assert m.grid_size == (1024, 8, 8)
    m.SetParameters(*P_DEFAULT)
    m.SetOccupancyBox((0, 0, 0), (1023, 7, 7), 0)
    m.UpdateOccupancy(True)
    ends = np.array([[0, 3, 3], [1023, 4, 4]], np.int32)
    for _ in range(3):
        m.SetOccupancy(ends, 1)
        m.UpdateOccupancy(True)
    st = m.UpdateESDF()
    assert st["inserted"] == 2
    f = m.download_field(("d2", "coc"))
    x = np.arange(1024)
    d2 = f["d2"].reshape(1024, 8, 8)
    want = np.minimum(x ** 2 + 0, (1023 - x) ** 2 + 1 + 1)  # voxel (x,3,3): to (0,3,3) or to (1023,4,4)
    assert np.array_equal(d2[:, 3, 3], want)
    assert f["coc"].reshape(1024, 8, 8, 3)[1000, 4, 4].tolist() == [1023, 4, 4]
    m.close()
    # 1025 voxels: beyond the plain 10-bit id; such maps decode ids relative to their voxel (tests/test_gpu_sharded.py)
    big = fiesta_amd.ESDFMap((0, 0, 0), 0.1, (102.45, 0.75, 0.75))
    assert big.grid_size == (1025, 8, 8)
    big.close()
    with pytest.raises(fiesta_amd.FiestaHipError):   # more than 32768 voxels on an axis: the shard protocol's coordinates
        fiesta_amd.ESDFMap((0, 0, 0), 0.001, (33.0, 0.004, 0.004))
    with pytest.raises(fiesta_amd.FiestaHipError):
        fiesta_amd.ESDFMap((0, 0, 0), -0.1, (1.0, 1.0, 1.0))
    with pytest.raises(fiesta_amd.FiestaHipError):
        fiesta_amd.ESDFMap((0, 0, 0), 0.1, (1.0, 1.0, 1.0), update_engine=99)
    with pytest.raises(fiesta_amd.FiestaHipError):
        fiesta_amd.ESDFMap((0, 0, 0), 0.1, (1.0, 1.0, 1.0), device=63)


def test_status_codes_not_exceptions(hip_lib):
    lib = hip_lib
    assert lib.fiesta_hip_update_esdf(None, None) != 0 and b"null" in lib.fiesta_hip_last_error()
    h = C.c_void_p()
    assert lib.fiesta_hip_create(None, C.byref(h)) != 0
    assert lib.fiesta_hip_destroy(None) == 0
    assert lib.fiesta_hip_version() >= 100 and lib.fiesta_hip_device_count() >= 1


def test_repeated_insert_same_voxel_and_redundant_observations(hip_lib, oracle_libs, best_oracle_kind):
    """The same voxel observed thousands of times in one batch (atomic counters), hits and misses mixed, across several
    cycles: majority vote and clamping must follow the oracle exactly."""
    import fiesta_amd
    gpu = fiesta_amd.ESDFMap((0, 0, 0), 0.2, (3.2, 3.2, 3.2))
    cpu = oracle_libs.OracleMap((0, 0, 0), 0.2, (3.2, 3.2, 3.2), kind=best_oracle_kind)
    for m in (gpu, cpu):
        m.SetParameters(*P_DEFAULT)
        m.SetOriginalRange()
    rng = np.random.RandomState(8)
    for cyc in range(12):
        v = np.repeat(rng.randint(0, 16, (6, 3)), 700, axis=0).astype(np.int32)
        o = (rng.rand(len(v)) < (0.7 if cyc % 3 else 0.3)).astype(np.int32)
        gpu.SetOccupancy(v, o)
        cpu.SetOccupancyVox(v, o)
        assert gpu.UpdateOccupancy(True) == cpu.UpdateOccupancy(True)
        assert (gpu.last_insert, gpu.last_delete) == (cpu.last_insert, cpu.last_delete)
        gpu.UpdateESDF()
        cpu.UpdateESDF()
        f, c = gpu.download_field(("occ", "logodds")), cpu.dump_dense(("occ", "logodds"))
        assert np.array_equal(f["occ"], c["occ"]) and np.array_equal(f["logodds"], c["logodds"])


def test_updated_voxel_unit_counts_first_observations(hip_lib):
    """snapshot_count_updated must see an unobserved -> observed transition (0xFFFFFFFF kept its tag bit after
    masking and compared equal to nothing): a map observed for the first time between snapshot and count reports every
    voxel, a second identical pass none, and an insert the whole grid (on both the tag-carrying and the clean words)."""
    import fiesta_amd
    n = 32
    m = fiesta_amd.ESDFMap((0, 0, 0), 0.1, (n * 0.1,) * 3)
    m.SetParameters(*P_DEFAULT)
    m.snapshot_save(0)
    m.SetOccupancyBox((0, 0, 0), (n - 1,) * 3, 0)
    m.UpdateOccupancy(True)
    m.UpdateESDF()
    assert m.snapshot_count_updated(0) == n ** 3        # -10000 -> +10000 on every voxel
    m.snapshot_save(0)
    m.SetOccupancyBox((0, 0, 0), (n - 1,) * 3, 0)
    m.UpdateOccupancy(True)
    m.UpdateESDF()
    assert m.snapshot_count_updated(0) == 0
    for _ in range(5):   # (observed free twice: the log-odds need more than three hits now)
        m.SetOccupancy(np.array([[5, 6, 7]], np.int32), 1)
        m.UpdateOccupancy(True)
    st = m.UpdateESDF()
    assert st["inserted"] == 1
    assert m.snapshot_count_updated(0) == n ** 3
    m.close()
