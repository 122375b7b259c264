"""Host-side mirror of the reference's operator API for the ESDF hot path.

``ESDFMap`` keeps the public method names, argument meaning and error conventions of
``class fiesta::ESDFMap`` (reference include/ESDFMap.h:111-166, src/ESDFMap.cpp) and forwards every
call through the C ABI of ``libfiesta_hip.so`` (include/fiesta_hip.h) to the HIP kernels.  Methods
accept either one voxel/position (scalar result, like the C++ class) or an (n,3) batch (array result);
the batch form is the fast path.  The reference is compiled C++; the C++ facade with the identical
class signature is include/fiesta/ESDFMap.h -- this module exists so that the parity tests and the
bench read like the reference's own driver (test/test_ESDF_Map.cpp:42-104).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import Config, FiestaHipError, RaycastParams, Stats, check

UNDEFINED = -10000   # undefined_  (src/ESDFMap.cpp:182)
INFINITY = 10000     # infinity_   (src/ESDFMap.cpp:181)
D2_INF = 0x7FFFFFFF
# what `update_engine=None` means (the library itself reads no environment: the test suites switch this attribute to run
# every scenario on both UpdateESDF engines)
DEFAULT_UPDATE_ENGINE = 0


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _d3(v):
    return np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(3))


ENGINES = {"auto": 0, "rounds": 1, "bulk": 2, "levels": 3, "envelope": 4, "cells": 5, "masked": 6}


class ESDFMap:
    """Drop-in for ``fiesta::ESDFMap``; array mode by default, hash-block mode with ``mode="hash"``."""

    def __init__(self, origin, resolution, map_size=None, reserve_size=0, mode="array", device=0,
                 update_engine=None, shard_lo=None, global_grid=None):
        self._lib = _lib.load()
        cfg = Config()
        cfg.mode = 0 if mode == "array" else 1
        cfg.device = int(device)
        cfg.origin[:] = list(_d3(origin))
        cfg.resolution = float(resolution)
        cfg.map_size[:] = list(_d3(map_size if map_size is not None else (0, 0, 0)))
        cfg.reserve_size = int(reserve_size)
        if update_engine is None or update_engine == 0 or update_engine == "auto":
            update_engine = DEFAULT_UPDATE_ENGINE
        cfg.update_engine = ENGINES.get(update_engine, update_engine)
        if shard_lo is not None:
            cfg.shard_lo[:] = [int(v) for v in shard_lo]
            cfg.global_grid[:] = [int(v) for v in global_grid]
        self.mode = mode
        self.resolution = float(resolution)
        self.origin = _d3(origin)
        self._h = C.c_void_p()
        check(self._lib.fiesta_hip_create(C.byref(cfg), C.byref(self._h)))
        gs = np.zeros(3, np.int32)
        check(self._lib.fiesta_hip_grid_size(self._h, _p(gs)))
        self.grid_size = tuple(int(v) for v in gs)
        self.last_insert = self.last_delete = 0
        self.served = {"levels": 0, "rounds": 0, "bulk": 0}   # UpdateESDF calls with work, by the engine that served them

    # -- life cycle ------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.fiesta_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def grid_total_size_(self) -> int:  # public data member of the array build (include/ESDFMap.h:115)
        n = C.c_int64(0)
        check(self._lib.fiesta_hip_grid_total_size(self._h, C.byref(n)))
        return n.value

    # -- parameters / window ------------------------------------------------------------------------
    def SetParameters(self, p_hit, p_miss, p_min, p_max, p_occ):
        check(self._lib.fiesta_hip_set_prob_params(self._h, p_hit, p_miss, p_min, p_max, p_occ))

    def SetUpdateRange(self, min_pos, max_pos, new_vec=True):
        check(self._lib.fiesta_hip_set_update_range(self._h, _p(_d3(min_pos)), _p(_d3(max_pos)), int(bool(new_vec))))

    def SetOriginalRange(self):
        check(self._lib.fiesta_hip_set_original_range(self._h))

    def set_update_engine(self, update_engine):
        """"auto" / "rounds" / "bulk" / "levels" / "envelope" / "cells" / "masked" (0 ... 6) from the next UpdateESDF on."""
        check(self._lib.fiesta_hip_set_update_engine(self._h, ENGINES.get(update_engine, update_engine)))

    # -- occupancy ingest ----------------------------------------------------------------------------
    def SetOccupancy(self, where, occ, want_ret=True):
        """SetOccupancy(Vector3i|Vector3d, int).  Integer input -> voxel overload, float -> position."""
        a = np.asarray(where)
        scalar = a.ndim == 1
        if np.issubdtype(a.dtype, np.integer):
            v = np.ascontiguousarray(a, dtype=np.int32).reshape(-1, 3)
            fn = self._lib.fiesta_hip_set_occupancy_vox
        else:
            v = np.ascontiguousarray(a, dtype=np.float64).reshape(-1, 3)
            fn = self._lib.fiesta_hip_set_occupancy_pos
        o = np.ascontiguousarray(np.broadcast_to(np.asarray(occ, dtype=np.int32), (len(v),)))
        ret = np.empty(len(v), np.int32) if want_ret else None
        check(fn(self._h, _p(v), _p(o), len(v), _p(ret)))
        if ret is None:
            return None
        return int(ret[0]) if scalar else ret

    def SetOccupancyDevice(self, vox_dev_ptr: int, occ_dev_ptr: int, n: int):
        """Batch already resident in HBM (n x 3 int32 voxels, n int32 flags)."""
        check(self._lib.fiesta_hip_set_occupancy_vox_dev(self._h, C.c_void_p(vox_dev_ptr), C.c_void_p(occ_dev_ptr), n))

    def SetOccupancyBox(self, lo, hi, occ):
        """SetOccupancy(Vector3i, occ) for every voxel of the inclusive voxel box [lo, hi], on the device."""
        a = np.ascontiguousarray(lo, dtype=np.int32).reshape(3)
        b = np.ascontiguousarray(hi, dtype=np.int32).reshape(3)
        check(self._lib.fiesta_hip_set_occupancy_box(self._h, _p(a), _p(b), int(occ)))

    def CheckUpdate(self) -> bool:
        out = C.c_int32(0)
        check(self._lib.fiesta_hip_check_update(self._h, C.byref(out)))
        return bool(out.value)

    def UpdateOccupancy(self, global_map=True) -> bool:
        ni, nd, any_ = C.c_int64(0), C.c_int64(0), C.c_int32(0)
        check(self._lib.fiesta_hip_update_occupancy(self._h, int(bool(global_map)), C.byref(ni), C.byref(nd),
                                                    C.byref(any_)))
        self.last_insert, self.last_delete = ni.value, nd.value
        return bool(any_.value)

    def UpdateESDF(self) -> dict:
        st = Stats()
        check(self._lib.fiesta_hip_update_esdf(self._h, C.byref(st)))
        d = st.as_dict()
        if d["inserted"] or d["deleted"] or d["rounds"] or d["bulk"]:   # (an update that had something to do)
            self.served["bulk" if d["bulk"] else "levels" if d["levels"] else "rounds"] += 1
        return d

    def level_trace(self):
        """The last level-engine update, level by level: [(frontier entries, microseconds inside the kernel), ...] for its
        first 48 levels, and the number of levels it ran (fiesta_hip_level_trace)."""
        out = (C.c_uint32 * 48)()
        n = C.c_int32(0)
        check(self._lib.fiesta_hip_level_trace(self._h, out, C.byref(n)))
        return [(int(w) >> 16, (int(w) & 0xFFFF) / 100.0) for w in list(out)[: min(n.value, 48)]], n.value

    def level_tuning(self, grid_groups=-1, spin_limit=-1):
        """Diagnostics of the level engine's wide levels (fiesta_hip_level_tuning): work-groups that take part (0: off), polls
        a barrier among them waits before the update is given up (0: at once)."""
        check(self._lib.fiesta_hip_level_tuning(self._h, int(grid_groups), int(spin_limit)))

    @property
    def only_levels(self) -> bool:
        """Every UpdateESDF of this map so far that had work ran the level engine from start to end (fiesta_hip_stats.levels):
        what the parity tests key their contract on (tests/scenarios.py: assert_envelope)."""
        return self.served["rounds"] == 0 and self.served["bulk"] == 0

    # -- ray casting -----------------------------------------------------------------------------------
    def RaycastFrame(self, points, transform, origin, min_ray_length, max_ray_length, l_cornor, r_cornor,
                     dedup=1, inverse=0):
        """One frame of Fiesta::RaycastProcess (include/Fiesta.h:194-278) on sensor-frame points.  inverse=1: this map
        is the SIGNED_NEEDED companion (inv_esdf_map_, :216-218, :249-251) -- end points free, crossed voxels occupied."""
        pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        T = np.ascontiguousarray(transform, dtype=np.float64).reshape(16)
        prm = RaycastParams(min_ray_length, max_ray_length, (C.c_double * 3)(*l_cornor),
                            (C.c_double * 3)(*r_cornor), int(dedup), int(inverse))
        check(self._lib.fiesta_hip_raycast_frame(self._h, _p(pts), len(pts), _p(T), _p(_d3(origin)), C.byref(prm)))

    def RaycastDepth(self, depth_mm, fx, fy, cx, cy, transform, origin, min_ray_length, max_ray_length,
                     l_cornor, r_cornor, dedup=1, inverse=0):
        """uint16 millimetre depth image -> points (include/Fiesta.h:341-351) -> ray cast, all on device."""
        d = np.ascontiguousarray(depth_mm, dtype=np.uint16)
        T = np.ascontiguousarray(transform, dtype=np.float64).reshape(16)
        prm = RaycastParams(min_ray_length, max_ray_length, (C.c_double * 3)(*l_cornor),
                            (C.c_double * 3)(*r_cornor), int(dedup), int(inverse))
        check(self._lib.fiesta_hip_raycast_depth(self._h, _p(d), d.shape[0], d.shape[1], fx, fy, cx, cy, _p(T),
                                                 _p(_d3(origin)), C.byref(prm)))

    @staticmethod
    def _depth_filter(rel_transform, tolerance, max_dist, min_dist, margin, reset):
        from ._lib import DepthFilter
        f = DepthFilter(tolerance, max_dist, min_dist, int(margin), int(bool(reset)))
        f.rel_transform[:] = list(np.asarray(rel_transform, np.float64).reshape(16))
        return f

    def RaycastDepthFiltered(self, depth_mm, fx, fy, cx, cy, transform, origin, min_ray_length, max_ray_length, l_cornor,
                             r_cornor, rel_transform, tolerance=0.1, max_dist=10.0, min_dist=0.1, margin=0, reset=False,
                             dedup=1):
        """RaycastDepth with DepthConversion's temporal consistency filter (include/Fiesta.h:352-379); rel_transform =
        inv(last_transform) @ transform. The previous image lives on the device; the first image of a run casts nothing."""
        d = np.ascontiguousarray(depth_mm, dtype=np.uint16)
        T = np.ascontiguousarray(transform, dtype=np.float64).reshape(16)
        prm = RaycastParams(min_ray_length, max_ray_length, (C.c_double * 3)(*l_cornor),
                            (C.c_double * 3)(*r_cornor), int(dedup), 0)
        f = self._depth_filter(rel_transform, tolerance, max_dist, min_dist, margin, reset)
        check(self._lib.fiesta_hip_raycast_depth_filtered(self._h, _p(d), d.shape[0], d.shape[1], fx, fy, cx, cy, _p(T),
                                                          _p(_d3(origin)), C.byref(prm), C.byref(f)))

    def DepthConversion(self, depth_mm, fx, fy, cx, cy, rel_transform=None, tolerance=0.1, max_dist=10.0, min_dist=0.1,
                        margin=0, reset=False):
        """Fiesta::DepthConversion: (rows*cols x 3 float32 points with NaN where the filter rejected, surviving count)."""
        d = np.ascontiguousarray(depth_mm, dtype=np.uint16)
        out = np.empty((d.shape[0] * d.shape[1], 3), np.float32)
        n = C.c_int64(0)
        f = None if rel_transform is None else self._depth_filter(rel_transform, tolerance, max_dist, min_dist, margin, reset)
        check(self._lib.fiesta_hip_depth_conversion(self._h, _p(d), d.shape[0], d.shape[1], fx, fy, cx, cy,
                                                    C.byref(f) if f is not None else None, _p(out), C.byref(n)))
        return out, n.value

    # -- queries ------------------------------------------------------------------------------------------
    def _query(self, where, fn_vox, fn_pos, out_dtype):
        a = np.asarray(where)
        scalar = a.ndim == 1
        if np.issubdtype(a.dtype, np.integer):
            v = np.ascontiguousarray(a, dtype=np.int32).reshape(-1, 3)
            fn = fn_vox
        else:
            v = np.ascontiguousarray(a, dtype=np.float64).reshape(-1, 3)
            fn = fn_pos
        out = np.empty(len(v), out_dtype)
        check(fn(self._h, _p(v), len(v), _p(out)))
        return out[0].item() if scalar else out

    def GetDistance(self, where):
        return self._query(where, self._lib.fiesta_hip_get_distance_vox, self._lib.fiesta_hip_get_distance_pos,
                           np.float64)

    def GetOccupancy(self, where):
        return self._query(where, self._lib.fiesta_hip_get_occupancy_vox, self._lib.fiesta_hip_get_occupancy_pos,
                           np.int32)

    def GetDistWithGradTrilinear(self, pos):
        a = np.asarray(pos, dtype=np.float64)
        scalar = a.ndim == 1
        v = np.ascontiguousarray(a).reshape(-1, 3)
        dist = np.empty(len(v), np.float64)
        grad = np.zeros((len(v), 3), np.float64)
        check(self._lib.fiesta_hip_get_dist_grad(self._h, _p(v), len(v), _p(dist), _p(grad)))
        return (float(dist[0]), grad[0]) if scalar else (dist, grad)

    def GetDistWithGradTrilinearDevice(self, pos_dev_ptr: int, n: int, dist_dev_ptr: int, grad_dev_ptr: int = 0):
        """device-resident batch (n x 3 f64 positions, n f64 distances, n x 3 f64 gradients or 0): the planner-side fast path"""
        check(self._lib.fiesta_hip_get_dist_grad_dev(self._h, C.c_void_p(pos_dev_ptr), n, C.c_void_p(dist_dev_ptr),
                                                     C.c_void_p(grad_dev_ptr) if grad_dev_ptr else None))

    @property
    def host_cache_fetches(self) -> int:
        """bricks of the field fetched for scalar host queries so far (fiesta_hip_host_cache_fetches)"""
        n = C.c_int64(0)
        check(self._lib.fiesta_hip_host_cache_fetches(self._h, C.byref(n)))
        return int(n.value)

    # -- whole field -----------------------------------------------------------------------------------------
    def download_field(self, want=("d2", "coc", "occ", "logodds")):
        n = self.grid_total_size_
        d2 = np.empty(n, np.int32) if "d2" in want else None
        coc = np.empty((n, 3), np.int32) if "coc" in want else None
        occ = np.empty(n, np.uint8) if "occ" in want else None
        lo = np.empty(n, np.float64) if "logodds" in want else None
        check(self._lib.fiesta_hip_download_field(self._h, _p(d2), _p(coc), _p(occ), _p(lo)))
        return {"d2": d2, "coc": coc, "occ": occ, "logodds": lo}

    def GetOccupiedVoxels(self) -> np.ndarray:
        """Voxel coordinates of all occupied voxels (the content of ESDFMap::GetPointCloud, src/ESDFMap.cpp:544-582)."""
        n = C.c_int64(0)
        check(self._lib.fiesta_hip_get_occupied_voxels(self._h, None, 0, C.byref(n)))
        out = np.empty((n.value, 3), np.int32)
        if n.value:
            check(self._lib.fiesta_hip_get_occupied_voxels(self._h, _p(out), n.value, C.byref(n)))
        return out

    def count_no_obstacle(self) -> int:
        """Observed voxels whose distance reads +10000 (on grids beyond 1024 per axis this includes everything farther than
        512 voxels from every obstacle: the reach of a stored id, include/fiesta_hip.h)."""
        n = C.c_int64(0)
        check(self._lib.fiesta_hip_count_no_obstacle(self._h, C.byref(n)))
        return n.value

    def GetSlice(self, z_vox: int) -> np.ndarray:
        """Distances of the plane z = z_vox as an (nx, ny) array (ESDFMap::GetSliceMarker, src/ESDFMap.cpp:639-699)."""
        out = np.empty(self.grid_size[:2], np.float64)
        check(self._lib.fiesta_hip_get_slice(self._h, int(z_vox), _p(out)))
        return out

    def save(self, path: str):
        """Raw checkpoint of the whole map state (include/fiesta_hip.h: fiesta_hip_save)."""
        check(self._lib.fiesta_hip_save(self._h, os.fsencode(path)))

    def load(self, path: str):
        check(self._lib.fiesta_hip_load(self._h, os.fsencode(path)))

    def GetPointCloud(self, vis_lower_bound: int, vis_upper_bound: int) -> np.ndarray:
        """ESDFMap::GetPointCloud (src/ESDFMap.cpp:544-582) as an (n, 3) float32 array of voxel centres (unordered)."""
        n = C.c_int64(0)
        check(self._lib.fiesta_hip_get_point_cloud(self._h, vis_lower_bound, vis_upper_bound, None, 0, C.byref(n)))
        out = np.empty((n.value, 3), np.float32)
        if n.value:
            check(self._lib.fiesta_hip_get_point_cloud(self._h, vis_lower_bound, vis_upper_bound, _p(out), n.value, C.byref(n)))
        return out

    def GetSliceMarker(self, slice_z: int, max_dist: float):
        """ESDFMap::GetSliceMarker (src/ESDFMap.cpp:639-699): (points (n,3) float64, colours (n,4) float32), unordered."""
        n = C.c_int64(0)
        check(self._lib.fiesta_hip_get_slice_marker(self._h, slice_z, max_dist, None, None, 0, C.byref(n)))
        xyz = np.empty((n.value, 3), np.float64)
        rgba = np.empty((n.value, 4), np.float32)
        if n.value:
            check(self._lib.fiesta_hip_get_slice_marker(self._h, slice_z, max_dist, _p(xyz), _p(rgba), n.value, C.byref(n)))
        return xyz, rgba

    def download_counts(self):
        """Pending (num_hit_, num_miss_) per voxel; num_miss_ counts all observations (src/ESDFMap.cpp:424)."""
        n = self.grid_total_size_
        hit = np.empty(n, np.int32)
        miss = np.empty(n, np.int32)
        check(self._lib.fiesta_hip_download_counts(self._h, _p(hit), _p(miss)))
        return hit, miss

    def download_hash(self):
        n = C.c_int64(0)
        check(self._lib.fiesta_hip_download_hash(self._h, C.byref(n), None, None, None, None))
        vox = np.empty((n.value, 3), np.int32)
        d2 = np.empty(n.value, np.int32)
        coc = np.empty((n.value, 3), np.int32)
        occ = np.empty(n.value, np.uint8)
        check(self._lib.fiesta_hip_download_hash(self._h, C.byref(n), _p(vox), _p(d2), _p(coc), _p(occ)))
        return {"vox": vox, "d2": d2, "coc": coc, "occ": occ}

    def hash_window(self):
        """Hash-block mode: (origin, moves) of the moving window -- map voxel of its lowest corner (it spans 1024 voxels per
        axis) and how often it has moved (include/fiesta_hip.h, "the moving window")."""
        org = np.zeros(3, np.int32)
        moves = C.c_int64(0)
        check(self._lib.fiesta_hip_hash_window(self._h, _p(org), C.byref(moves)))
        return org, moves.value

    def hash_recentre(self, centre_vox):
        c = np.ascontiguousarray(centre_vox, np.int32).reshape(3)
        check(self._lib.fiesta_hip_hash_recentre(self._h, _p(c)))

    def distance_from_d2(self, d2):
        """distance_buffer_ as the reference stores it: -10000 / +10000 sentinels, else sqrt(d2)*res."""
        d2 = np.asarray(d2)
        out = np.sqrt(np.maximum(d2, 0).astype(np.float64)) * self.resolution
        out[d2 < 0] = UNDEFINED
        out[d2 == D2_INF] = INFINITY
        return out

    def snapshot_save(self, slot=0):
        check(self._lib.fiesta_hip_snapshot_save(self._h, slot))

    def snapshot_restore(self, slot=0):
        check(self._lib.fiesta_hip_snapshot_restore(self._h, slot))

    def snapshot_count_updated(self, slot=0) -> int:
        n = C.c_int64(0)
        check(self._lib.fiesta_hip_snapshot_count_updated(self._h, slot, C.byref(n)))
        return n.value

    def synchronize(self):
        check(self._lib.fiesta_hip_synchronize(self._h))

    # -- shard interface (driven by fiesta_amd.sharded.ShardedESDFMap) ---------------------------
    def shard_info(self) -> dict:
        info = _lib.ShardInfo()
        check(self._lib.fiesta_hip_shard_info_get(self._h, C.byref(info)))
        return {k: tuple(getattr(info, k)) for k, _ in _lib.ShardInfo._fields_}

    def halo_pack(self, lo, hi) -> np.ndarray:
        lo = np.ascontiguousarray(lo, dtype=np.int32)
        hi = np.ascontiguousarray(hi, dtype=np.int32)
        out = np.empty(tuple(int(b - a + 1) for a, b in zip(lo, hi)), np.uint32)
        check(self._lib.fiesta_hip_halo_pack(self._h, _p(lo), _p(hi), _p(out)))
        return out

    def halo_apply(self, lo, hi, words) -> int:
        lo = np.ascontiguousarray(lo, dtype=np.int32)
        hi = np.ascontiguousarray(hi, dtype=np.int32)
        w = np.ascontiguousarray(words, dtype=np.uint32)
        assert w.size == int(np.prod(hi - lo + 1)), "halo buffer does not match the box"
        n = C.c_int64(0)
        check(self._lib.fiesta_hip_halo_apply(self._h, _p(lo), _p(hi), _p(w), C.byref(n)))
        return n.value

    def halo_pack_dev(self, lo, hi, out_ptr: int):
        lo = np.ascontiguousarray(lo, dtype=np.int32)
        hi = np.ascontiguousarray(hi, dtype=np.int32)
        check(self._lib.fiesta_hip_halo_pack_dev(self._h, _p(lo), _p(hi), C.c_void_p(out_ptr)))

    def halo_apply_dev(self, lo, hi, in_ptr: int) -> int:
        lo = np.ascontiguousarray(lo, dtype=np.int32)
        hi = np.ascontiguousarray(hi, dtype=np.int32)
        n = C.c_int64(0)
        check(self._lib.fiesta_hip_halo_apply_dev(self._h, _p(lo), _p(hi), C.c_void_p(in_ptr), C.byref(n)))
        return n.value

    def export_transitions(self) -> np.ndarray:
        n = C.c_int64(0)
        check(self._lib.fiesta_hip_export_transitions(self._h, None, 0, C.byref(n)))
        out = np.empty(2 * n.value, np.uint32)   # two words per entry: x | y << 16, z | occupied << 31
        if n.value:
            check(self._lib.fiesta_hip_export_transitions(self._h, _p(out), n.value, C.byref(n)))
        return out

    def apply_transitions(self, entries):
        e = np.ascontiguousarray(entries, dtype=np.uint32).reshape(-1)
        check(self._lib.fiesta_hip_apply_transitions(self._h, _p(e), len(e) // 2))

    def export_transitions_dev(self, out_ptr: int, capacity: int) -> int:
        n = C.c_int64(0)
        check(self._lib.fiesta_hip_export_transitions_dev(self._h, C.c_void_p(out_ptr) if out_ptr else None, capacity,
                                                          C.byref(n)))
        return n.value

    def apply_transitions_dev(self, ptr: int, n: int):
        check(self._lib.fiesta_hip_apply_transitions_dev(self._h, C.c_void_p(ptr), n))

    def esdf_seed(self) -> dict:
        st = Stats()
        check(self._lib.fiesta_hip_esdf_seed(self._h, C.byref(st)))
        return st.as_dict()

    def relax_pending(self):
        st = Stats()
        n = C.c_int64(0)
        check(self._lib.fiesta_hip_relax_pending(self._h, C.byref(st), C.byref(n)))
        return n.value, st.as_dict()


def signed_distance(esdf_map: "ESDFMap", inverse_map: "ESDFMap", pos_or_vox) -> np.ndarray:
    """Signed distance of a SIGNED_NEEDED pair (include/Fiesta.h:39-41; the reference leaves the combination as a TODO,
    :515-518): the map's distance to the nearest occupied voxel minus the inverse map's distance to the nearest voxel
    observed free -- positive in free space, negative inside obstacles.  NaN where either map holds no distance."""
    d, di = esdf_map.GetDistance(pos_or_vox), inverse_map.GetDistance(pos_or_vox)
    ok = (np.abs(d) < INFINITY) & (np.abs(di) < INFINITY)
    return np.where(ok, d - di, np.nan)
