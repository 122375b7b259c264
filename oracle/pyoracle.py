"""oracle/pyoracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/oracle_api.h.  Only tests/, __graft_entry__.smoke() and bench.py's
``cpu_baseline`` leg may import this module; the product package ``fiesta_amd`` never does.

Two interchangeable back ends export the same symbols:

* ``kind="ref"``  -> oracle/_ref/libfiesta_ref_{array,hash}.so: the reference's own
  src/ESDFMap.cpp + src/raycast.cpp compiled verbatim (oracle/Makefile, target ``ref``);
* ``kind="port"`` -> oracle/libfiesta_port.so: the CPU restatement oracle/esdf_port.cpp.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_ROOT = "/root/reference"

UNDEFINED = -10000
INFINITY = 10000


class EsdfStats(C.Structure):
    _fields_ = [("inserted", C.c_int64), ("deleted", C.c_int64), ("expanded", C.c_int64),
                ("change_num", C.c_int64), ("seconds", C.c_double)]


class RaycastParams(C.Structure):
    _fields_ = [("min_ray_length", C.c_double), ("max_ray_length", C.c_double),
                ("l_cornor", C.c_double * 3), ("r_cornor", C.c_double * 3)]


def lib_path(kind: str, mode: str = "array") -> str:
    if kind == "port":
        return os.path.join(_HERE, "libfiesta_port.so")
    if kind == "ref":
        return os.path.join(_HERE, "_ref", f"libfiesta_ref_{mode}.so")
    raise ValueError(kind)


def build(kind: str = "all") -> None:
    """Run oracle/Makefile.  ``ref`` needs /root/reference (this container only)."""
    targets = []
    if kind in ("all", "port"):
        targets.append("port")
    if kind in ("all", "ref") and os.path.exists(os.path.join(REFERENCE_ROOT, "src", "ESDFMap.cpp")):
        targets.append("ref")
    if targets:
        subprocess.run(["make", "-C", _HERE, "-s"] + targets, check=True)


def available(kind: str, mode: str = "array") -> bool:
    return os.path.exists(lib_path(kind, mode))


_LIBS: dict = {}


def _load(kind: str, mode: str):
    key = (kind, mode if kind == "ref" else "any")
    if key in _LIBS:
        return _LIBS[key]
    lib = C.CDLL(lib_path(kind, mode))
    vp, i64, dbl, i32 = C.c_void_p, C.c_int64, C.c_double, C.c_int
    sig = {
        "oracle_create": (vp, [i32, vp, dbl, vp, i32]),
        "oracle_destroy": (None, [vp]),
        "oracle_kind": (C.c_char_p, []),
        "oracle_set_parameters": (None, [vp, dbl, dbl, dbl, dbl, dbl]),
        "oracle_grid_total_size": (i64, [vp]),
        "oracle_grid_size": (None, [vp, vp]),
        "oracle_set_original_range": (None, [vp]),
        "oracle_set_update_range": (None, [vp, vp, vp, i32]),
        "oracle_set_occupancy_vox": (None, [vp, vp, vp, i64, vp]),
        "oracle_set_occupancy_pos": (None, [vp, vp, vp, i64, vp]),
        "oracle_check_update": (i32, [vp]),
        "oracle_update_occupancy": (i32, [vp, i32, vp, vp]),
        "oracle_update_esdf": (None, [vp, vp]),
        "oracle_get_distance_vox": (None, [vp, vp, i64, vp]),
        "oracle_get_distance_pos": (None, [vp, vp, i64, vp]),
        "oracle_get_dist_grad": (None, [vp, vp, i64, vp, vp]),
        "oracle_get_occupancy_vox": (None, [vp, vp, i64, vp]),
        "oracle_get_occupancy_pos": (None, [vp, vp, i64, vp]),
        "oracle_dump_dense": (None, [vp, vp, vp, vp, vp]),
        "oracle_dump_counts": (None, [vp, vp, vp]),
        "oracle_dump_hash": (i64, [vp, vp, vp, vp, vp]),
        "oracle_check_consistency": (i32, [vp]),
        "oracle_raycast_frame_signed": (None, [vp, vp, vp, i64, vp, vp, vp]),
        "oracle_get_point_cloud": (i64, [vp, i32, i32, vp, i64]),
        "oracle_get_slice_marker": (i64, [vp, i32, C.c_double, vp, vp, i64]),
        "oracle_raycast": (i32, [vp, vp, vp, vp, vp, i32]),
        "oracle_raycast_frame": (None, [vp, vp, i64, vp, vp, vp]),
        "oracle_depth_conversion": (i64, [vp, vp, i32, i32, dbl, dbl, dbl, dbl, i32, vp, dbl, dbl, dbl, i32, vp]),
    }
    if kind == "port":   # the model of the GPU's level-synchronous schedule lives in the restatement only
        sig["oracle_set_schedule"] = (None, [vp, i32])
        sig["oracle_levels_run"] = (i64, [vp])
        sig["oracle_fifo_probe"] = (None, [vp, i32])
        sig["oracle_fifo_layers"] = (i64, [vp, vp, i64])
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _LIBS[key] = lib
    return lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _d3(v):
    return np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(3))


class OracleMap:
    """The reference ``fiesta::ESDFMap`` (or its restatement) driven through the oracle C API."""

    def __init__(self, origin, resolution, map_size=None, reserve_size=0, mode="array", kind="port"):
        self.kind, self.mode = kind, mode
        self.lib = _load(kind, mode)
        self.resolution = float(resolution)
        self.origin = _d3(origin)
        ms = _d3(map_size if map_size is not None else (0, 0, 0))
        self.h = self.lib.oracle_create(0 if mode == "array" else 1, _p(self.origin), float(resolution),
                                        _p(ms), int(reserve_size))
        if not self.h:
            raise RuntimeError(f"oracle_create failed for kind={kind} mode={mode}")
        gs = np.zeros(3, np.int32)
        self.lib.oracle_grid_size(self.h, _p(gs))
        self.grid_size = tuple(int(x) for x in gs)

    def close(self):
        if self.h:
            self.lib.oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def describe(self):
        return self.lib.oracle_kind().decode()

    @property
    def grid_total_size(self):
        return int(self.lib.oracle_grid_total_size(self.h))

    def SetParameters(self, p_hit, p_miss, p_min, p_max, p_occ):
        self.lib.oracle_set_parameters(self.h, p_hit, p_miss, p_min, p_max, p_occ)

    def SetOriginalRange(self):
        self.lib.oracle_set_original_range(self.h)

    def SetUpdateRange(self, min_pos, max_pos, new_vec=True):
        self.lib.oracle_set_update_range(self.h, _p(_d3(min_pos)), _p(_d3(max_pos)), int(bool(new_vec)))

    def SetOccupancyVox(self, vox, occ):
        vox = np.ascontiguousarray(vox, dtype=np.int32).reshape(-1, 3)
        occ = np.ascontiguousarray(np.broadcast_to(np.asarray(occ, dtype=np.int32), (len(vox),)))
        ret = np.empty(len(vox), np.int32)
        self.lib.oracle_set_occupancy_vox(self.h, _p(vox), _p(occ), len(vox), _p(ret))
        return ret

    def SetOccupancyPos(self, pos, occ):
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        occ = np.ascontiguousarray(np.broadcast_to(np.asarray(occ, dtype=np.int32), (len(pos),)))
        ret = np.empty(len(pos), np.int32)
        self.lib.oracle_set_occupancy_pos(self.h, _p(pos), _p(occ), len(pos), _p(ret))
        return ret

    def CheckUpdate(self):
        return bool(self.lib.oracle_check_update(self.h))

    def UpdateOccupancy(self, global_map=True):
        ni, nd = C.c_int64(0), C.c_int64(0)
        r = self.lib.oracle_update_occupancy(self.h, int(bool(global_map)), C.byref(ni), C.byref(nd))
        self.last_insert, self.last_delete = ni.value, nd.value
        return bool(r)

    def UpdateESDF(self):
        st = EsdfStats()
        self.lib.oracle_update_esdf(self.h, C.byref(st))
        return {k: getattr(st, k) for k, _ in EsdfStats._fields_}

    def GetDistanceVox(self, vox):
        vox = np.ascontiguousarray(vox, dtype=np.int32).reshape(-1, 3)
        out = np.empty(len(vox), np.float64)
        self.lib.oracle_get_distance_vox(self.h, _p(vox), len(vox), _p(out))
        return out

    def GetDistancePos(self, pos):
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        out = np.empty(len(pos), np.float64)
        self.lib.oracle_get_distance_pos(self.h, _p(pos), len(pos), _p(out))
        return out

    def GetDistWithGradTrilinear(self, pos):
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        dist = np.empty(len(pos), np.float64)
        grad = np.zeros((len(pos), 3), np.float64)
        self.lib.oracle_get_dist_grad(self.h, _p(pos), len(pos), _p(dist), _p(grad))
        return dist, grad

    def GetOccupancyVox(self, vox):
        vox = np.ascontiguousarray(vox, dtype=np.int32).reshape(-1, 3)
        out = np.empty(len(vox), np.int32)
        self.lib.oracle_get_occupancy_vox(self.h, _p(vox), len(vox), _p(out))
        return out

    def GetOccupancyPos(self, pos):
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        out = np.empty(len(pos), np.int32)
        self.lib.oracle_get_occupancy_pos(self.h, _p(pos), len(pos), _p(out))
        return out

    def dump_dense(self, want=("dist", "coc", "occ", "logodds")):
        n = self.grid_total_size
        dist = np.empty(n, np.float64) if "dist" in want else None
        coc = np.empty((n, 3), np.int32) if "coc" in want else None
        occ = np.empty(n, np.uint8) if "occ" in want else None
        lo = np.empty(n, np.float64) if "logodds" in want else None
        self.lib.oracle_dump_dense(self.h, _p(dist), _p(coc), _p(occ), _p(lo))
        return {"dist": dist, "coc": coc, "occ": occ, "logodds": lo}

    def dump_counts(self):
        """Pending (num_hit_, num_miss_); hash mode: one entry per allocated slot, in the order of dump_hash()."""
        n = self.grid_total_size if self.mode != "hash" else int(self.lib.oracle_dump_hash(self.h, None, None, None, None))
        hit = np.empty(n, np.int32)
        miss = np.empty(n, np.int32)
        self.lib.oracle_dump_counts(self.h, _p(hit), _p(miss))
        return hit, miss

    def dump_hash(self):
        n = int(self.lib.oracle_dump_hash(self.h, None, None, None, None))
        vox = np.empty((n, 3), np.int32)
        dist = np.empty(n, np.float64)
        coc = np.empty((n, 3), np.int32)
        occ = np.empty(n, np.uint8)
        self.lib.oracle_dump_hash(self.h, _p(vox), _p(dist), _p(coc), _p(occ))
        return {"vox": vox, "dist": dist, "coc": coc, "occ": occ}

    def set_schedule(self, schedule):
        """port only: 0 = the reference's FIFO, 1 = the CPU model of the GPU's level engine (esdf_port.cpp: relax_levels);
        2 .. 6 = experiments on that model (tools/dev/schedule_experiment.py)."""
        self.lib.oracle_set_schedule(self.h, int(schedule))

    def fifo_probe(self, enable=True):
        """port only: watch the update queue's layers (esdf_port.cpp: relax, fifo_probe) from now on"""
        self.lib.oracle_fifo_probe(self.h, int(bool(enable)))

    def fifo_layers(self):
        """rows (entries, entries depending on an earlier entry of the same layer, longest chain) per layer since fifo_probe()"""
        n = int(self.lib.oracle_fifo_layers(self.h, None, 0))
        out = np.zeros((max(n, 1), 3), np.int64)
        self.lib.oracle_fifo_layers(self.h, _p(out), n)
        return out[:n]

    @property
    def levels_run(self):
        return int(self.lib.oracle_levels_run(self.h))

    def CheckConsistency(self):
        return bool(self.lib.oracle_check_consistency(self.h))

    def GetPointCloud(self, vis_lower_bound, vis_upper_bound):
        n = int(self.lib.oracle_get_point_cloud(self.h, vis_lower_bound, vis_upper_bound, None, 0))
        out = np.empty((n, 3), np.float32)
        self.lib.oracle_get_point_cloud(self.h, vis_lower_bound, vis_upper_bound, _p(out), n)
        return out

    def GetSliceMarker(self, slice_z, max_dist):
        n = int(self.lib.oracle_get_slice_marker(self.h, slice_z, max_dist, None, None, 0))
        xyz = np.empty((n, 3), np.float64)
        rgba = np.empty((n, 4), np.float32)
        self.lib.oracle_get_slice_marker(self.h, slice_z, max_dist, _p(xyz), _p(rgba), n)
        return xyz, rgba

    def raycast_frame(self, points, transform, origin, min_ray, max_ray, l_cornor, r_cornor, inverse_map=None):
        """One frame; inverse_map: the SIGNED_NEEDED companion map fed with inverted observations (include/Fiesta.h:216,249)."""
        pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        T = np.ascontiguousarray(transform, dtype=np.float64).reshape(16)
        prm = RaycastParams(min_ray, max_ray, (C.c_double * 3)(*l_cornor), (C.c_double * 3)(*r_cornor))
        if inverse_map is None:
            self.lib.oracle_raycast_frame(self.h, _p(pts), len(pts), _p(T), _p(_d3(origin)), C.byref(prm))
        else:
            self.lib.oracle_raycast_frame_signed(self.h, inverse_map.h, _p(pts), len(pts), _p(T), _p(_d3(origin)), C.byref(prm))


def raycast(start, end, minv, maxv, kind="port", cap=4096):
    lib = _load(kind, "array")
    out = np.empty((cap, 3), np.float64)
    n = lib.oracle_raycast(_p(_d3(start)), _p(_d3(end)), _p(_d3(minv)), _p(_d3(maxv)), _p(out), cap)
    if n < 0:
        raise IndexError("Too many RaycasMultithread voxels")
    return out[:n].copy()


def depth_conversion(cur, last, fx, fy, cx, cy, rel=None, tolerance=0.1, max_dist=10.0, min_dist=0.1, margin=0, kind="port"):
    """Fiesta::DepthConversion restated (oracle/depth_filter.inc): the frame's cloud (n x 3 float32) in pixel order.
    rel None: no filter; else rel = inv(last_transform) @ transform and `last` the previous image (None: first image)."""
    lib = _load(kind, "array")
    cur = np.ascontiguousarray(cur, dtype=np.uint16)
    rows, cols = cur.shape
    out = np.empty((rows * cols, 3), np.float32)
    use = rel is not None
    relm = np.ascontiguousarray(rel if use else np.eye(4), dtype=np.float64).reshape(16)
    lastp = None if last is None else np.ascontiguousarray(last, dtype=np.uint16)
    n = lib.oracle_depth_conversion(_p(cur), _p(lastp), rows, cols, fx, fy, cx, cy, int(use), _p(relm), tolerance, max_dist,
                                    min_dist, margin, _p(out))
    return out[:n].copy()
