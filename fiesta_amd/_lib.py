"""ctypes loader of the C-ABI shared library ``libfiesta_hip.so`` (see include/fiesta_hip.h).

There is deliberately no fallback: if the HIP library is missing or no gfx950 device is usable the
product path fails loudly (``FiestaHipError``).  Nothing under ``oracle/`` is ever imported here.
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfiesta_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "fiesta_hip.h")


class FiestaHipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"fiesta_hip error {code}: {message}")
        self.code = code


class Config(C.Structure):
    _fields_ = [("mode", C.c_int32), ("device", C.c_int32), ("origin", C.c_double * 3),
                ("resolution", C.c_double), ("map_size", C.c_double * 3), ("reserve_size", C.c_int32),
                ("update_engine", C.c_int32), ("shard_lo", C.c_int32 * 3), ("global_grid", C.c_int32 * 3)]


class Stats(C.Structure):
    _fields_ = [("inserted", C.c_int64), ("deleted", C.c_int64), ("invalidated", C.c_int64),
                ("rounds", C.c_int64), ("tile_visits", C.c_int64), ("sweeps", C.c_int64),
                ("voxel_writes", C.c_int64), ("device_ms", C.c_double), ("host_ms", C.c_double),
                ("relax_ms", C.c_double), ("relax_launches", C.c_int64), ("prof", C.c_int64 * 8),
                ("bulk", C.c_int64), ("ft_rows_ms", C.c_double), ("ft_plane_ms", C.c_double), ("ft_x_ms", C.c_double),
                ("ft_overflow", C.c_int64 * 6), ("observed_voxels", C.c_int64), ("occupied_voxels", C.c_int64),
                ("ft_max_d2", C.c_int64), ("dropped_observations", C.c_int64), ("levels", C.c_int64), ("grid_levels", C.c_int64),
                ("cells", C.c_int64), ("nn_cells_ms", C.c_double), ("nn_lists_ms", C.c_double), ("nn_fill_ms", C.c_double),
                ("nn_entries", C.c_int64), ("nn_failed", C.c_int64),
                ("nn_incremental", C.c_int64), ("nn_dirty_cells", C.c_int64), ("nn_brute_cells", C.c_int64), ("masked", C.c_int64), ("mask_uncertified", C.c_int64), ("mask_iterations", C.c_int64), ("mask_walks", C.c_int64), ("mask_quads", C.c_int64),
                ("mask_certify_ms", C.c_double), ("mask_repair_ms", C.c_double), ("path_notes", C.c_int64)]

    # fiesta_hip_stats.path_notes (include/fiesta_hip.h: FIESTA_HIP_NOTE_*): why an update took the path it took
    NOTES = ("partly_observed", "partial_window", "window_history", "late_observation", "first_wave_pending", "density",
             "cells_backoff", "cells_failed", "incremental_redone", "small_delta", "masked_gave_up", "levels_gave_up", "sharded",
             "id_wrap", "engine_pinned")

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["prof"] = list(self.prof)
        d["ft_overflow"] = list(self.ft_overflow)
        d["why"] = [name for bit, name in enumerate(self.NOTES) if (self.path_notes >> bit) & 1]
        return d


class ShardInfo(C.Structure):
    _fields_ = [("local_dims", C.c_int32 * 3), ("local_origin", C.c_int32 * 3), ("owned_lo", C.c_int32 * 3),
                ("owned_hi", C.c_int32 * 3), ("global_grid", C.c_int32 * 3)]


class DepthFilter(C.Structure):
    _fields_ = [("tolerance", C.c_double), ("max_dist", C.c_double), ("min_dist", C.c_double), ("margin", C.c_int32),
                ("reset", C.c_int32), ("rel_transform", C.c_double * 16)]


class RaycastParams(C.Structure):
    _fields_ = [("min_ray_length", C.c_double), ("max_ray_length", C.c_double), ("l_cornor", C.c_double * 3),
                ("r_cornor", C.c_double * 3), ("dedup", C.c_int32), ("inverse", C.c_int32)]


def declared_symbols(header_path: str = HEADER_PATH):
    """Names of every function include/fiesta_hip.h declares (used by the CPU export test)."""
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fiesta_hip_[a-z0-9_]+)\s*\(", text)))


_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7 (+ HSA runtime); libfiesta_hip.so needs the same
    SONAME from /opt/rocm.  Two HIP runtimes cannot share a process ("No HIP GPUs are available" in whichever
    initialises second), and which one wins would otherwise depend on import order.  If torch is installed, bind
    to its copy -- torch is only plumbing here (device buffers for the RCCL transport, bench.py), but it must be
    able to coexist.  Without torch the system ROCm runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return  # torch's runtime is already mapped; the dynamic loader will reuse it by SONAME
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load libfiesta_hip.so (raises FiestaHipError if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FiestaHipError(-1, f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                 "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(LIB_PATH)
    vp, i64, i32, dbl = C.c_void_p, C.c_int64, C.c_int32, C.c_double
    sig = {
        "fiesta_hip_last_error": (C.c_char_p, []),
        "fiesta_hip_version": (C.c_int, []),
        "fiesta_hip_device_count": (C.c_int, []),
        "fiesta_hip_create": (C.c_int, [vp, vp]),
        "fiesta_hip_destroy": (C.c_int, [vp]),
        "fiesta_hip_grid_size": (C.c_int, [vp, vp]),
        "fiesta_hip_grid_total_size": (C.c_int, [vp, vp]),
        "fiesta_hip_voxel_key": (C.c_int, [vp, vp, C.c_int64, vp]),
        "fiesta_hip_save": (C.c_int, [vp, C.c_char_p]),
        "fiesta_hip_load": (C.c_int, [vp, C.c_char_p]),
        "fiesta_hip_get_point_cloud": (C.c_int, [vp, i32, i32, vp, i64, vp]),
        "fiesta_hip_get_slice_marker": (C.c_int, [vp, i32, dbl, vp, vp, i64, vp]),
        "fiesta_hip_hash_window": (C.c_int, [vp, vp, vp]),
        "fiesta_hip_hash_recentre": (C.c_int, [vp, vp]),
        "fiesta_hip_rccl_unique_id": (C.c_int, [vp]),
        "fiesta_hip_shard_box": (C.c_int, [vp, C.c_int32, C.c_int32, vp, vp]),
        "fiesta_hip_shard_group_create": (C.c_int, [vp, vp, C.c_int32, C.c_int32, vp, vp]),
        "fiesta_hip_shard_group_create_hosted": (C.c_int, [vp, C.c_int32, C.c_int32, vp, vp]),
        "fiesta_hip_shard_group_precheck": (C.c_int, [vp, vp, C.c_int32, C.c_int32, C.c_int32]),
        "fiesta_hip_shard_group_comm_info": (C.c_int, [vp, vp, vp]),
        "fiesta_hip_shard_group_destroy": (C.c_int, [vp]),
        "fiesta_hip_shard_group_update_occupancy": (C.c_int, [vp, C.c_int32, vp, vp, vp]),
        "fiesta_hip_shard_group_update_esdf": (C.c_int, [vp, vp, vp, vp]),
        "fiesta_hip_set_prob_params": (C.c_int, [vp, dbl, dbl, dbl, dbl, dbl]),
        "fiesta_hip_set_update_range": (C.c_int, [vp, vp, vp, C.c_int]),
        "fiesta_hip_set_original_range": (C.c_int, [vp]),
        "fiesta_hip_set_update_engine": (C.c_int, [vp, C.c_int32]),
        "fiesta_hip_level_trace": (C.c_int, [vp, vp, vp]),
        "fiesta_hip_level_tuning": (C.c_int, [vp, C.c_int32, C.c_int64]),
        "fiesta_hip_count_no_obstacle": (C.c_int, [vp, vp]),
        "fiesta_hip_set_occupancy_vox": (C.c_int, [vp, vp, vp, i64, vp]),
        "fiesta_hip_set_occupancy_pos": (C.c_int, [vp, vp, vp, i64, vp]),
        "fiesta_hip_set_occupancy_vox_dev": (C.c_int, [vp, vp, vp, i64]),
        "fiesta_hip_set_occupancy_box": (C.c_int, [vp, vp, vp, i32]),
        "fiesta_hip_raycast_frame": (C.c_int, [vp, vp, i64, vp, vp, vp]),
        "fiesta_hip_raycast_frame_dev": (C.c_int, [vp, vp, i64, vp, vp, vp]),
        "fiesta_hip_raycast_depth": (C.c_int, [vp, vp, i32, i32, dbl, dbl, dbl, dbl, vp, vp, vp]),
        "fiesta_hip_raycast_depth_filtered": (C.c_int, [vp, vp, i32, i32, dbl, dbl, dbl, dbl, vp, vp, vp, vp]),
        "fiesta_hip_depth_conversion": (C.c_int, [vp, vp, i32, i32, dbl, dbl, dbl, dbl, vp, vp, vp]),
        "fiesta_hip_raycast_single": (C.c_int, [vp, vp, vp, vp, vp, i32, vp, i32]),
        "fiesta_hip_check_update": (C.c_int, [vp, vp]),
        "fiesta_hip_update_occupancy": (C.c_int, [vp, i32, vp, vp, vp]),
        "fiesta_hip_update_esdf": (C.c_int, [vp, vp]),
        "fiesta_hip_get_distance_vox": (C.c_int, [vp, vp, i64, vp]),
        "fiesta_hip_get_distance_pos": (C.c_int, [vp, vp, i64, vp]),
        "fiesta_hip_get_dist_grad": (C.c_int, [vp, vp, i64, vp, vp]),
        "fiesta_hip_get_dist_grad_dev": (C.c_int, [vp, vp, i64, vp, vp]),
        "fiesta_hip_host_cache_fetches": (C.c_int, [vp, vp]),
        "fiesta_hip_get_occupancy_vox": (C.c_int, [vp, vp, i64, vp]),
        "fiesta_hip_get_occupancy_pos": (C.c_int, [vp, vp, i64, vp]),
        "fiesta_hip_download_field": (C.c_int, [vp, vp, vp, vp, vp]),
        "fiesta_hip_download_counts": (C.c_int, [vp, vp, vp]),
        "fiesta_hip_get_occupied_voxels": (C.c_int, [vp, vp, i64, vp]),
        "fiesta_hip_get_slice": (C.c_int, [vp, i32, vp]),
        "fiesta_hip_download_hash": (C.c_int, [vp, vp, vp, vp, vp, vp]),
        "fiesta_hip_snapshot_save": (C.c_int, [vp, i32]),
        "fiesta_hip_snapshot_restore": (C.c_int, [vp, i32]),
        "fiesta_hip_snapshot_count_updated": (C.c_int, [vp, i32, vp]),
        "fiesta_hip_shard_info_get": (C.c_int, [vp, vp]),
        "fiesta_hip_halo_pack_dev": (C.c_int, [vp, vp, vp, vp]),
        "fiesta_hip_halo_apply_dev": (C.c_int, [vp, vp, vp, vp, vp]),
        "fiesta_hip_export_transitions_dev": (C.c_int, [vp, vp, i64, vp]),
        "fiesta_hip_apply_transitions_dev": (C.c_int, [vp, vp, i64]),
        "fiesta_hip_halo_pack": (C.c_int, [vp, vp, vp, vp]),
        "fiesta_hip_halo_apply": (C.c_int, [vp, vp, vp, vp, vp]),
        "fiesta_hip_export_transitions": (C.c_int, [vp, vp, i64, vp]),
        "fiesta_hip_apply_transitions": (C.c_int, [vp, vp, i64]),
        "fiesta_hip_esdf_seed": (C.c_int, [vp, vp]),
        "fiesta_hip_relax_pending": (C.c_int, [vp, vp, vp]),
        "fiesta_hip_synchronize": (C.c_int, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    lib._fiesta_signatures = sig
    _lib = lib
    return lib


def last_error() -> str:
    return load().fiesta_hip_last_error().decode(errors="replace")


def check(status: int):
    if status != 0:
        raise FiestaHipError(status, load().fiesta_hip_last_error().decode(errors="replace"))


def device_count() -> int:
    return int(load().fiesta_hip_device_count())
