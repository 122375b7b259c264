// fiesta_amd/csrc/c_api.hip -- the extern "C" boundary declared in include/fiesta_hip.h.
// Every entry point converts C++ exceptions into a status code + thread-local message; nothing throws
// across the ABI and no HIP / C++ type appears in a signature.
#include <cstring>
#include <string>

#include "../../include/fiesta_hip.h"
#include "dense_map.hpp"
#include "hash_map.hpp"
#include "shard_group.hpp"

using fiesta::DenseMap;
using fiesta::Error;
using fiesta::HashMap;

struct fiesta_hip_shard_group {
  fiesta::ShardGroup *g = nullptr;
};

struct fiesta_hip_map {
  int mode = 0;
  DenseMap *dense = nullptr;
  HashMap *hash = nullptr;
};

namespace {
thread_local std::string g_last_error;

template <typename F>
int guarded(F &&f) {
  try {
    f();
    return FIESTA_HIP_OK;
  } catch (const Error &e) {
    g_last_error = e.what();
    return e.code;
  } catch (const std::exception &e) {
    g_last_error = e.what();
    return FIESTA_HIP_ERR_DEVICE;
  } catch (...) {
    g_last_error = "unknown error";
    return FIESTA_HIP_ERR_DEVICE;
  }
}
void need(bool ok, const char *msg) {
  if (!ok) throw Error(FIESTA_HIP_ERR_INVALID, msg);
}
DenseMap &dense(fiesta_hip_map *m, const char *what) {
  need(m != nullptr, "null map handle");
  if (!m->dense) throw Error(FIESTA_HIP_ERR_INVALID, std::string(what) + ": only available on array-mode maps");
  return *m->dense;
}
}  // namespace

extern "C" {

const char *fiesta_hip_last_error(void) { return g_last_error.c_str(); }
int fiesta_hip_version(void) { return 100; }

int fiesta_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  int ok = 0;
  for (int i = 0; i < n; ++i) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, i) == hipSuccess && std::strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
  }
  return ok;
}

int fiesta_hip_create(const fiesta_hip_config *cfg, fiesta_hip_map **out) {
  return guarded([&] {
    need(cfg && out, "null argument");
    *out = nullptr;
    auto *m = new fiesta_hip_map;
    try {
      m->mode = cfg->mode;
      if (cfg->mode == FIESTA_HIP_MODE_ARRAY)
        m->dense = new DenseMap(*cfg);
      else if (cfg->mode == FIESTA_HIP_MODE_HASH)
        m->hash = new HashMap(*cfg);
      else
        throw Error(FIESTA_HIP_ERR_INVALID, "unknown mode");
    } catch (...) {
      delete m;
      throw;
    }
    *out = m;
  });
}

int fiesta_hip_destroy(fiesta_hip_map *m) {
  return guarded([&] {
    if (!m) return;
    delete m->dense;
    delete m->hash;
    delete m;
  });
}

int fiesta_hip_grid_size(fiesta_hip_map *m, int32_t out[3]) {
  return guarded([&] {
    need(m && out, "null argument");
    if (m->dense) {
      out[0] = m->dense->geom().nx;
      out[1] = m->dense->geom().ny;
      out[2] = m->dense->geom().nz;
    } else {
      out[0] = out[1] = out[2] = 0;
    }
  });
}
int fiesta_hip_grid_total_size(fiesta_hip_map *m, int64_t *out) {
  return guarded([&] {
    need(m && out, "null argument");
    *out = m->dense ? m->dense->total() : m->hash->allocated_voxels();
  });
}

int fiesta_hip_voxel_key(fiesta_hip_map *m, const int32_t *vox, int64_t n, int32_t *out) {
  return guarded([&] {
    need(m && (n == 0 || (vox && out)), "null argument");
    for (int64_t i = 0; i < n; ++i) {
      const int32_t *v = vox + 3 * i;
      if (m->dense) {
        const fiesta::Geom &g = m->dense->geom();
        out[i] = (v[0] - g.gx0) * g.ny * g.nz + (v[1] - g.gy0) * g.nz + (v[2] - g.gz0);
      } else {
        out[i] = HashMap::voxel_key(v[0], v[1], v[2]);
      }
    }
  });
}

int fiesta_hip_hash_window(fiesta_hip_map *m, int32_t origin[3], int64_t *moves) {
  return guarded([&] {
    need(m && origin, "null argument");
    need(m->hash != nullptr, "fiesta_hip_hash_window: not a hash-block map");
    m->hash->window_origin(origin);
    if (moves) *moves = m->hash->window_moves();
  });
}
int fiesta_hip_hash_recentre(fiesta_hip_map *m, const int32_t centre[3]) {
  return guarded([&] {
    need(m && centre, "null argument");
    need(m->hash != nullptr, "fiesta_hip_hash_recentre: not a hash-block map");
    m->hash->recentre(centre);
  });
}

int fiesta_hip_set_prob_params(fiesta_hip_map *m, double p_hit, double p_miss, double p_min, double p_max,
                               double p_occ) {
  return guarded([&] {
    need(m != nullptr, "null map handle");
    if (m->dense)
      m->dense->set_prob_params(p_hit, p_miss, p_min, p_max, p_occ);
    else
      m->hash->set_prob_params(p_hit, p_miss, p_min, p_max, p_occ);
  });
}
int fiesta_hip_set_update_range(fiesta_hip_map *m, const double mn[3], const double mx[3], int new_vec) {
  return guarded([&] {
    need(m && mn && mx, "null argument");
    if (m->dense)
      m->dense->set_update_range(mn, mx, new_vec != 0);
    else
      m->hash->set_update_range(mn, mx, new_vec != 0);
  });
}
int fiesta_hip_set_original_range(fiesta_hip_map *m) {
  return guarded([&] {
    need(m != nullptr, "null map handle");
    if (m->dense)
      m->dense->set_original_range();
    else
      m->hash->set_original_range();
  });
}

int fiesta_hip_set_update_engine(fiesta_hip_map *m, int32_t engine) {
  return guarded([&] {
    need(m != nullptr, "null map handle");
    need(engine >= 0 && engine <= 6, "unknown update_engine");
    if (m->dense)
      m->dense->set_update_engine(engine);
    else
      m->hash->set_update_engine(engine > 3 ? 0 : engine);  // (no transform on a hash-block map: 2, 4, 5 and 6 mean 0 there)
  });
}

int fiesta_hip_level_trace(fiesta_hip_map *m, uint32_t out[48], int32_t *n_levels) {
  return guarded([&] {
    need(m && out && n_levels, "bad argument");
    *n_levels = m->dense ? m->dense->level_trace(out) : m->hash->level_trace(out);
  });
}

int fiesta_hip_level_tuning(fiesta_hip_map *m, int32_t grid_groups, int64_t spin_limit) {
  return guarded([&] {
    need(m != nullptr, "null map handle");
    need(grid_groups <= 32, "at most 32 work-groups (the CUs of one XCD)");
    need(spin_limit <= 0xFFFFFFFFll, "spin_limit does not fit 32 bits");
    if (m->dense)
      m->dense->level_tuning(grid_groups, spin_limit);
    else
      m->hash->level_tuning(grid_groups, spin_limit);
  });
}

int fiesta_hip_set_occupancy_vox(fiesta_hip_map *m, const int32_t *vox, const int32_t *occ, int64_t n, int32_t *ret) {
  return guarded([&] {
    need(m && (n == 0 || (vox && occ)) && n >= 0, "bad argument");
    if (m->dense)
      m->dense->observe_vox(vox, occ, n, ret, false);
    else
      m->hash->observe_vox(vox, occ, n, ret);
  });
}
int fiesta_hip_set_occupancy_pos(fiesta_hip_map *m, const double *pos, const int32_t *occ, int64_t n, int32_t *ret) {
  return guarded([&] {
    need(m && (n == 0 || (pos && occ)) && n >= 0, "bad argument");
    if (m->dense)
      m->dense->observe_pos(pos, occ, n, ret);
    else
      m->hash->observe_pos(pos, occ, n, ret);
  });
}
int fiesta_hip_set_occupancy_vox_dev(fiesta_hip_map *m, const int32_t *vox_dev, const int32_t *occ_dev, int64_t n) {
  return guarded([&] {
    need(m && (n == 0 || (vox_dev && occ_dev)) && n >= 0, "bad argument");
    if (m->dense)
      m->dense->observe_vox(vox_dev, occ_dev, n, nullptr, true);
    else
      m->hash->observe_vox(vox_dev, occ_dev, n, nullptr, true);
  });
}

int fiesta_hip_set_occupancy_box(fiesta_hip_map *m, const int32_t lo[3], const int32_t hi[3], int32_t occ) {
  return guarded([&] {
    need(m && lo && hi && (occ == 0 || occ == 1), "bad argument");
    if (m->dense)
      m->dense->observe_box(lo, hi, occ);
    else
      m->hash->observe_box(lo, hi, occ);
  });
}

int fiesta_hip_raycast_frame(fiesta_hip_map *m, const float *points, int64_t n, const double T[16],
                             const double origin[3], const fiesta_hip_raycast_params *p) {
  return guarded([&] {
    need(m && (n == 0 || points) && T && origin && p && n >= 0, "bad argument");
    if (m->dense)
      m->dense->raycast_frame(points, n, T, origin, p, false);
    else
      m->hash->raycast_frame(points, n, T, origin, p, false);
  });
}
int fiesta_hip_raycast_frame_dev(fiesta_hip_map *m, const float *points_dev, int64_t n, const double T[16],
                                 const double origin[3], const fiesta_hip_raycast_params *p) {
  return guarded([&] {
    need(m && (n == 0 || points_dev) && T && origin && p && n >= 0, "bad argument");
    if (m->dense)
      m->dense->raycast_frame(points_dev, n, T, origin, p, true);
    else
      m->hash->raycast_frame(points_dev, n, T, origin, p, true);
  });
}
int fiesta_hip_raycast_depth(fiesta_hip_map *m, const uint16_t *depth, int32_t rows, int32_t cols, double fx,
                             double fy, double cx, double cy, const double T[16], const double origin[3],
                             const fiesta_hip_raycast_params *p) {
  return guarded([&] {
    need(m && depth && rows > 0 && cols > 0 && T && origin && p, "bad argument");
    if (m->dense)
      m->dense->raycast_depth(depth, rows, cols, fx, fy, cx, cy, T, origin, p);
    else
      m->hash->raycast_depth(depth, rows, cols, fx, fy, cx, cy, T, origin, p);
  });
}
int fiesta_hip_raycast_depth_filtered(fiesta_hip_map *m, const uint16_t *depth, int32_t rows, int32_t cols, double fx, double fy,
                                      double cx, double cy, const double T[16], const double origin[3],
                                      const fiesta_hip_raycast_params *p, const fiesta_hip_depth_filter *f) {
  return guarded([&] {
    need(m && depth && rows > 0 && cols > 0 && T && origin && p && f, "bad argument");
    if (m->dense)
      m->dense->raycast_depth(depth, rows, cols, fx, fy, cx, cy, T, origin, p, f);
    else
      m->hash->raycast_depth(depth, rows, cols, fx, fy, cx, cy, T, origin, p, f);
  });
}
int fiesta_hip_depth_conversion(fiesta_hip_map *m, const uint16_t *depth, int32_t rows, int32_t cols, double fx, double fy,
                                double cx, double cy, const fiesta_hip_depth_filter *f, float *points_out, int64_t *n_valid) {
  return guarded([&] {
    need(depth && rows > 0 && cols > 0 && points_out, "bad argument");
    const int64_t n = dense(m, "depth_conversion").depth_conversion(depth, rows, cols, fx, fy, cx, cy, f, points_out);
    if (n_valid) *n_valid = n;
  });
}
int fiesta_hip_raycast_single(const double start[3], const double end[3], const double minv[3],
                              const double maxv[3], double *out, int32_t cap, int32_t *n_out, int32_t device) {
  return guarded([&] {
    need(start && end && minv && maxv && n_out && (cap == 0 || out), "bad argument");
    fiesta::raycast_single(start, end, minv, maxv, out, cap, n_out, device);
  });
}

int fiesta_hip_check_update(fiesta_hip_map *m, int32_t *out) {
  return guarded([&] {
    need(m && out, "null argument");
    *out = (m->dense ? m->dense->check_update() : m->hash->check_update()) ? 1 : 0;
  });
}
int fiesta_hip_update_occupancy(fiesta_hip_map *m, int32_t global_map, int64_t *n_insert, int64_t *n_delete,
                                int32_t *any) {
  return guarded([&] {
    need(m != nullptr, "null map handle");
    const bool r = m->dense ? m->dense->update_occupancy(global_map != 0, n_insert, n_delete)
                            : m->hash->update_occupancy(global_map != 0, n_insert, n_delete);
    if (any) *any = r ? 1 : 0;
  });
}
int fiesta_hip_update_esdf(fiesta_hip_map *m, fiesta_hip_stats *stats) {
  return guarded([&] {
    need(m != nullptr, "null map handle");
    if (m->dense) {
      m->dense->update_esdf(stats);
      if (stats) stats->path_notes = (int64_t)m->dense->path_notes();
    } else {
      m->hash->update_esdf(stats);
    }
  });
}

int fiesta_hip_get_distance_vox(fiesta_hip_map *m, const int32_t *vox, int64_t n, double *out) {
  return guarded([&] {
    need(m && (n == 0 || (vox && out)) && n >= 0, "bad argument");
    if (m->dense)
      m->dense->get_distance_vox(vox, n, out);
    else
      m->hash->get_distance_vox(vox, n, out);
  });
}
int fiesta_hip_get_distance_pos(fiesta_hip_map *m, const double *pos, int64_t n, double *out) {
  return guarded([&] {
    need(m && (n == 0 || (pos && out)) && n >= 0, "bad argument");
    if (m->dense)
      m->dense->get_distance_pos(pos, n, out);
    else
      m->hash->get_distance_pos(pos, n, out);
  });
}
int fiesta_hip_get_dist_grad(fiesta_hip_map *m, const double *pos, int64_t n, double *dist, double *grad) {
  return guarded([&] {
    need(m && (n == 0 || (pos && dist)) && n >= 0, "bad argument");
    if (m->dense)
      m->dense->get_dist_grad(pos, n, dist, grad, false);
    else
      m->hash->get_dist_grad(pos, n, dist, grad);
  });
}
int fiesta_hip_get_dist_grad_dev(fiesta_hip_map *m, const double *pos_dev, int64_t n, double *dist_dev,
                                 double *grad_dev) {
  return guarded([&] {
    need(m && (n == 0 || (pos_dev && dist_dev)) && n >= 0, "bad argument");
    dense(m, "get_dist_grad_dev").get_dist_grad(pos_dev, n, dist_dev, grad_dev, true);
  });
}
int fiesta_hip_host_cache_fetches(fiesta_hip_map *m, int64_t *fetches) {
  return guarded([&] {
    need(m && fetches, "bad argument");
    *fetches = m->dense ? m->dense->host_brick_fetches() : m->hash->host_brick_fetches();
  });
}
int fiesta_hip_get_occupancy_vox(fiesta_hip_map *m, const int32_t *vox, int64_t n, int32_t *out) {
  return guarded([&] {
    need(m && (n == 0 || (vox && out)) && n >= 0, "bad argument");
    if (m->dense)
      m->dense->get_occupancy_vox(vox, n, out);
    else
      m->hash->get_occupancy_vox(vox, n, out);
  });
}
int fiesta_hip_get_occupancy_pos(fiesta_hip_map *m, const double *pos, int64_t n, int32_t *out) {
  return guarded([&] {
    need(m && (n == 0 || (pos && out)) && n >= 0, "bad argument");
    if (m->dense)
      m->dense->get_occupancy_pos(pos, n, out);
    else
      m->hash->get_occupancy_pos(pos, n, out);
  });
}

int fiesta_hip_download_field(fiesta_hip_map *m, int32_t *d2, int32_t *coc, uint8_t *occ, double *logodds) {
  return guarded([&] { dense(m, "download_field").download_field(d2, coc, occ, logodds); });
}
int fiesta_hip_download_counts(fiesta_hip_map *m, int32_t *num_hit, int32_t *num_miss) {
  return guarded([&] {
    need(m != nullptr, "null map");
    if (m->dense)
      m->dense->download_counts(num_hit, num_miss);
    else
      m->hash->download_counts(num_hit, num_miss);
  });
}
int fiesta_hip_count_no_obstacle(fiesta_hip_map *m, int64_t *n_out) {
  return guarded([&] {
    need(m && n_out, "null argument");
    *n_out = dense(m, "count_no_obstacle").count_no_obstacle();
  });
}
int fiesta_hip_get_occupied_voxels(fiesta_hip_map *m, int32_t *vox, int64_t capacity, int64_t *n_out) {
  return guarded([&] {
    need(n_out != nullptr && capacity >= 0, "bad argument");
    *n_out = dense(m, "get_occupied_voxels").occupied_voxels(vox, capacity);
  });
}
int fiesta_hip_get_slice(fiesta_hip_map *m, int32_t z_vox, double *out) {
  return guarded([&] {
    need(out != nullptr, "null argument");
    dense(m, "get_slice").slice_distances(z_vox, out);
  });
}
int fiesta_hip_save(fiesta_hip_map *m, const char *path) {
  return guarded([&] {
    need(m && path, "null argument");
    if (m->dense)
      m->dense->checkpoint(path, true);
    else
      m->hash->checkpoint(path, true);
  });
}
int fiesta_hip_load(fiesta_hip_map *m, const char *path) {
  return guarded([&] {
    need(m && path, "null argument");
    if (m->dense)
      m->dense->checkpoint(path, false);
    else
      m->hash->checkpoint(path, false);
  });
}
int fiesta_hip_get_point_cloud(fiesta_hip_map *m, int32_t vis_lower_bound, int32_t vis_upper_bound, float *xyz,
                               int64_t capacity, int64_t *n_out) {
  return guarded([&] {
    need(m && n_out && capacity >= 0, "bad argument");
    *n_out = m->dense ? m->dense->point_cloud(vis_lower_bound, vis_upper_bound, xyz, capacity)
                      : m->hash->point_cloud(vis_lower_bound, vis_upper_bound, xyz, capacity);
  });
}
int fiesta_hip_get_slice_marker(fiesta_hip_map *m, int32_t slice, double max_dist, double *xyz, float *rgba,
                                int64_t capacity, int64_t *n_out) {
  return guarded([&] {
    need(m && n_out && capacity >= 0, "bad argument");
    *n_out = m->dense ? m->dense->slice_marker(slice, max_dist, xyz, rgba, capacity)
                      : m->hash->slice_marker(slice, max_dist, xyz, rgba, capacity);
  });
}
int fiesta_hip_download_hash(fiesta_hip_map *m, int64_t *n_out, int32_t *vox, int32_t *d2, int32_t *coc,
                             uint8_t *occ) {
  return guarded([&] {
    need(m && n_out, "null argument");
    if (!m->hash) throw Error(FIESTA_HIP_ERR_INVALID, "download_hash: only available on hash-mode maps");
    *n_out = m->hash->download(vox, d2, coc, occ);
  });
}

int fiesta_hip_snapshot_save(fiesta_hip_map *m, int32_t slot) {
  return guarded([&] {
    need(m != nullptr, "null map");
    if (m->dense) {
      m->dense->snapshot_save(slot);
    } else {  // hash-block maps keep ONE copy of the state words, enough for the "updated voxels" unit (no restore)
      need(slot == 0, "hash-mode maps have snapshot slot 0 only");
      m->hash->snapshot_save();
    }
  });
}
int fiesta_hip_snapshot_restore(fiesta_hip_map *m, int32_t slot) {
  return guarded([&] { dense(m, "snapshot_restore").snapshot_restore(slot); });
}
int fiesta_hip_snapshot_count_updated(fiesta_hip_map *m, int32_t slot, int64_t *updated) {
  return guarded([&] {
    need(updated != nullptr, "null argument");
    need(m != nullptr, "null map");
    if (m->dense) {
      *updated = m->dense->snapshot_count_updated(slot);
    } else {
      need(slot == 0, "hash-mode maps have snapshot slot 0 only");
      *updated = m->hash->snapshot_count_updated();
    }
  });
}

int fiesta_hip_shard_info_get(fiesta_hip_map *m, fiesta_hip_shard_info *out) {
  return guarded([&] {
    need(out != nullptr, "null argument");
    const fiesta::Geom &g = dense(m, "shard_info").geom();
    out->local_dims[0] = g.nx, out->local_dims[1] = g.ny, out->local_dims[2] = g.nz;
    out->local_origin[0] = g.gx0, out->local_origin[1] = g.gy0, out->local_origin[2] = g.gz0;
    out->owned_lo[0] = g.ox0, out->owned_lo[1] = g.oy0, out->owned_lo[2] = g.oz0;
    out->owned_hi[0] = g.ox1, out->owned_hi[1] = g.oy1, out->owned_hi[2] = g.oz1;
    out->global_grid[0] = g.GX, out->global_grid[1] = g.GY, out->global_grid[2] = g.GZ;
  });
}
int fiesta_hip_halo_pack_dev(fiesta_hip_map *m, const int32_t lo[3], const int32_t hi[3], uint32_t *out_dev) {
  return guarded([&] {
    need(lo && hi && out_dev, "null argument");
    dense(m, "halo_pack").halo_pack(lo, hi, out_dev);
  });
}
int fiesta_hip_halo_apply_dev(fiesta_hip_map *m, const int32_t lo[3], const int32_t hi[3], const uint32_t *in_dev,
                              int64_t *n_changed) {
  return guarded([&] {
    need(lo && hi && in_dev, "null argument");
    const int64_t k = dense(m, "halo_apply").halo_apply(lo, hi, in_dev);
    if (n_changed) *n_changed = k;
  });
}
int fiesta_hip_export_transitions_dev(fiesta_hip_map *m, uint32_t *out_dev, int64_t capacity, int64_t *n_out) {
  return guarded([&] {
    need(n_out != nullptr, "null argument");
    *n_out = dense(m, "export_transitions").export_transitions(out_dev, capacity);
  });
}
int fiesta_hip_apply_transitions_dev(fiesta_hip_map *m, const uint32_t *entries_dev, int64_t n) {
  return guarded([&] {
    need(n == 0 || entries_dev, "null argument");
    dense(m, "apply_transitions").apply_transitions(entries_dev, n);
  });
}
int fiesta_hip_halo_pack(fiesta_hip_map *m, const int32_t lo[3], const int32_t hi[3], uint32_t *out) {
  return guarded([&] {
    need(lo && hi && out, "null argument");
    DenseMap &d = dense(m, "halo_pack");
    const int64_t n = (int64_t)(hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1) * (hi[2] - lo[2] + 1);
    need(n > 0, "empty box");
    uint32_t *buf = d.scratch_u32(n);
    d.halo_pack(lo, hi, buf);
    d.copy_to_host(out, buf, n * sizeof(uint32_t));
  });
}
int fiesta_hip_halo_apply(fiesta_hip_map *m, const int32_t lo[3], const int32_t hi[3], const uint32_t *in,
                          int64_t *n_changed) {
  return guarded([&] {
    need(lo && hi && in, "null argument");
    DenseMap &d = dense(m, "halo_apply");
    const int64_t n = (int64_t)(hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1) * (hi[2] - lo[2] + 1);
    need(n > 0, "empty box");
    uint32_t *buf = d.scratch_u32(n);
    d.copy_to_device(buf, in, n * sizeof(uint32_t));
    const int64_t k = d.halo_apply(lo, hi, buf);
    if (n_changed) *n_changed = k;
  });
}
int fiesta_hip_export_transitions(fiesta_hip_map *m, uint32_t *out, int64_t capacity, int64_t *n_out) {
  return guarded([&] {
    need(n_out != nullptr, "null argument");
    DenseMap &d = dense(m, "export_transitions");
    const int64_t n = d.export_transitions(nullptr, 0);
    *n_out = n;
    if (!out || n == 0) return;
    need(n <= capacity, "transition buffer too small");
    uint32_t *buf = d.scratch_u32(2 * n);
    d.export_transitions(buf, n);
    d.copy_to_host(out, buf, 2 * n * sizeof(uint32_t));
  });
}
int fiesta_hip_apply_transitions(fiesta_hip_map *m, const uint32_t *entries, int64_t n) {
  return guarded([&] {
    need(n == 0 || entries, "null argument");
    if (n == 0) return;
    DenseMap &d = dense(m, "apply_transitions");
    uint32_t *buf = d.scratch_u32(2 * n);
    d.copy_to_device(buf, entries, 2 * n * sizeof(uint32_t));
    d.apply_transitions(buf, n);
  });
}
int fiesta_hip_esdf_seed(fiesta_hip_map *m, fiesta_hip_stats *stats) {
  return guarded([&] { dense(m, "esdf_seed").update_esdf(stats, true); });
}
int fiesta_hip_relax_pending(fiesta_hip_map *m, fiesta_hip_stats *stats, int64_t *pending) {
  return guarded([&] { dense(m, "relax_pending").relax_pending(stats, pending); });
}

int fiesta_hip_rccl_unique_id(uint8_t id[128]) {
  return guarded([&] {
    need(id != nullptr, "null argument");
    fiesta::rccl_unique_id(id);
  });
}
int fiesta_hip_shard_box(const int32_t gg[3], int32_t world, int32_t rank, int32_t lo[3], int32_t size[3]) {
  return guarded([&] {
    need(gg && lo && size, "null argument");
    need(rank >= 0 && rank < world, "rank out of range");
    const int g3[3] = {gg[0], gg[1], gg[2]};
    int l3[3], s3[3];
    fiesta::shard_box(g3, world, rank, l3, s3);
    for (int i = 0; i < 3; ++i) lo[i] = l3[i], size[i] = s3[i];
  });
}
int fiesta_hip_shard_group_create(fiesta_hip_map *const *shards, const int32_t *ranks, int32_t n_local, int32_t world,
                                  const uint8_t *rccl_id, fiesta_hip_shard_group **out) {
  return guarded([&] {
    need(shards && ranks && out && n_local > 0, "null argument");
    *out = nullptr;
    std::vector<DenseMap *> maps;
    std::vector<int> rk;
    for (int i = 0; i < n_local; ++i) {
      maps.push_back(&dense(shards[i], "shard_group_create"));
      rk.push_back(ranks[i]);
    }
    auto *h = new fiesta_hip_shard_group;
    try {
      h->g = new fiesta::ShardGroup(maps, rk, world, rccl_id);
    } catch (...) {
      delete h;
      throw;
    }
    *out = h;
  });
}
int fiesta_hip_shard_group_create_hosted(fiesta_hip_map *shard, int32_t rank, int32_t world, const fiesta_hip_shard_transport *t,
                                         fiesta_hip_shard_group **out) {
  return guarded([&] {
    need(shard && out && t && t->all_gather && t->exchange, "null argument");
    *out = nullptr;
    std::vector<DenseMap *> maps{&dense(shard, "shard_group_create_hosted")};
    std::vector<int> rk{rank};
    auto *h = new fiesta_hip_shard_group;
    try {
      h->g = new fiesta::ShardGroup(maps, rk, world, nullptr, t);
    } catch (...) {
      delete h;
      throw;
    }
    *out = h;
  });
}
int fiesta_hip_shard_group_precheck(fiesta_hip_map *const *shards, const int32_t *ranks, int32_t n_local, int32_t world, int32_t use_rccl) {
  return guarded([&] {
    need(shards && ranks && n_local > 0, "null argument");
    std::vector<DenseMap *> maps;
    std::vector<int> rk;
    for (int i = 0; i < n_local; ++i) {
      maps.push_back(&dense(shards[i], "shard_group_precheck"));
      rk.push_back(ranks[i]);
    }
    fiesta::ShardGroup::precheck(maps, rk, world, use_rccl != 0);
  });
}
int fiesta_hip_shard_group_comm_info(fiesta_hip_shard_group *g, int32_t *nranks, int32_t *rank) {
  return guarded([&] {
    need(g && g->g, "null shard group");
    int n = 0, r = 0;
    g->g->comm_info(&n, &r);
    if (nranks) *nranks = n;
    if (rank) *rank = r;
  });
}
int fiesta_hip_shard_group_destroy(fiesta_hip_shard_group *g) {
  return guarded([&] {
    if (!g) return;
    delete g->g;
    delete g;
  });
}
int fiesta_hip_shard_group_update_occupancy(fiesta_hip_shard_group *g, int32_t global_map, int64_t *n_insert, int64_t *n_delete,
                                            int32_t *any) {
  return guarded([&] {
    need(g && g->g, "null shard group");
    const bool a = g->g->update_occupancy(global_map != 0, n_insert, n_delete);
    if (any) *any = a ? 1 : 0;
  });
}
int fiesta_hip_shard_group_update_esdf(fiesta_hip_shard_group *g, fiesta_hip_stats *stats, int32_t *sweeps, int64_t *entries_sent) {
  return guarded([&] {
    need(g && g->g, "null shard group");
    g->g->update_esdf(stats, sweeps, entries_sent);
  });
}

int fiesta_hip_synchronize(fiesta_hip_map *m) {
  return guarded([&] {
    need(m != nullptr, "null map handle");
    if (m->dense)
      m->dense->synchronize();
    else
      m->hash->synchronize();
  });
}

}  // extern "C"
