// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (see oracle/oracle_api.h).
//
// C-API wrapper around the reference's own fiesta::ESDFMap and Raycast, which oracle/Makefile compiles
// VERBATIM from /root/reference/src/{ESDFMap,raycast}.cpp against oracle/shim. Nothing in this file
// re-implements the map algorithm; it only drives the reference class, reaches its private buffers for
// dumps (#define private public around the include), mutes the std::cout prints the reference makes
// inside the hot path (src/ESDFMap.cpp:188,237,277,394) and parses the two counters it prints.
// The only restated logic is Fiesta::RaycastProcess (include/Fiesta.h:194-278), which lives in a header
// that cannot be compiled without ROS/PCL/OpenCV.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <queue>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include <Eigen/Eigen>
#include <pcl/kdtree/kdtree_flann.h>
#include <sensor_msgs/PointCloud.h>
#include <visualization_msgs/Marker.h>

#define private public
#include "ESDFMap.h"
#undef private
#include "raycast.h"

#include "oracle_api.h"

namespace {
struct CoutMute {
  std::streambuf *old;
  std::stringbuf buf;
  CoutMute() : old(std::cout.rdbuf(&buf)) {}
  ~CoutMute() { std::cout.rdbuf(old); }
  std::string str() const { return buf.str(); }
};
inline Eigen::Vector3d V3d(const double *p) { return Eigen::Vector3d(p[0], p[1], p[2]); }
inline Eigen::Vector3i V3i(const int32_t *p) { return Eigen::Vector3i(p[0], p[1], p[2]); }
}  // namespace

struct oracle_map {
  fiesta::ESDFMap *map = nullptr;
  double resolution = 0;
  // Fiesta<>::set_free_/set_occ_/tot_ (include/Fiesta.h:107-110,287)
#ifdef HASH_TABLE
  std::unordered_set<int> set_free_, set_occ_;
#else
  std::vector<int> set_free_, set_occ_;
#endif
  int tot_ = 0;
};

extern "C" {

const char *oracle_kind(void) {
#ifdef HASH_TABLE
  return "reference-hash";
#else
  return "reference-array";
#endif
}

oracle_map *oracle_create(int mode, const double origin[3], double resolution, const double map_size[3],
                          int reserve_size) {
  CoutMute mute;
#ifdef HASH_TABLE
  if (mode != 1) return nullptr;
  (void)map_size;
  oracle_map *m = new oracle_map;
  m->map = new fiesta::ESDFMap(V3d(origin), resolution, reserve_size);
#else
  if (mode != 0) return nullptr;
  (void)reserve_size;
  oracle_map *m = new oracle_map;
  m->map = new fiesta::ESDFMap(V3d(origin), resolution, V3d(map_size));
  m->set_free_.assign(m->map->grid_total_size_, 0);
  m->set_occ_.assign(m->map->grid_total_size_, 0);
#endif
  m->resolution = resolution;
  return m;
}

void oracle_destroy(oracle_map *m) {
  if (!m) return;
  delete m->map;
  delete m;
}

void oracle_set_parameters(oracle_map *m, double p_hit, double p_miss, double p_min, double p_max,
                           double p_occ) {
  m->map->SetParameters(p_hit, p_miss, p_min, p_max, p_occ);
}

int64_t oracle_grid_total_size(oracle_map *m) {
#ifdef HASH_TABLE
  return m->map->count;
#else
  return m->map->grid_total_size_;
#endif
}

void oracle_grid_size(oracle_map *m, int32_t out[3]) {
#ifdef HASH_TABLE
  (void)m;
  out[0] = out[1] = out[2] = 0;
#else
  for (int i = 0; i < 3; ++i) out[i] = m->map->grid_size_(i);
#endif
}

void oracle_set_original_range(oracle_map *m) { m->map->SetOriginalRange(); }

void oracle_set_update_range(oracle_map *m, const double min_pos[3], const double max_pos[3], int new_vec) {
  m->map->SetUpdateRange(V3d(min_pos), V3d(max_pos), new_vec != 0);
}

void oracle_set_occupancy_vox(oracle_map *m, const int32_t *vox, const int32_t *occ, int64_t n, int32_t *ret) {
  CoutMute mute;
  for (int64_t i = 0; i < n; ++i) {
    int r = m->map->SetOccupancy(V3i(vox + 3 * i), occ[i]);
    if (ret) ret[i] = r;
  }
}

void oracle_set_occupancy_pos(oracle_map *m, const double *pos, const int32_t *occ, int64_t n, int32_t *ret) {
  CoutMute mute;
  for (int64_t i = 0; i < n; ++i) {
    int r = m->map->SetOccupancy(V3d(pos + 3 * i), occ[i]);
    if (ret) ret[i] = r;
  }
}

int oracle_check_update(oracle_map *m) { return m->map->CheckUpdate() ? 1 : 0; }

int oracle_update_occupancy(oracle_map *m, int global_map, int64_t *n_insert, int64_t *n_delete) {
  CoutMute mute;
  bool r = m->map->UpdateOccupancy(global_map != 0);
  if (n_insert) *n_insert = (int64_t)m->map->insert_queue_.size();
  if (n_delete) *n_delete = (int64_t)m->map->delete_queue_.size();
  return r ? 1 : 0;
}

void oracle_update_esdf(oracle_map *m, oracle_esdf_stats *stats) {
  CoutMute mute;
  int64_t ins = (int64_t)m->map->insert_queue_.size(), del = (int64_t)m->map->delete_queue_.size();
  auto t0 = std::chrono::steady_clock::now();
  m->map->UpdateESDF();
  auto t1 = std::chrono::steady_clock::now();
  if (stats) {
    stats->inserted = ins;
    stats->deleted = del;
    stats->expanded = -1;
    stats->change_num = -1;
    stats->seconds = std::chrono::duration<double>(t1 - t0).count();
    // "Expanding T nodes, with change_num = C, accumulator = A" (src/ESDFMap.cpp:394)
    std::string s = mute.str();
    size_t p = s.find("Expanding ");
    if (p != std::string::npos) {
      long long t = 0, c = 0;
      if (std::sscanf(s.c_str() + p, "Expanding %lld nodes, with change_num = %lld", &t, &c) == 2) {
        stats->expanded = t;
        stats->change_num = c;
      }
    }
  }
}

void oracle_get_distance_vox(oracle_map *m, const int32_t *vox, int64_t n, double *out) {
  for (int64_t i = 0; i < n; ++i) out[i] = m->map->GetDistance(V3i(vox + 3 * i));
}
void oracle_get_distance_pos(oracle_map *m, const double *pos, int64_t n, double *out) {
  for (int64_t i = 0; i < n; ++i) out[i] = m->map->GetDistance(V3d(pos + 3 * i));
}
void oracle_get_dist_grad(oracle_map *m, const double *pos, int64_t n, double *dist, double *grad) {
  for (int64_t i = 0; i < n; ++i) {
    Eigen::Vector3d g(0, 0, 0);
    dist[i] = m->map->GetDistWithGradTrilinear(V3d(pos + 3 * i), g);
    grad[3 * i] = g(0);
    grad[3 * i + 1] = g(1);
    grad[3 * i + 2] = g(2);
  }
}
void oracle_get_occupancy_vox(oracle_map *m, const int32_t *vox, int64_t n, int32_t *out) {
  for (int64_t i = 0; i < n; ++i) out[i] = m->map->GetOccupancy(V3i(vox + 3 * i));
}
void oracle_get_occupancy_pos(oracle_map *m, const double *pos, int64_t n, int32_t *out) {
  for (int64_t i = 0; i < n; ++i) out[i] = m->map->GetOccupancy(V3d(pos + 3 * i));
}

void oracle_dump_dense(oracle_map *m, double *dist, int32_t *coc, uint8_t *occ, double *logodds) {
#ifdef HASH_TABLE
  (void)m; (void)dist; (void)coc; (void)occ; (void)logodds;
#else
  fiesta::ESDFMap &e = *m->map;
  const int64_t n = e.grid_total_size_;
  for (int64_t i = 0; i < n; ++i) {
    if (dist) dist[i] = e.distance_buffer_[i];
    if (coc) {
      coc[3 * i] = e.closest_obstacle_[i](0);
      coc[3 * i + 1] = e.closest_obstacle_[i](1);
      coc[3 * i + 2] = e.closest_obstacle_[i](2);
    }
    if (occ) occ[i] = e.Exist((int)i) ? 1 : 0;
    if (logodds) logodds[i] = e.occupancy_buffer_[i];
  }
#endif
}

void oracle_dump_counts(oracle_map *m, int32_t *num_hit, int32_t *num_miss) {
#ifndef HASH_TABLE
  for (int64_t i = 0; i < m->map->grid_total_size_; ++i) {
    if (num_hit) num_hit[i] = m->map->num_hit_[i];
    if (num_miss) num_miss[i] = m->map->num_miss_[i];
  }
#else  // hash build: the order of oracle_dump_hash (slot k+1 of the reference's vectors)
  fiesta::ESDFMap &e = *m->map;
  for (int64_t k = 0; k < e.count - 1; ++k) {
    if (num_hit) num_hit[k] = e.num_hit_[k + 1];
    if (num_miss) num_miss[k] = e.num_miss_[k + 1];
  }
#endif
}

int64_t oracle_dump_hash(oracle_map *m, int32_t *vox, double *dist, int32_t *coc, uint8_t *occ) {
#ifdef HASH_TABLE
  fiesta::ESDFMap &e = *m->map;
  const int64_t n = e.count - 1;
  for (int64_t k = 0; k < n; ++k) {
    const int i = (int)k + 1;
    if (vox) {
      vox[3 * k] = e.vox_buffer_[i](0);
      vox[3 * k + 1] = e.vox_buffer_[i](1);
      vox[3 * k + 2] = e.vox_buffer_[i](2);
    }
    if (dist) dist[k] = e.distance_buffer_[i];
    if (coc) {
      coc[3 * k] = e.closest_obstacle_[i](0);
      coc[3 * k + 1] = e.closest_obstacle_[i](1);
      coc[3 * k + 2] = e.closest_obstacle_[i](2);
    }
    if (occ) occ[k] = e.Exist(i) ? 1 : 0;
  }
  return n;
#else
  (void)m; (void)vox; (void)dist; (void)coc; (void)occ;
  return 0;
#endif
}

int oracle_check_consistency(oracle_map *m) {
  CoutMute mute;
  return m->map->CheckConsistency() ? 1 : 0;
}

int64_t oracle_get_point_cloud(oracle_map *m, int vis_lower_bound, int vis_upper_bound, float *xyz, int64_t cap) {
  sensor_msgs::PointCloud pc;
  m->map->GetPointCloud(pc, vis_lower_bound, vis_upper_bound);
  for (size_t i = 0; i < pc.points.size() && (int64_t)i < cap; ++i)
    xyz[3 * i] = pc.points[i].x, xyz[3 * i + 1] = pc.points[i].y, xyz[3 * i + 2] = pc.points[i].z;
  return (int64_t)pc.points.size();
}
int64_t oracle_get_slice_marker(oracle_map *m, int slice, double max_dist, double *xyz, float *rgba, int64_t cap) {
  visualization_msgs::Marker mk;
  m->map->GetSliceMarker(mk, slice, 0, Eigen::Vector4d(0, 0, 0, 1), max_dist);
  for (size_t i = 0; i < mk.points.size() && (int64_t)i < cap; ++i) {
    xyz[3 * i] = mk.points[i].x, xyz[3 * i + 1] = mk.points[i].y, xyz[3 * i + 2] = mk.points[i].z;
    rgba[4 * i] = mk.colors[i].r, rgba[4 * i + 1] = mk.colors[i].g, rgba[4 * i + 2] = mk.colors[i].b, rgba[4 * i + 3] = mk.colors[i].a;
  }
  return (int64_t)mk.points.size();
}

int oracle_raycast(const double start[3], const double end[3], const double minv[3], const double maxv[3],
                   double *out, int cap) {
  std::vector<Eigen::Vector3d> o;
  std::streambuf *olderr = std::cerr.rdbuf(nullptr);
  int n;
  try {
    Raycast(V3d(start), V3d(end), V3d(minv), V3d(maxv), &o);
    n = (int)o.size();
  } catch (const std::out_of_range &) {
    n = -1;
  }
  std::cerr.rdbuf(olderr);
  std::cerr.clear();
  for (int i = 0; i < (int)o.size() && i < cap; ++i) {
    out[3 * i] = o[i](0);
    out[3 * i + 1] = o[i](1);
    out[3 * i + 2] = o[i](2);
  }
  return n;
}

// Restatement of Fiesta::RaycastProcess(0, cloud.size(), tt) + the tt bump of RaycastMultithread with
// ray_cast_num_thread_ == 0 (include/Fiesta.h:194-303). Same loop directions, same break/continue.
static void frame_impl(oracle_map *m, oracle_map *inv, const float *points, int64_t n, const double T[16],
                       const double origin_[3], const oracle_raycast_params *p);
void oracle_raycast_frame(oracle_map *m, const float *points, int64_t n, const double T[16],
                          const double origin_[3], const oracle_raycast_params *p) {
  frame_impl(m, nullptr, points, n, T, origin_, p);
}
void oracle_raycast_frame_signed(oracle_map *m, oracle_map *inv, const float *points, int64_t n, const double T[16],
                                 const double origin_[3], const oracle_raycast_params *p) {
  frame_impl(m, inv, points, n, T, origin_, p);
}
// inv != nullptr: the SIGNED_NEEDED lines (:216-218, :249-251)
static void frame_impl(oracle_map *m, oracle_map *inv, const float *points, int64_t n, const double T[16],
                       const double origin_[3], const oracle_raycast_params *p) {
  CoutMute mute;
#ifdef HASH_TABLE
  m->set_free_.clear();
  m->set_occ_.clear();
#endif
  const int tt = ++m->tot_;
  const double res = m->resolution;
  const Eigen::Vector3d origin = V3d(origin_), half(0.5, 0.5, 0.5);
  const Eigen::Vector3d lc = V3d(p->l_cornor) / res, rc = V3d(p->r_cornor) / res;
  std::vector<Eigen::Vector3d> output;
  for (int64_t idx = 0; idx < n; ++idx) {
    const double px = points[3 * idx], py = points[3 * idx + 1], pz = points[3 * idx + 2];
    int cnt = 0;
    if (std::isnan(px) || std::isnan(py) || std::isnan(pz)) continue;  // :202
    double h[4];
    for (int r = 0; r < 4; ++r) h[r] = T[4 * r] * px + T[4 * r + 1] * py + T[4 * r + 2] * pz + T[4 * r + 3] * 1.0;
    Eigen::Vector3d point = Eigen::Vector3d(h[0], h[1], h[2]) / h[3];  // :204-205
    int tmp_idx;
    double length = (point - origin).norm();
    if (length < p->min_ray_length)
      continue;  // :209
    else if (length > p->max_ray_length) {
      point = (point - origin) / length * p->max_ray_length + origin;  // :212
      tmp_idx = m->map->SetOccupancy((Eigen::Vector3d)point, 0);
    } else
      tmp_idx = m->map->SetOccupancy((Eigen::Vector3d)point, 1);  // :215
    if (inv) tmp_idx = inv->map->SetOccupancy((Eigen::Vector3d)point, 0);  // :216-218
    if (tmp_idx != -10000) {  // :221-232
#ifdef HASH_TABLE
      if (m->set_occ_.find(tmp_idx) != m->set_occ_.end()) continue;
      m->set_occ_.insert(tmp_idx);
#else
      if (tmp_idx >= 0 && tmp_idx < (int)m->set_occ_.size()) {  // guard: reference indexes unchecked
        if (m->set_occ_[tmp_idx] == tt) continue;
        m->set_occ_[tmp_idx] = tt;
      }
#endif
    }
    Raycast(origin / res, point / res, lc, rc, &output);  // :233-237
    for (int i = (int)output.size() - 2; i >= 0; i--) {   // :239-276
      Eigen::Vector3d tmp = (output[i] + half) * res;
      length = (tmp - origin).norm();
      if (length < p->min_ray_length) break;
      if (length > p->max_ray_length) continue;
      int fidx = m->map->SetOccupancy(tmp, 0);
      if (inv) fidx = inv->map->SetOccupancy(tmp, 1);  // :249-251
      if (fidx != -10000) {
#ifdef HASH_TABLE
        if (m->set_free_.find(fidx) != m->set_free_.end()) {
          if (++cnt >= 1) {
            cnt = 0;
            break;
          }
        } else {
          m->set_free_.insert(fidx);
          cnt = 0;
        }
#else
        if (fidx >= 0 && fidx < (int)m->set_free_.size()) {
          if (m->set_free_[fidx] == tt) {
            if (++cnt >= 1) {
              cnt = 0;
              break;
            }
          } else {
            m->set_free_[fidx] = tt;
            cnt = 0;
          }
        }
#endif
      }
    }
  }
}

#include "depth_filter.inc"

}  // extern "C"
