// include/fiesta/ESDFMap.h -- drop-in `fiesta::ESDFMap` backed by the MI355X engine (libfiesta_hip.so).
//
// Same public surface as the reference class (HKUST-Aerial-Robotics/FIESTA include/ESDFMap.h:111-166):
// constructors, SetParameters, CheckUpdate, UpdateOccupancy, UpdateESDF, SetOccupancy x2, GetOccupancy x2,
// GetDistance x2, GetDistWithGradTrilinear, SetUpdateRange, SetOriginalRange, the public data member
// grid_total_size_, CheckConsistency / CheckWithGroundTruth.  The ROS-typed visualisation getters
// (GetPointCloud / GetSliceMarker, :144-145) are templates on the message type -- there is no ROS in this build -- next
// to plain-array equivalents (GetOccupiedVoxels, GetSlice).
//
// Header-only and free of HIP types: everything goes through the C ABI of include/fiesta_hip.h.  Array vs
// hash-block storage is a RUNTIME choice (the constructor overload), not the reference's -DHASH_TABLE macro.
//
// Differences a caller can observe (all documented in INTEGRATION.md):
//   * SetOccupancy calls are buffered on the host and applied as ONE device batch at the next
//     CheckUpdate / UpdateOccupancy / query; return values are computed on the host and are identical.
//   * single-point queries cost one device round trip each -- use the *Batch forms in planners.
//   * closest-obstacle ids are tie-equivalent, not FIFO-order-identical (parity contract).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../fiesta_hip.h"
#include "vec3.h"

namespace fiesta {

class ESDFMap {
 public:
  // dense array: ESDFMap(origin, resolution, map_size) (src/ESDFMap.cpp:171-213)
  ESDFMap(Eigen::Vector3d origin, double resolution, Eigen::Vector3d map_size, int device = 0) {
    fiesta_hip_config c = make_config(origin, resolution, device);
    c.mode = FIESTA_HIP_MODE_ARRAY;
    for (int i = 0; i < 3; ++i) c.map_size[i] = map_size(i);
    open(c);
    for (int i = 0; i < 3; ++i) {
      min_range_[i] = origin(i);
      max_range_[i] = origin(i) + map_size(i);
    }
  }
  // hash blocks: ESDFMap(origin, resolution, reserve_size) (src/ESDFMap.cpp:130-167)
  ESDFMap(Eigen::Vector3d origin, double resolution, int reserve_size = 0, int device = 0) {
    fiesta_hip_config c = make_config(origin, resolution, device);
    c.mode = FIESTA_HIP_MODE_HASH;
    c.reserve_size = reserve_size;
    open(c);
    hash_ = true;
    for (int i = 0; i < 3; ++i) {
      min_range_[i] = -1e30;
      max_range_[i] = 1e30;
    }
  }
  ~ESDFMap() {
    if (h_) fiesta_hip_destroy(h_);
  }
  ESDFMap(const ESDFMap &) = delete;
  ESDFMap &operator=(const ESDFMap &) = delete;

  void SetParameters(double p_hit, double p_miss, double p_min, double p_max, double p_occ) {
    ck(fiesta_hip_set_prob_params(h_, p_hit, p_miss, p_min, p_max, p_occ));
  }
  bool CheckUpdate() {
    Flush();
    int32_t out = 0;
    ck(fiesta_hip_check_update(h_, &out));
    return out != 0;
  }
  bool UpdateOccupancy(bool global_map) {
    Flush();
    int32_t any = 0;
    ck(fiesta_hip_update_occupancy(h_, global_map ? 1 : 0, &last_insert_, &last_delete_, &any));
    return any != 0;
  }
  void UpdateESDF() { ck(fiesta_hip_update_esdf(h_, &last_stats_)); }

  // Occupancy Management (src/ESDFMap.cpp:401-437). Returns what the reference returns.
  int SetOccupancy(Eigen::Vector3d pos, int occ) {
    if (occ != 1 && occ != 0) return FIESTA_HIP_UNDEFINED;  // "occ value error!" (:402-405)
    if (!PosInMap(pos)) return FIESTA_HIP_UNDEFINED;
    Eigen::Vector3i vox;
    Pos2Vox(pos, vox);
    return SetOccupancy(vox, occ);
  }
  int SetOccupancy(Eigen::Vector3i vox, int occ) {
    pend_vox_.push_back(vox(0));
    pend_vox_.push_back(vox(1));
    pend_vox_.push_back(vox(2));
    pend_occ_.push_back(occ);
    if (pend_occ_.size() >= kFlushAt) Flush();
    return Vox2Idx(vox);  // the caller's de-duplication key (include/Fiesta.h:221-232,253-273); unique per voxel
  }
  int GetOccupancy(Eigen::Vector3d pos) {
    Flush();
    int32_t out = 0;
    const double p[3] = {pos(0), pos(1), pos(2)};
    ck(fiesta_hip_get_occupancy_pos(h_, p, 1, &out));
    return out;
  }
  int GetOccupancy(Eigen::Vector3i vox) {
    Flush();
    int32_t out = 0;
    const int32_t v[3] = {vox(0), vox(1), vox(2)};
    ck(fiesta_hip_get_occupancy_vox(h_, v, 1, &out));
    return out;
  }

  // Distance Field Management (src/ESDFMap.cpp:467-540).  Rule for the buffered SetOccupancy calls: whatever can
  // observe them flushes them first (CheckUpdate, UpdateOccupancy, GetOccupancy, the range setters, the exports);
  // distance queries do NOT flush -- distances only change inside UpdateESDF, which only sees what UpdateOccupancy
  // (a flush) fused before it, so pending observations cannot change any answer below.
  double GetDistance(Eigen::Vector3d pos) {
    double out = 0;
    const double p[3] = {pos(0), pos(1), pos(2)};
    ck(fiesta_hip_get_distance_pos(h_, p, 1, &out));
    return out;
  }
  double GetDistance(Eigen::Vector3i vox) {
    double out = 0;
    const int32_t v[3] = {vox(0), vox(1), vox(2)};
    ck(fiesta_hip_get_distance_vox(h_, v, 1, &out));
    return out;
  }
  double GetDistWithGradTrilinear(Eigen::Vector3d pos, Eigen::Vector3d &grad) {
    double d = 0, g[3] = {0, 0, 0};
    const double p[3] = {pos(0), pos(1), pos(2)};
    ck(fiesta_hip_get_dist_grad(h_, p, 1, &d, g));
    for (int i = 0; i < 3; ++i) grad(i) = g[i];
    return d;
  }
  // batch forms (the fast path): pos is n x 3, dist n, grad n x 3 (nullable)
  void GetDistanceBatch(const double *pos, int64_t n, double *dist) { ck(fiesta_hip_get_distance_pos(h_, pos, n, dist)); }
  void GetDistWithGradTrilinearBatch(const double *pos, int64_t n, double *dist, double *grad) {
    ck(fiesta_hip_get_dist_grad(h_, pos, n, dist, grad));
  }

  // Local Range (src/ESDFMap.cpp:792-824)
  void SetUpdateRange(Eigen::Vector3d min_pos, Eigen::Vector3d max_pos, bool new_vec = true) {
    Flush();  // pending observations were made under the old window
    const double a[3] = {min_pos(0), min_pos(1), min_pos(2)}, b[3] = {max_pos(0), max_pos(1), max_pos(2)};
    ck(fiesta_hip_set_update_range(h_, a, b, new_vec ? 1 : 0));
  }
  void SetOriginalRange() {
    Flush();
    ck(fiesta_hip_set_original_range(h_));
  }

  // Visualisation, as plain arrays (reference: GetPointCloud / GetSliceMarker, src/ESDFMap.cpp:544-699)
  void GetOccupiedVoxels(std::vector<Eigen::Vector3d> *centres) {
    Flush();
    centres->clear();
    int64_t n = 0;
    ck(fiesta_hip_get_occupied_voxels(h_, nullptr, 0, &n));
    std::vector<int32_t> vox((size_t)3 * n);
    if (n) ck(fiesta_hip_get_occupied_voxels(h_, vox.data(), n, &n));
    for (int64_t i = 0; i < n; ++i)  // voxel centres, as GetPointCloud emits them (src/ESDFMap.cpp:560-575)
      centres->push_back(Eigen::Vector3d((vox[3 * i] + 0.5) * res_ + origin_[0], (vox[3 * i + 1] + 0.5) * res_ + origin_[1],
                                         (vox[3 * i + 2] + 0.5) * res_ + origin_[2]));
  }
  // The reference's own getters (include/ESDFMap.h:144-145, src/ESDFMap.cpp:544-699).  The message types are template
  // parameters so that this header builds without ROS; sensor_msgs::PointCloud and visualization_msgs::Marker fit as
  // they are (fields used: header.frame_id, points[i].x/y/z, and for the marker id, type, action, scale, pose.orientation,
  // colors[i].r/g/b/a).  Filtering, Vox2Pos and the rainbow run on the device; the order of the points is unspecified.
  template <class PointCloudMsg>
  void GetPointCloud(PointCloudMsg &m, int vis_lower_bound, int vis_upper_bound) {
    Flush();
    m.header.frame_id = "world";
    m.points.clear();
    int64_t n = 0;
    ck(fiesta_hip_get_point_cloud(h_, vis_lower_bound, vis_upper_bound, nullptr, 0, &n));
    std::vector<float> xyz((size_t)3 * n);
    if (n) ck(fiesta_hip_get_point_cloud(h_, vis_lower_bound, vis_upper_bound, xyz.data(), n, &n));
    m.points.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) m.points[i].x = xyz[3 * i], m.points[i].y = xyz[3 * i + 1], m.points[i].z = xyz[3 * i + 2];
  }
  template <class MarkerMsg, class Color /* Eigen::Vector4d; the reference ignores it too */>
  void GetSliceMarker(MarkerMsg &m, int slice, int id, Color /*color*/, double max_dist) {
    Flush();
    m.header.frame_id = "world";
    m.id = id;
    m.type = MarkerMsg::POINTS;
    m.action = MarkerMsg::MODIFY;
    m.scale.x = m.scale.y = m.scale.z = res_;
    m.pose.orientation.w = 1;
    m.pose.orientation.x = m.pose.orientation.y = m.pose.orientation.z = 0;
    m.points.clear();
    m.colors.clear();
    int64_t n = 0;
    ck(fiesta_hip_get_slice_marker(h_, slice, max_dist, nullptr, nullptr, 0, &n));
    std::vector<double> xyz((size_t)3 * n);
    std::vector<float> rgba((size_t)4 * n);
    if (n) ck(fiesta_hip_get_slice_marker(h_, slice, max_dist, xyz.data(), rgba.data(), n, &n));
    m.points.resize((size_t)n);
    m.colors.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
      m.points[i].x = xyz[3 * i], m.points[i].y = xyz[3 * i + 1], m.points[i].z = xyz[3 * i + 2];
      m.colors[i].r = rgba[4 * i], m.colors[i].g = rgba[4 * i + 1], m.colors[i].b = rgba[4 * i + 2], m.colors[i].a = rgba[4 * i + 3];
    }
  }
  // distances of the plane z = z_vox, nx * ny values, x-major (the data behind GetSliceMarker)
  void GetSlice(int z_vox, std::vector<double> *dist) {
    dist->resize((size_t)gs_[0] * gs_[1]);
    ck(fiesta_hip_get_slice(h_, z_vox, dist->data()));
  }

  // DEBUG checkers (src/ESDFMap.cpp:856-1054). The doubly-linked lists they walk do not exist here; the
  // equivalent invariant is: every finite voxel's closest obstacle is occupied and sits at the stored distance.
  bool CheckConsistency() {
    Flush();
    if (hash_) return CheckConsistencyHash();
    const size_t n = (size_t)grid_total_size_;
    std::vector<int32_t> d2(n), coc(3 * n);
    std::vector<uint8_t> occ(n);
    ck(fiesta_hip_download_field(h_, d2.data(), coc.data(), occ.data(), nullptr));
    for (size_t i = 0; i < n; ++i) {
      if (d2[i] < 0 || d2[i] == INT32_MAX) continue;
      const int64_t z = i % gs_[2], y = (i / gs_[2]) % gs_[1], x = i / ((int64_t)gs_[2] * gs_[1]);
      const int64_t cx = coc[3 * i], cy = coc[3 * i + 1], cz = coc[3 * i + 2];
      if (cx < 0 || cy < 0 || cz < 0 || cx >= gs_[0] || cy >= gs_[1] || cz >= gs_[2]) return false;
      if (!occ[(cx * gs_[1] + cy) * gs_[2] + cz]) return false;
      if ((x - cx) * (x - cx) + (y - cy) * (y - cy) + (z - cz) * (z - cz) != d2[i]) return false;
    }
    return true;
  }
  bool CheckWithGroundTruth() { return CheckConsistency(); }

  const fiesta_hip_stats &LastStats() const { return last_stats_; }
  int64_t LastInsertCount() const { return last_insert_; }
  int64_t LastDeleteCount() const { return last_delete_; }
  fiesta_hip_map *Handle() { return h_; }

  // Apply the buffered SetOccupancy calls now (called implicitly where the result could be observed).
  void Flush() {
    if (pend_occ_.empty()) return;
    ck(fiesta_hip_set_occupancy_vox(h_, pend_vox_.data(), pend_occ_.data(), (int64_t)pend_occ_.size(), nullptr));
    pend_vox_.clear();
    pend_occ_.clear();
  }

  int grid_total_size_ = 0;  // public in the reference's array build (include/ESDFMap.h:115)

 private:
  static constexpr size_t kFlushAt = 1u << 20;
  fiesta_hip_map *h_ = nullptr;
  bool hash_ = false;
  double origin_[3], res_ = 0, min_range_[3], max_range_[3];
  int32_t gs_[3] = {0, 0, 0};
  int64_t last_insert_ = 0, last_delete_ = 0;
  fiesta_hip_stats last_stats_{};
  std::vector<int32_t> pend_vox_, pend_occ_;

  fiesta_hip_config make_config(const Eigen::Vector3d &origin, double resolution, int device) {
    fiesta_hip_config c{};
    c.device = device;
    c.resolution = resolution;
    res_ = resolution;
    for (int i = 0; i < 3; ++i) c.origin[i] = origin_[i] = origin(i);
    return c;
  }
  void open(const fiesta_hip_config &c) {
    if (fiesta_hip_create(&c, &h_) != FIESTA_HIP_OK)
      throw std::runtime_error(std::string("fiesta::ESDFMap: ") + fiesta_hip_last_error());
    int64_t n = 0;
    ck(fiesta_hip_grid_size(h_, gs_));
    ck(fiesta_hip_grid_total_size(h_, &n));
    grid_total_size_ = (int)n;
  }
  static void ck(int status) {
    if (status != FIESTA_HIP_OK) throw std::runtime_error(std::string("fiesta::ESDFMap: ") + fiesta_hip_last_error());
  }
  bool PosInMap(const Eigen::Vector3d &p) const {  // src/ESDFMap.cpp:46-61
    for (int i = 0; i < 3; ++i)
      if (p(i) < min_range_[i] || p(i) > max_range_[i]) return false;
    return true;
  }
  void Pos2Vox(const Eigen::Vector3d &p, Eigen::Vector3i &v) const {  // :74-77
    for (int i = 0; i < 3; ++i) v(i) = (int)std::floor((p(i) - origin_[i]) / res_);
  }
  int Vox2Idx(const Eigen::Vector3i &v) const {  // :84-93; hash flavour: see fiesta_hip_voxel_key
    if (!hash_) return v(0) * gs_[1] * gs_[2] + v(1) * gs_[2] + v(2);
    const int32_t vox[3] = {v(0), v(1), v(2)};
    int32_t key = FIESTA_HIP_UNDEFINED;
    ck(fiesta_hip_voxel_key(h_, vox, 1, &key));
    return key;
  }
  bool CheckConsistencyHash() {  // the same invariant over the allocated voxels of a hash-block map
    int64_t n = 0;
    ck(fiesta_hip_download_hash(h_, &n, nullptr, nullptr, nullptr, nullptr));
    std::vector<int32_t> vox(3 * (size_t)n), d2((size_t)n), coc(3 * (size_t)n);
    std::vector<uint8_t> occ((size_t)n);
    if (n) ck(fiesta_hip_download_hash(h_, &n, vox.data(), d2.data(), coc.data(), occ.data()));
    auto key = [](int64_t x, int64_t y, int64_t z) { return ((x + (1 << 20)) << 42) | ((y + (1 << 20)) << 21) | (z + (1 << 20)); };
    std::vector<int64_t> occupied;
    for (int64_t i = 0; i < n; ++i)
      if (occ[i]) occupied.push_back(key(vox[3 * i], vox[3 * i + 1], vox[3 * i + 2]));
    std::sort(occupied.begin(), occupied.end());
    for (int64_t i = 0; i < n; ++i) {
      if (d2[i] < 0 || d2[i] == INT32_MAX) continue;
      const int64_t dx = vox[3 * i] - coc[3 * i], dy = vox[3 * i + 1] - coc[3 * i + 1], dz = vox[3 * i + 2] - coc[3 * i + 2];
      if (dx * dx + dy * dy + dz * dz != d2[i]) return false;
      if (!std::binary_search(occupied.begin(), occupied.end(), key(coc[3 * i], coc[3 * i + 1], coc[3 * i + 2]))) return false;
    }
    return true;
  }
};

// Raycast(start, end, min, max, &output) (include/raycast.h:16-18, src/raycast.cpp:56-158); voxel units.
inline void Raycast(const Eigen::Vector3d &start, const Eigen::Vector3d &end, const Eigen::Vector3d &min,
                    const Eigen::Vector3d &max, std::vector<Eigen::Vector3d> *output, int device = 0) {
  const double a[3] = {start(0), start(1), start(2)}, b[3] = {end(0), end(1), end(2)};
  const double lo[3] = {min(0), min(1), min(2)}, hi[3] = {max(0), max(1), max(2)};
  std::vector<double> buf(3 * 1502);
  int32_t n = 0;
  if (fiesta_hip_raycast_single(a, b, lo, hi, buf.data(), 1502, &n, device) != FIESTA_HIP_OK)
    throw std::out_of_range("Too many RaycasMultithread voxels");  // src/raycast.cpp:127-130
  output->clear();
  for (int i = 0; i < n; ++i) output->push_back(Eigen::Vector3d(buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]));
}


// One sensor frame for a -DSIGNED_NEEDED pair (include/Fiesta.h:39-41, 216-218, 249-251, 515-518): what
// Fiesta::RaycastMultithread does to esdf_map_ and inv_esdf_map_ -- the inverse map sees the end points as free and the
// crossed voxels as occupied.  points: n x 3 floats in the sensor frame, transform row-major 4x4.  Both maps must have
// the same geometry; the caller then runs UpdateOccupancy / UpdateESDF on both (:507-518).
inline void RaycastFrameSigned(ESDFMap &esdf_map, ESDFMap &inv_esdf_map, const float *points, int64_t n,
                               const double transform[16], const Eigen::Vector3d &raycast_origin, double min_ray_length,
                               double max_ray_length, const Eigen::Vector3d &l_cornor, const Eigen::Vector3d &r_cornor) {
  const double o[3] = {raycast_origin(0), raycast_origin(1), raycast_origin(2)};
  esdf_map.Flush();  // buffered SetOccupancy calls come BEFORE the frame, as they would in the reference
  inv_esdf_map.Flush();
  fiesta_hip_raycast_params p{min_ray_length, max_ray_length, {l_cornor(0), l_cornor(1), l_cornor(2)},
                              {r_cornor(0), r_cornor(1), r_cornor(2)}, /*dedup=*/1, /*inverse=*/0};
  if (fiesta_hip_raycast_frame(esdf_map.Handle(), points, n, transform, o, &p) != FIESTA_HIP_OK)
    throw std::runtime_error(fiesta_hip_last_error());
  p.inverse = 1;
  if (fiesta_hip_raycast_frame(inv_esdf_map.Handle(), points, n, transform, o, &p) != FIESTA_HIP_OK)
    throw std::runtime_error(fiesta_hip_last_error());
}
// ... and the quantity the pair exists for (the reference leaves it as a TODO): distance to the nearest obstacle minus
// distance to the nearest voxel observed free -- positive in free space, negative inside obstacles.
// NaN where either map holds no distance there (-10000 outside the map / +10000 unobserved or no obstacle), like the
// Python helper fiesta_amd.signed_distance.
inline double SignedDistance(ESDFMap &esdf_map, ESDFMap &inv_esdf_map, const Eigen::Vector3d &pos) {
  const double d = esdf_map.GetDistance(pos), di = inv_esdf_map.GetDistance(pos);
  if (!(std::fabs(d) < 10000.0) || !(std::fabs(di) < 10000.0)) return std::nan("");
  return d - di;
}

}  // namespace fiesta
