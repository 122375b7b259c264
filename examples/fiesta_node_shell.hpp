// examples/fiesta_node_shell.hpp -- the call sites of the reference's node, WITHOUT ROS, written once against "a map
// type with the reference's ESDFMap surface" and compiled twice:
//   * against include/fiesta/ESDFMap.h (the HIP drop-in class)          -> examples/node_shell_demo.cpp
//   * against the verbatim reference class /root/reference/include/ESDFMap.h -> oracle/node_shell_ref.cpp (test infrastructure)
// so that the substitution INTEGRATION.md describes is something a compiler has checked, not prose
// (tests/test_node_shell.py runs both on the same frames and compares what the maps end up holding).
//
// What is restated here, and from where (paths into the reference tree):
//   NodeShell::NodeShell        Fiesta::Fiesta                include/Fiesta.h:88-133   map construction (array / hash
//                                                                                        overload), SetParameters, the
//                                                                                        per-frame stamp arrays set_free_ /
//                                                                                        set_occ_ sized grid_total_size_
//   NodeShell::RaycastProcess   Fiesta::RaycastProcess        include/Fiesta.h:194-278  per point: SetOccupancy(Vector3d,int),
//                                                                                        the de-dup keyed by its return value,
//                                                                                        the free function Raycast, the walk
//   NodeShell::RaycastMultithread  Fiesta::RaycastMultithread include/Fiesta.h:281-303  ray_cast_num_thread_ == 0
//   NodeShell::UpdateEsdfEvent  Fiesta::UpdateEsdfEvent       include/Fiesta.h:481-539  CheckUpdate -> range -> UpdateOccupancy
//                                                                                        -> UpdateESDF (PROBABILISTIC build)
// Not restated: subscribers, message synchronisation, depth conversion, visualisation, timing (ROS / OpenCV / PCL types).
// `SetFrame` stands for what SynchronizationAndProcess leaves behind for a frame (:415-429): cloud_, transform_,
// raycast_origin_.  HASH is the reference's compile-time switch HASH_TABLE as a template parameter.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <unordered_set>
#include <vector>

namespace fiesta_shell {

struct Parameters {  // the fields of fiesta::Parameters these call sites read (include/parameters.h:133-167)
  double resolution_ = 0.1;
  Eigen::Vector3d l_cornor_, r_cornor_, map_size_, radius_;
  double min_ray_length_ = 0.5, max_ray_length_ = 5.0;
  double p_hit_ = 0.70, p_miss_ = 0.35, p_min_ = 0.12, p_max_ = 0.97, p_occ_ = 0.80;  // src/parameters.cpp:28-32
  bool global_update_ = true;
  int reserved_size_ = 1000000;
};

template <class Map, bool HASH>
class NodeShell {
 public:
  explicit NodeShell(const Parameters &p) : parameters_(p) {
    if constexpr (HASH)  // (each build of the reference has only its own constructor)
      esdf_map_ = new Map(Eigen::Vector3d(0, 0, 0), parameters_.resolution_, parameters_.reserved_size_);
    else
      esdf_map_ = new Map(parameters_.l_cornor_, parameters_.resolution_, parameters_.map_size_);
    esdf_map_->SetParameters(parameters_.p_hit_, parameters_.p_miss_, parameters_.p_min_, parameters_.p_max_, parameters_.p_occ_);
    if constexpr (!HASH) {
      set_free_.resize(grid_total(*esdf_map_));
      set_occ_.resize(grid_total(*esdf_map_));
      std::fill(set_free_.begin(), set_free_.end(), 0);
      std::fill(set_occ_.begin(), set_occ_.end(), 0);
    }
  }
  ~NodeShell() { delete esdf_map_; }
  NodeShell(const NodeShell &) = delete;
  NodeShell &operator=(const NodeShell &) = delete;

  // one synchronised sensor frame: the cloud in the sensor frame, transform_ (row-major 4x4), raycast_origin_, cur_pos_
  void SetFrame(const float *points, size_t n, const double transform[16], const Eigen::Vector3d &origin) {
    cloud_.assign(points, points + 3 * n);
    for (int k = 0; k < 16; ++k) transform_[k] = transform[k];
    raycast_origin_ = origin;
    sync_pos_ = origin;
    new_msg_ = true;
  }

  void RaycastMultithread() {
    if (HASH) {
      hset_free_.clear();
      hset_occ_.clear();
    }
    const int tt = ++tot_;
    RaycastProcess(0, (int)(cloud_.size() / 3), tt);
  }

  void UpdateEsdfEvent() {
    if (!new_msg_) return;
    new_msg_ = false;
    cur_pos_ = sync_pos_;
    esdf_cnt_++;
    if (esdf_map_->CheckUpdate()) {
      if (parameters_.global_update_)
        esdf_map_->SetOriginalRange();
      else
        esdf_map_->SetUpdateRange(cur_pos_ - parameters_.radius_, cur_pos_ + parameters_.radius_);
      esdf_map_->UpdateOccupancy(parameters_.global_update_);
      esdf_map_->UpdateESDF();
    }
  }

  Map *esdf_map_ = nullptr;
  int esdf_cnt_ = 0;

 private:
  // grid_total_size_ is a public data member of the reference's ARRAY build only (include/ESDFMap.h:114-116)
  template <class M>
  static auto grid_total(M &m) -> decltype((size_t)m.grid_total_size_) {
    return (size_t)m.grid_total_size_;
  }

  void RaycastProcess(int i, int part, int tt) {
    using Eigen::Vector3d;
    const Vector3d half = Vector3d(0.5, 0.5, 0.5);
    for (int idx = part * i; idx < part * (i + 1); idx++) {
      std::vector<Vector3d> output;
      if ((size_t)idx >= cloud_.size() / 3) break;  // (the reference tests `>`: off by one, include/Fiesta.h:198)
      const float px = cloud_[3 * idx], py = cloud_[3 * idx + 1], pz = cloud_[3 * idx + 2];
      int cnt = 0;
      if (std::isnan(px) || std::isnan(py) || std::isnan(pz)) continue;
      double h[4];  // transform_ * Vector4d(pt.x, pt.y, pt.z, 1)
      for (int r = 0; r < 4; ++r)
        h[r] = transform_[4 * r] * (double)px + transform_[4 * r + 1] * (double)py + transform_[4 * r + 2] * (double)pz + transform_[4 * r + 3] * 1.0;
      Vector3d point = Vector3d(h[0], h[1], h[2]) / h[3];

      int tmp_idx;
      double length = (point - raycast_origin_).norm();
      if (length < parameters_.min_ray_length_)
        continue;
      else if (length > parameters_.max_ray_length_) {
        point = (point - raycast_origin_) / length * parameters_.max_ray_length_ + raycast_origin_;
        tmp_idx = esdf_map_->SetOccupancy((Vector3d)point, 0);
      } else
        tmp_idx = esdf_map_->SetOccupancy((Vector3d)point, 1);

      if (tmp_idx != -10000) {
        if (HASH) {
          if (hset_occ_.find(tmp_idx) != hset_occ_.end())
            continue;
          else
            hset_occ_.insert(tmp_idx);
        } else {
          if (set_occ_[tmp_idx] == tt)
            continue;
          else
            set_occ_[tmp_idx] = tt;
        }
      }
      Raycast(raycast_origin_ / parameters_.resolution_, point / parameters_.resolution_, parameters_.l_cornor_ / parameters_.resolution_,
              parameters_.r_cornor_ / parameters_.resolution_, &output);

      for (int k = (int)output.size() - 2; k >= 0; k--) {
        Vector3d tmp = (output[k] + half) * parameters_.resolution_;
        length = (tmp - raycast_origin_).norm();
        if (length < parameters_.min_ray_length_) break;
        if (length > parameters_.max_ray_length_) continue;
        int fidx = esdf_map_->SetOccupancy(tmp, 0);
        if (fidx != -10000) {
          if (HASH) {
            if (hset_free_.find(fidx) != hset_free_.end()) {
              if (++cnt >= 1) {
                cnt = 0;
                break;
              }
            } else {
              hset_free_.insert(fidx);
              cnt = 0;
            }
          } else {
            if (set_free_[fidx] == tt) {
              if (++cnt >= 1) {
                cnt = 0;
                break;
              }
            } else {
              set_free_[fidx] = tt;
              cnt = 0;
            }
          }
        }
      }
    }
  }

  Parameters parameters_;
  std::vector<float> cloud_;
  double transform_[16];
  Eigen::Vector3d raycast_origin_, sync_pos_, cur_pos_;
  bool new_msg_ = false;
  int tot_ = 0;
  std::vector<int> set_free_, set_occ_;             // array build (include/Fiesta.h:107-110)
  std::unordered_set<int> hset_free_, hset_occ_;    // HASH_TABLE build
};

// ---- the frames both drivers read: a small binary file written by the test ------------------------------------------
//   int32 n_frames, n_points;  per frame: double T[16], double origin[3], float points[n_points][3]
struct Frames {
  int n_frames = 0, n_points = 0;
  std::vector<double> T, origin;
  std::vector<float> points;
  bool read(const char *path) {
    FILE *f = std::fopen(path, "rb");
    if (!f) return false;
    bool ok = std::fread(&n_frames, 4, 1, f) == 1 && std::fread(&n_points, 4, 1, f) == 1 && n_frames > 0 && n_points > 0;
    if (ok) {
      T.resize((size_t)16 * n_frames), origin.resize((size_t)3 * n_frames), points.resize((size_t)3 * n_points * n_frames);
      for (int k = 0; ok && k < n_frames; ++k)
        ok = std::fread(&T[16 * k], 8, 16, f) == 16 && std::fread(&origin[3 * k], 8, 3, f) == 3 &&
             std::fread(&points[(size_t)3 * n_points * k], 4, (size_t)3 * n_points, f) == (size_t)3 * n_points;
    }
    std::fclose(f);
    return ok;
  }
};
inline unsigned long long fnv(const void *p, size_t bytes, unsigned long long h = 1469598103934665603ull) {
  const unsigned char *c = (const unsigned char *)p;
  for (size_t i = 0; i < bytes; ++i) h = (h ^ c[i]) * 1099511628211ull;
  return h;
}

}  // namespace fiesta_shell
