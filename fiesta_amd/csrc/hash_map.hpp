// fiesta_amd/csrc/hash_map.hpp -- hash-of-8^3-blocks ESDF map resident in HBM (host-side class).
// Replaces the HASH_TABLE+BLOCK+BITWISE flavour of fiesta::ESDFMap (src/ESDFMap.cpp:130-167,705-765).
#pragma once
#include "../../include/fiesta_hip.h"
#include "common.hpp"

namespace fiesta {

class HashMap {
 public:
  explicit HashMap(const fiesta_hip_config &) { throw Error(FIESTA_HIP_ERR_INVALID, "hash mode: not built yet"); }
  int64_t allocated_voxels() { return 0; }
  void set_prob_params(double, double, double, double, double) {}
  void set_update_range(const double *, const double *, bool) {}
  void set_original_range() {}
  void observe_vox(const int32_t *, const int32_t *, int64_t, int32_t *) {}
  void observe_pos(const double *, const int32_t *, int64_t, int32_t *) {}
  void raycast_frame(const float *, int64_t, const double *, const double *, const fiesta_hip_raycast_params *) {}
  bool check_update() { return false; }
  bool update_occupancy(bool, int64_t *, int64_t *) { return false; }
  void update_esdf(fiesta_hip_stats *) {}
  void get_distance_vox(const int32_t *, int64_t, double *) {}
  void get_distance_pos(const double *, int64_t, double *) {}
  void get_dist_grad(const double *, int64_t, double *, double *) {}
  void get_occupancy_vox(const int32_t *, int64_t, int32_t *) {}
  void get_occupancy_pos(const double *, int64_t, int32_t *) {}
  int64_t download(int32_t *, int32_t *, int32_t *, uint8_t *) { return 0; }
  void synchronize() {}
};

void raycast_single(const double *start, const double *end, const double *minv, const double *maxv, double *out,
                    int32_t cap, int32_t *n_out, int32_t device);

}  // namespace fiesta
