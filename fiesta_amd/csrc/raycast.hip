// placeholder, replaced below in this round
#include "dense_map.hpp"
#include "hash_map.hpp"
namespace fiesta {
void DenseMap::raycast_frame(const float *, int64_t, const double *, const double *, const fiesta_hip_raycast_params *, bool) {
  throw Error(FIESTA_HIP_ERR_INVALID, "raycast: not built yet");
}
void DenseMap::raycast_depth(const uint16_t *, int, int, double, double, double, double, const double *, const double *, const fiesta_hip_raycast_params *) {
  throw Error(FIESTA_HIP_ERR_INVALID, "raycast: not built yet");
}
void raycast_single(const double *, const double *, const double *, const double *, double *, int32_t, int32_t *, int32_t) {
  throw Error(FIESTA_HIP_ERR_INVALID, "raycast: not built yet");
}
}
