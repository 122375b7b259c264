// fiesta_amd/csrc/hash_map.hpp -- sparse ("hash-block") ESDF map resident in HBM (host-side class).
// Replaces the HASH_TABLE+BLOCK+BITWISE flavour of fiesta::ESDFMap (src/ESDFMap.cpp:130-167,705-765).
//
// The reference keeps `unordered_map<blockId, base index>` over 8^3-voxel blocks appended to growing vectors
// (FindAndInsert, :732-765; capacity doubles, :705-730). The GPU equivalent here is a PAGE POOL behind a dense
// PAGE DIRECTORY: a page is one relaxation tile (16 x 16 x 32 voxels, z fastest -- 32 KiB of 4-byte words, 256
// whole bitmap words), the directory is a flat array over the 64 x 64 x 32 tiles of a 1024^3-voxel virtual
// window centred on the map origin (512 KiB), so a lookup is one load, never a probe chain, and the relaxation
// kernel only swaps its address arithmetic (k_relax_q<.., PAGED>). Memory stays proportional to the observed
// space; the pool doubles when it runs out (like IncreaseCapacity). Pages are allocated when a voxel in them is
// OBSERVED; the reference additionally allocates the blocks its neighbour READS touch (they stay unobserved
// forever and never influence a distance) -- see DESIGN.md.
#pragma once
#include "../../include/fiesta_hip.h"
#include "common.hpp"
#include "dense_map.hpp"

namespace fiesta {

class HashMap {
 public:
  static constexpr int kWin = 1024, kHalf = 512;   // virtual window, voxels per axis / offset of voxel 0
  static constexpr int kTX = 16, kTY = 16, kTZ = 32;
  static constexpr int kNTX = kWin / kTX, kNTY = kWin / kTY, kNTZ = kWin / kTZ;
  static constexpr int kNTiles = kNTX * kNTY * kNTZ;
  static constexpr int kPageVox = kTX * kTY * kTZ, kPageRows = kTX * kTY;

  // what SetOccupancy returns for a voxel: a key unique per voxel of the window, -10000 outside (see fiesta_hip_voxel_key)
  static int32_t voxel_key(int vx, int vy, int vz) {
    const int x = vx + kHalf, y = vy + kHalf, z = vz + kHalf;
    const bool ok = (unsigned)x < (unsigned)kWin && (unsigned)y < (unsigned)kWin && (unsigned)z < (unsigned)kWin;
    return ok ? (int32_t)pack_coc(x, y, z) : FIESTA_HIP_UNDEFINED;
  }
  explicit HashMap(const fiesta_hip_config &cfg);
  ~HashMap();

  int64_t allocated_voxels() { return (int64_t)npages_ * kPageVox; }
  int64_t allocated_pages() const { return npages_; }
  void set_prob_params(double p_hit, double p_miss, double p_min, double p_max, double p_occ);
  void set_update_range(const double *mn, const double *mx, bool new_vec);
  void set_original_range();
  void observe_vox(const int32_t *vox, const int32_t *occ, int64_t n, int32_t *ret, bool dev = false);
  void observe_box(const int32_t *lo, const int32_t *hi, int occ);  // inclusive voxel box, map voxel coordinates
  void snapshot_save();                                             // copy of the state words (benchmark unit only)
  int64_t snapshot_count_updated();
  void observe_pos(const double *pos, const int32_t *occ, int64_t n, int32_t *ret);
  // ray-cast front end (raycast.hip): points are host or device (dev) n x 3 floats / a host uint16 depth image
  void raycast_frame(const float *points, int64_t n, const double *T, const double *origin, const fiesta_hip_raycast_params *p,
                     bool dev);
  void raycast_depth(const uint16_t *depth, int rows, int cols, double fx, double fy, double cx, double cy, const double *T,
                     const double *origin, const fiesta_hip_raycast_params *p,
                     const fiesta_hip_depth_filter *f = nullptr);
  bool check_update();
  bool update_occupancy(bool global_map, int64_t *n_ins, int64_t *n_del);
  void update_esdf(fiesta_hip_stats *st);
  void get_distance_vox(const int32_t *vox, int64_t n, double *out);
  void get_distance_pos(const double *pos, int64_t n, double *out);
  void get_dist_grad(const double *pos, int64_t n, double *dist, double *grad);
  void get_occupancy_vox(const int32_t *vox, int64_t n, int32_t *out);
  void get_occupancy_pos(const double *pos, int64_t n, int32_t *out);
  // every voxel of every allocated page, page order: vox (map voxel coordinates), d2, coc, occ; returns the count
  int64_t download(int32_t *vox, int32_t *d2, int32_t *coc, uint8_t *occ);
  void download_counts(int32_t *num_hit, int32_t *num_miss);  // same order as download()
  void synchronize();

 private:
  void use_device() const;
  void ensure_pages(int64_t need_total);
  void allocate_marked();
  unsigned long long read_counter(int which);
  void zero_counter(int which);
  void run_rounds(fiesta_hip_stats *st, uint32_t first_count);
  void free_raycast_state();
  struct RaycastState;
  RaycastState *rc_ = nullptr;

  Geom g_;  // the virtual window as a "grid": nx = ny = nz = 1024, coordinates offset by kHalf
  ProbParams pp_{0, 0, 0, 0, 0};
  int device_ = 0;
  hipStream_t stream_ = nullptr;
  hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
  int prof_ = 0;

  // page pool (grows), indexed by page * kPageVox + in-page offset / page * kPageRows + row
  DevBuf<vox_t> coc_;
  DevBuf<double> logodds_;
  DevBuf<unsigned long long> cnt_;
  DevBuf<uint32_t> occbits_, rbits_, cbits_[2];
  DevBuf<int32_t> page_tile_;
  DevBuf<vox_t> shadow_;
  int64_t shadow_vox_ = -1;
  int64_t npages_ = 0, cap_pages_ = 0;

  // directory space (fixed size kNTiles)
  int32_t *dir_ = nullptr;
  uint32_t *need_ = nullptr;
  uint32_t *tile_epoch_ = nullptr, *cstamp_[2] = {nullptr, nullptr};
  uint32_t *tile_flag_[2] = {nullptr, nullptr}, *tile_list_[2] = {nullptr, nullptr};
  uint32_t epoch_ = 0, serial_ = 0;

  DevBuf<uint32_t> touched_, ins_, del_;
  int64_t touched_upper_ = 0;
  int64_t dropped_host_ = 0;  // voxels of observe_box() requests clipped away by the window (added to C_DROPPED in stats)
  unsigned long long *counters_ = nullptr, *h_counters_ = nullptr;
  DevBuf<unsigned char> stage_a_, stage_b_, stage_c_, stage_d_;
};

void raycast_single(const double *start, const double *end, const double *minv, const double *maxv, double *out,
                    int32_t cap, int32_t *n_out, int32_t device);

}  // namespace fiesta
