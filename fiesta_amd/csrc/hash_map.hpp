// fiesta_amd/csrc/hash_map.hpp -- sparse ("hash-block") ESDF map resident in HBM (host-side class).
// Replaces the HASH_TABLE+BLOCK+BITWISE flavour of fiesta::ESDFMap (src/ESDFMap.cpp:130-167,705-765).
//
// The reference keeps `unordered_map<blockId, base index>` over 8^3-voxel blocks appended to growing vectors
// (FindAndInsert, :732-765; capacity doubles, :705-730). The GPU equivalent here is a PAGE POOL behind a dense
// PAGE DIRECTORY: a page is one relaxation tile (16 x 16 x 32 voxels, z fastest -- 32 KiB of 4-byte words, 256
// whole bitmap words), the directory is a flat array over the 64 x 64 x 32 tiles of a 1024^3-voxel WINDOW (512 KiB),
// so a lookup is one load, never a probe chain, and the relaxation kernel only swaps its address arithmetic
// (k_relax_q<.., PAGED>). Memory stays proportional to the observed space; the pool doubles when it runs out (like
// IncreaseCapacity). Pages are allocated when a voxel in them is OBSERVED; the reference additionally allocates the
// blocks its neighbour READS touch (they stay unobserved forever and never influence a distance) -- see DESIGN.md.
//
// The map itself is UNBOUNDED like the reference's (any int voxel coordinate): the window MOVES.  It starts centred on
// the map origin; an observation batch (or ray-cast frame) whose bounding box does not fit the current window recentres
// it, per axis, on that box (ensure_window).  Pages whose tile leaves the window are PARKED: they keep their content and
// their place in the pool and in download() and still ANSWER queries (through the map-wide page table, PageTable below),
// but take no part in observations or UpdateESDF while parked; when
// the window comes back over them they are re-attached and their distance field is rebuilt from the obstacles in and
// around them at the next UpdateESDF (like voxels behind a deleted obstacle).  Inside the window the field is the ESDF
// of the obstacles INSIDE THE WINDOW.  Closest-obstacle ids are map coordinates modulo 1024 decoded relative to their
// voxel (common.hpp), so they never change when the window moves.
#pragma once
#include "../../include/fiesta_hip.h"
#include "common.hpp"
#include "dense_map.hpp"

namespace fiesta {

// every page's tile in MAP coordinates, sorted: what a query outside the window searches (hash_map.hip: h_lookup)
struct PageTable {
  const unsigned long long *keys;  // (tx + 2^20) << 42 | (ty + 2^20) << 21 | (tz + 2^20)
  const int32_t *pages;
  int n;
};

class HashMap {
 public:
  static constexpr int kWin = 1024, kHalf = 512;   // window, voxels per axis / half of it (initial offset of map voxel 0)
  static constexpr int kTX = 16, kTY = 16, kTZ = 32;
  static constexpr int kNTX = kWin / kTX, kNTY = kWin / kTY, kNTZ = kWin / kTZ;
  static constexpr int kNTiles = kNTX * kNTY * kNTZ;
  static constexpr int kPageVox = kTX * kTY * kTZ, kPageRows = kTX * kTY;

  // what SetOccupancy returns for a voxel: its map coordinates modulo 1024, packed -- unique among the voxels of any one
  // window position (see fiesta_hip_voxel_key); never -10000: no voxel is outside an unbounded map
  static int32_t voxel_key(int vx, int vy, int vz) { return (int32_t)pack_coc(vx, vy, vz); }
  void window_origin(int32_t out[3]) const { out[0] = g_.gx0, out[1] = g_.gy0, out[2] = g_.gz0; }
  int64_t window_moves() const { return moves_; }
  // move the window so that map voxel `centre` is at its middle (rounded to whole tiles), whatever it holds now
  void recentre(const int32_t centre[3]);
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
  void set_update_engine(int e) { update_engine_ = e; }
  int level_trace(uint32_t *out48) const;  // fiesta_hip_level_trace
  void level_tuning(int grid_groups, long long spin_limit);  // fiesta_hip_level_tuning
  void get_distance_vox(const int32_t *vox, int64_t n, double *out);
  void get_distance_pos(const double *pos, int64_t n, double *out);
  void get_dist_grad(const double *pos, int64_t n, double *dist, double *grad);
  void get_occupancy_vox(const int32_t *vox, int64_t n, int32_t *out);
  void get_occupancy_pos(const double *pos, int64_t n, int32_t *out);
  // every voxel of every allocated page, page order: vox (map voxel coordinates), d2, coc, occ; returns the count
  int64_t download(int32_t *vox, int32_t *d2, int32_t *coc, uint8_t *occ);
  void download_counts(int32_t *num_hit, int32_t *num_miss);  // same order as download()
  void checkpoint(const char *path, bool write);  // raw dump / load of the whole state (checkpoint.hpp)
  int64_t point_cloud(int vis_lower_bound, int vis_upper_bound, float *xyz, int64_t cap);  // GetPointCloud, as arrays
  int64_t slice_marker(int slice, double max_dist, double *xyz, float *rgba, int64_t cap);  // GetSliceMarker
  void synchronize();

 private:
  void use_device() const;
  void ensure_window(const int64_t lo[3], const int64_t hi[3]);  // bounding box of a batch, map voxels, inclusive
  void ensure_window_vox(const int32_t *vox, int64_t n, bool dev);
  void move_window(const int32_t origin[3]);
  void refresh_range();
  void ensure_pages(int64_t need_total);
  void pristine_pages(int64_t first, int64_t count);
  bool allocate_marked();
  unsigned long long read_counter(int which);
  void zero_counter(int which);
  void zero_counters(int first, int n);
  void run_rounds(fiesta_hip_stats *st, uint32_t first_count);
  bool run_levels(fiesta_hip_stats *st, unsigned long long ni, unsigned long long nd, bool scan);  // false: the rounds finish
  PageTable page_table();  // the map-wide page table of the query kernels (hash_map.hip)
  // scalar queries (n <= kHostQueries positions per call, host pointers): a host-side cache of 16^3-voxel bricks of DISTANCES (f64,
  // as the query kernels compute them) and occupancy bits, keyed by map brick coordinates, fetched on first touch (r06: the dense
  // map's HostBricks for the paged map).  field_epoch_ is bumped by everything that may change the field or what a lookup finds.
  static constexpr int64_t kHostQueries = 8;
  struct HostBricks;
  HostBricks *bricks_ = nullptr;
  uint64_t field_epoch_ = 1;
  const double *host_brick(int vx, int vy, int vz);
  double host_distance(int vx, int vy, int vz);
  int host_occ(int vx, int vy, int vz);
 public:
  int64_t host_brick_fetches() const;
 private:
  DevBuf<unsigned long long> ptab_keys_;
  DevBuf<int32_t> ptab_pages_;
  int64_t ptab_pages_built_ = -1;
  void free_raycast_state();
  struct RaycastState;
  RaycastState *rc_ = nullptr;

  Geom g_;  // the window as a "grid": nx = ny = nz = 1024; (gx0, gy0, gz0) = map voxel of window voxel 0; wrap = 1
  int64_t ur_[6], pr_[6];    // SetUpdateRange's current / previous box in MAP voxels (min xyz, max xyz); g_.w*/p* derive
  bool force_scan_ = false;  // the window moved: the next UpdateESDF revalidates every resident voxel
  int64_t moves_ = 0;
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
  DevBuf<int32_t> page_tile_;   // window tile of a page, -1 while it is parked
  DevBuf<int32_t> page_gtile_;  // 3 per page: its tile in MAP tile coordinates (voxel / (16, 16, 32)), never changes
  DevBuf<uint32_t> page_fresh_; // 1: re-attached by the last window move, not yet rebuilt
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
  unsigned long long host_ni_ = 0, host_nd_ = 0;  // insert / delete queue lengths as last read
  bool host_queues_valid_ = false;
  int chain_hint_ = 4;  // rounds in the first chain of the next update (the previous update's count + 1)
  // UpdateESDF engine: 1 = frontier rounds only, 3 = level engine whenever its lists hold the update, else: level engine
  // for updates of at most small_update_ voxels (a sensor frame), rounds otherwise
  int update_engine_ = 0;
  int small_update_ = 4096;
  LevelEngine *lv_ = nullptr;  // level_kernels.hpp
  hipEvent_t lv_done_ = nullptr;
  int64_t dropped_host_ = 0;  // voxels of observe_box() requests clipped away by the window (added to C_DROPPED in stats)
  unsigned long long *counters_ = nullptr, *h_counters_ = nullptr;
  DevBuf<unsigned char> stage_a_, stage_b_, stage_c_, stage_d_;
};

void raycast_single(const double *start, const double *end, const double *minv, const double *maxv, double *out,
                    int32_t cap, int32_t *n_out, int32_t device);

}  // namespace fiesta
