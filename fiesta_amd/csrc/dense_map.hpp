// fiesta_amd/csrc/dense_map.hpp -- dense-array ESDF map resident in HBM (host-side class).
// Replaces the dense flavour of fiesta::ESDFMap (include/ESDFMap.h:37-166, src/ESDFMap.cpp).
#pragma once
#include <chrono>
#include <vector>

#include "../../include/fiesta_hip.h"
#include "common.hpp"

namespace fiesta {

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t cap = 0;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  // Exactly n elements if there are fewer now, contents dropped (large scratch: no doubling).
  void ensure_exact(size_t n, hipStream_t s) {
    if (n <= cap) return;
    if (p) {
      FIESTA_HIP_CHECK(hipStreamSynchronize(s));
      (void)hipFree(p);
      p = nullptr, cap = 0;
    }
    FIESTA_HIP_CHECK(hipMalloc((void **)&p, n * sizeof(T)));
    cap = n;
  }
  // Grow to at least n elements; optionally preserve the first `keep` elements.
  void ensure(size_t n, hipStream_t s, size_t keep = 0) {
    if (n <= cap) return;
    size_t ncap = cap ? cap : 1024;
    while (ncap < n) ncap *= 2;
    T *q = nullptr;
    FIESTA_HIP_CHECK(hipMalloc((void **)&q, ncap * sizeof(T)));
    if (p && keep) FIESTA_HIP_CHECK(hipMemcpyAsync(q, p, keep * sizeof(T), hipMemcpyDeviceToDevice, s));
    if (p) {
      FIESTA_HIP_CHECK(hipStreamSynchronize(s));
      (void)hipFree(p);
    }
    p = q;
    cap = ncap;
  }
};

// Device-side counters, one 64-bit word each.
enum Counter {
  C_TOUCHED = 0,   // length of the touched-voxel list (the reference's occupancy_queue_)
  C_INSERT,        // insert_queue_
  C_DELETE,        // delete_queue_
  C_OBSERVED,      // voxels observed at least once (first-observation transitions counted by k_fuse)
  C_NOCC,          // occupied voxels (Exist() count), kept by k_fuse
  C_DROPPED,       // hash-block maps: observations that fell outside the addressable window and were ignored (cumulative)
  C_MAXD2,         // upper bound of every finite d^2 the work-queue engine ever stored (bounds the delete scan)
  C_DBOX0,         // bounding box of the pending delete queue, local voxel coordinates: min x,y,z then max x,y,z
  C_DBOX5 = C_DBOX0 + 5,
  C_LATE,          // voxels first observed while obstacles existed that no wave has reached yet (k_fuse marks them in latebits_)
  C_LIST0,         // lengths of the active-tile lists: round r of an update reads counter r % 3, appends to (r + 1) % 3
  C_LIST1,         //   and clears (r + 2) % 3 (consumed by round r - 1, needed empty by round r + 1): no memset between rounds.
  C_LIST2,         //   Between updates pending tiles sit in list 0 / C_LIST0 and the other two are zero.
  C_INVALIDATED,   // stats
  C_SWEEPS,
  C_WRITES,
  C_VISITS,
  C_ROUNDS,      // rounds of the compact-list mode that found a non-empty list (counted by the kernel)
  C_SCRATCH,
  C_REMOTE_DEL,  // sharded maps: some shard reported an occupied->free transition since the last UpdateESDF
  C_FT_OVF0,     // bulk path: column groups that went through spill mode in pass A ([0]) and pass B ([3]); the others unused
  C_FT_OVF5 = C_FT_OVF0 + 5,
  C_NN_CURSOR = C_FT_OVF0 + 1,   // cell transform (nn_kernels.hpp), in the slots the envelope passes leave unused: sites handed out,
  C_NN_FAILED = C_FT_OVF0 + 2,   //   cells that got no list (non-zero: the update is served by the envelope passes instead),
  C_NN_BRUTE = C_FT_OVF0 + 0,    //   cells without a list that k_nn_close serves one by one,
  C_NN_DIRTY = C_FT_OVF0 + 3,    //   cells an incremental transform redoes,
  C_NN_ENTRIES = C_FT_OVF0 + 4,  //   list entries in total
  C_FUSE_TICKET = C_FT_OVF0 + 5,  // k_fuse: work-groups that have finished (the last one reports and clears it: zero between launches)
  C_FT_MAXD2,    // bulk path: largest d^2 written (2^30: a voxel found no obstacle in its region)
  C_PROF0,  // 8 profiling slots (FIESTA_HIP_PROF=1): cycles in stage / propagate / write-back, queue items, ...
  C_COUNT = C_PROF0 + 8
};

struct LevelEngine;  // level_kernels.hpp

struct Snapshot {
  DevBuf<vox_t> coc;
  DevBuf<double> logodds;
  DevBuf<unsigned long long> cnt;
  DevBuf<uint32_t> occbits, gocc, obsbits, latebits;
  DevBuf<uint32_t> touched, ins, del;
  unsigned long long counters[C_COUNT];
  Geom g;
  bool stale_inf = false;
  bool win_dirty = false;
  bool valid = false;
};

class DenseMap {
 public:
  explicit DenseMap(const fiesta_hip_config &cfg);
  ~DenseMap();

  const Geom &geom() const { return g_; }
  int64_t total() const { return g_.n; }

  void set_prob_params(double p_hit, double p_miss, double p_min, double p_max, double p_occ);
  void set_update_range(const double *mn, const double *mx, bool new_vec);
  void set_original_range();

  void observe_vox(const int32_t *vox, const int32_t *occ, int64_t n, int32_t *ret, bool dev);
  void observe_pos(const double *pos, const int32_t *occ, int64_t n, int32_t *ret);
  void observe_box(const int32_t *lo, const int32_t *hi, int occ);

  bool check_update();
  bool update_occupancy(bool global_map, int64_t *n_ins, int64_t *n_del);
  void update_esdf(fiesta_hip_stats *st, bool seed_only = false);
  uint32_t path_notes() const { return notes_; }  // FIESTA_HIP_NOTE_* of the last update_esdf
  // the bulk transform, step by step, for the sharded driver (shard_group.hip): probe = queue sizes + local eligibility,
  // try = transform with a margin around the shard (false: region too large), commit = consume the queues
  void bulk_probe(unsigned long long *ni, unsigned long long *nd, long long *nocc, bool *eligible);
  bool bulk_try(fiesta_hip_stats *st, int margin, bool *exact);
  void bulk_commit(fiesta_hip_stats *st);
  int update_engine() const { return update_engine_; }
  void set_update_engine(int e) { update_engine_ = e; }
  // 2, 4, 5: a transform whenever the map's state allows one (4: the envelope passes only, 5: the cell transform first)
  bool bulk_pinned() const { return update_engine_ == 2 || update_engine_ == 4 || update_engine_ == 5; }
  int level_trace(uint32_t *out48) const;  // fiesta_hip_level_trace
  void level_tuning(int grid_groups, long long spin_limit);  // fiesta_hip_level_tuning
  void set_alone_in_group(bool alone) { alone_in_group_ = alone; }
  bool bulk_pays(double delta, double nocc, double n) const;  // is the fixed sweep cheaper than the frontier rounds?
  static bool bulk_pays_model(double delta, double nocc, double n, double ft_last_ms, double bulk_ratio);
  double ft_last_ms() const { return ft_last_ms_; }
  double bulk_ratio() const { return bulk_ratio_; }
  // continue relaxing tiles that are already flagged (after ghost entries were applied)
  void relax_pending(fiesta_hip_stats *st, int64_t *pending);

  void get_distance_vox(const int32_t *vox, int64_t n, double *out);
  void get_distance_pos(const double *pos, int64_t n, double *out);
  void get_dist_grad(const double *pos, int64_t n, double *dist, double *grad, bool dev);
  void get_occupancy_vox(const int32_t *vox, int64_t n, int32_t *out);
  void get_occupancy_pos(const double *pos, int64_t n, int32_t *out);
  int64_t host_brick_fetches() const;  // bricks fetched for scalar queries so far (tests, bench)

  void download_field(int32_t *d2, int32_t *coc, uint8_t *occ, double *logodds);
  void download_counts(int32_t *num_hit, int32_t *num_miss);
  int64_t occupied_voxels(int32_t *vox, int64_t cap);  // returns the total count (may exceed cap)
  int64_t count_no_obstacle();
  void slice_distances(int z_vox, double *out);        // nx * ny doubles, x-major
  // GetPointCloud / GetSliceMarker as arrays; both return the total count (may exceed cap), order unspecified
  void checkpoint(const char *path, bool write);  // raw dump / load of the whole state (checkpoint.hpp)
  int64_t point_cloud(int vis_lower_bound, int vis_upper_bound, float *xyz, int64_t cap);
  int64_t slice_marker(int slice, double max_dist, double *xyz, float *rgba, int64_t cap);
  void snapshot_save(int slot);
  void snapshot_restore(int slot);
  int64_t snapshot_count_updated(int slot);

  void halo_pack(const int32_t *lo, const int32_t *hi, uint32_t *out_dev);
  int64_t halo_apply(const int32_t *lo, const int32_t *hi, const uint32_t *in_dev);
  // sparse ghost exchange (shard_group.hip): entries {receiver's linear cell index, word}, 2 words each
  void halo_diff(const int32_t *lo, const int32_t *hi, uint32_t *shadow_dev, const int32_t *recv_lo, const int32_t *recv_dims,
                 uint32_t *entries_dev, unsigned long long *count_dev);
  void halo_apply_sparse(const uint32_t *entries_dev, int64_t n, unsigned long long *changed_dev);
  int64_t pending_tiles();
  void bulk_reserve(int margin);
  const unsigned long long *counter_dev(int which) const;
  int64_t export_transitions(uint32_t *out_dev, int64_t cap);
  void apply_transitions(const uint32_t *ent_dev, int64_t n);

  void synchronize();
  // staging helpers for the host-buffer forms of the shard calls
  uint32_t *scratch_u32(int64_t n) {
    use_device();
    stage_c_.ensure((size_t)n * sizeof(uint32_t), stream_);
    return (uint32_t *)stage_c_.p;
  }
  void copy_to_host(void *dst, const void *src_dev, size_t bytes) {
    use_device();
    FIESTA_HIP_CHECK(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, stream_));
    FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
  }
  void copy_to_device(void *dst_dev, const void *src, size_t bytes) {
    use_device();
    FIESTA_HIP_CHECK(hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, stream_));
    FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
  }

  // raycast front end (raycast.hip)
  void raycast_frame(const float *points, int64_t n, const double *T, const double *origin,
                     const fiesta_hip_raycast_params *p, bool dev);
  void raycast_depth(const uint16_t *depth, int rows, int cols, double fx, double fy, double cx, double cy,
                     const double *T, const double *origin, const fiesta_hip_raycast_params *p,
                     const fiesta_hip_depth_filter *f = nullptr);

  // Fiesta::DepthConversion alone (tests, inspection): the frame's points to the host, rejected pixels as NaN
  int64_t depth_conversion(const uint16_t *depth, int rows, int cols, double fx, double fy, double cx, double cy,
                           const fiesta_hip_depth_filter *f, float *points_out);
  hipStream_t stream() const { return stream_; }
  int device() const { return device_; }

 private:
  void use_device() const;
  unsigned long long read_counter(int which);
  void zero_counter(int which);
  void zero_counters(int first, int n);
  void ensure_touched_capacity(int64_t extra);
  void run_rounds(fiesta_hip_stats *st, uint32_t first_count, int first_list);
  bool run_levels(fiesta_hip_stats *st, unsigned long long ni, unsigned long long nd);  // false: the rounds have to finish
  bool bulk_eligible(unsigned long long ni, unsigned long long nd);
  bool run_bulk(fiesta_hip_stats *st, int margin, bool *exact);
  // the masked transform (mask_kernels.hpp): large deltas on partially observed maps
  bool masked_eligible(unsigned long long ni, unsigned long long nd);
  bool run_masked(fiesta_hip_stats *st, std::chrono::steady_clock::time_point h0);  // false: nothing committed, the rounds serve the update
  bool cells_wanted();                  // should this update try the cell transform (nn_kernels.hpp) before the envelope passes?
  // false: not applicable to this map (nothing launched).  incremental: only the cells whose search window holds a voxel of the
  // insert / delete queues get a new list and a new fill (the lists of the last transform must still be valid: nn_valid_)
  bool run_cells(fiesta_hip_stats *st, int margin, bool publish, bool incremental = false, unsigned long long ni = 0, unsigned long long nd = 0);
  void bulk_finish(fiesta_hip_stats *st, std::chrono::steady_clock::time_point h0, bool cells = false, bool published = false);
  void reset_stats_counters(bool lists = false, bool queues = false);
  void enable_distance_tracking();
  void collect_stats(fiesta_hip_stats *st);

  Geom g_;
  ProbParams pp_;
  int device_ = 0;
  hipStream_t stream_ = nullptr;
  hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
  std::vector<hipEvent_t> evpool_;  // per-launch timing of the relaxation kernel
  hipEvent_t pool_event(size_t i);

  // per-voxel state
  vox_t *coc_ = nullptr;               // 4 B/voxel, hot
  double *logodds_ = nullptr;          // 8 B/voxel, cold (occupancy_buffer_)
  unsigned long long *cnt_ = nullptr;  // 8 B/voxel, cold: hits<<32 | observations (num_hit_/num_miss_)
  uint32_t *occbits_ = nullptr;        // 1 bit/voxel: Exist(idx)
  uint32_t *rbits_ = nullptr;          // 1 bit/voxel: voxel joined the frontier during the current update
  uint32_t *obsbits_ = nullptr;        // 1 bit/voxel: observed at least once (k_fuse; rebuilt from the field after a restore / load)
  uint32_t *latebits_ = nullptr;       // 1 bit/voxel: first observed while obstacles existed and not reached by a wave since (C_LATE counts them)
  uint32_t *gocc_ = nullptr;           // sharded maps: 1 bit/voxel of the GLOBAL grid, replicated on every shard
  int64_t ngoccwords_ = 0;
  int64_t nbitwords_ = 0;

  // tiles
  static constexpr int tx_ = 16, ty_ = 16;  // tile extent in x,y (z extent is always 32)
  int ntx_ = 0, nty_ = 0, ntz_ = 0, ntiles_ = 0;
  uint32_t *tile_epoch_ = nullptr;
  // UpdateESDF engine (fiesta_hip_config.update_engine): 0 = choose per update, 1 = frontier rounds only,
  // 2 = bulk feature transform whenever the map state allows it, 3 = level engine for every update it can hold,
  // 4 = as 2 with the envelope passes only, 5 = as 2 with the cell transform wherever it applies, 6 = as 0, and on partially
  // observed maps the masked transform for every update the map's history allows
  int update_engine_ = 0;
  LevelEngine *lv_ = nullptr;   // the level engine's lists and control block (level_kernels.hpp), created on first use
  hipEvent_t lv_done_ = nullptr;
  double bulk_ratio_ = -1;  // >= 0 (a tuning build's FIESTA_HIP_BULK_RATIO): bulk when inserts + deletes exceed this fraction of the occupied voxels
  // Late observations: a voxel first observed while obstacles exist stays at "no obstacle" until a wave reaches it
  // (the reference never queues it), so the field is no longer the transform of the occupied set and the bulk path is
  // off until a scan finds no such voxel left (k_count_stale) or the map holds no obstacle.
  bool stale_inf_ = false;
  bool win_dirty_ = false;  // an update ran under a partial window while obstacles existed (see bulk_eligible)
  // masked transform (mask_kernels.hpp): the sites' bitmap, the side buffer that becomes the field, the marks of the voxels under
  // repair, the walk list, the quads under repair with their stamps, per-cell summaries of obsbits_, its counters (device +
  // pinned host copy)
  DevBuf<uint32_t> effocc_, mask_out_, mask_ubits_, mask_uq_, mask_qstamp_, cellnb_;
  DevBuf<unsigned long long> cellst_;  // per cell: the 27 cellobs around it, two bits each (the walks' register view)
  DevBuf<unsigned long long> mask_walks_, mask_ptab_;  // (idx, winner) pairs; the hidden sites' portals (site word, direction mask)
  DevBuf<uint8_t> cellobs_, celldist_;
  unsigned long long *mask_ctr_ = nullptr, *h_mask_ctr_ = nullptr;
  uint32_t mask_serial_ = 0;   // tags of the repair iterations (stamps are never cleared)
  uint32_t notes_ = 0;  // FIESTA_HIP_NOTE_* of the current / last update_esdf (fiesta_hip_stats.path_notes)
  long long mask_obs_count_ = -1;  // observed voxels when the cells' summaries (cellobs_ ... cellst_) were built; -1: rebuild
  size_t mask_seg_cap_ = 0;    // entries per segment of the walk list (doubles when a scene needs more)
  int mask_chain_hint_ = 10;   // repair iterations launched before the first read-back (the last update's count + 2)
  // while a masked transform runs: what the transforms read and write instead of occbits_ / coc_
  const uint32_t *tr_occ_ = nullptr;
  vox_t *tr_out_ = nullptr;
  const uint8_t *tr_cellobs_ = nullptr;
  int ft_s0_ = 16;            // ring entries per lane in LDS (16 or 32; FIESTA_HIP_FT_S0: an experiment's knob)
  static constexpr int kFtBlocks = 1024;  // work-groups of a pass (4 waves each): what 256 CUs hold at once with 16-entry rings
  double ft_last_ms_ = 0;      // kernel time of the last bulk update (the engine choice's idea of this scene's sweep)
  bool ft_in_place_ = true;   // the last transform wrote the field itself (no side buffer: its result could not be inexact)
  bool alone_in_group_ = true;  // a shard: the only one of its group (ShardGroup tells)
  bool queues_zeroed_ = false;      // this update's reset already cleared C_INSERT / C_DELETE (bulk_finish need not)
  bool ft_counters_clean_ = false;  // reset_stats_counters() ran and no transform has used the spill counters since
  DevBuf<uint32_t> ft_inter_, ft_out_;
  // cell transform (nn_kernels.hpp): first site per cell, the sites, one record (list) per cell
  DevBuf<uint32_t> nn_ctab_, nn_sites_, nn_lists_, nn_dirty_flag_, nn_dirty_list_, nn_fail_list_;
  long long nn_last_brute_ = 0;  // cells the last cell transform served by brute force (sizes the closing launch of the next)
  bool nn_valid_ = false;        // the lists describe the occupancy as of the last UpdateESDF and the field is their transform: the
                                 // next update may be incremental (cleared by every other engine, restore and load)
  double nn_last_ms_ = 0;        // kernel time of the last cell transform that succeeded ...
  long long nn_last_nocc_ = -1;  // ... and the obstacle count it ran on
  bool nn_clean_ = false;        // the last update was a cell transform whose fill cleaned the counters up behind itself (nn_fill_done)
  unsigned long long nn_tag_ = 1ull << 40;  // serial number of the cell transforms that report for themselves
  bool tried_cells_ = false;     // bulk_try: the transform waiting for bulk_commit is the cell transform's
  int nn_fail_streak_ = 0, nn_skip_ = 0;  // failed attempts in a row; eligible updates still to be left to the envelope passes
  DevBuf<unsigned long long> ft_spill_;  // backing store of the transform's rings (run_bulk)
  DevBuf<uint16_t> ft_rowlist_;
  DevBuf<int32_t> ft_rowcnt_;
  hipEvent_t ft_ev_[4] = {nullptr, nullptr, nullptr, nullptr};
  uint32_t *cbits_[2] = {nullptr, nullptr};     // 1 bit/voxel: changed in the round of that parity
  uint32_t *cstamp_[2] = {nullptr, nullptr};    // per tile: serial of the round that wrote cbits_[parity]
  int prof_ = 0;
  int spatial_blocks_ = 1024;  // work-groups of the spatial walk (multiple of 8: one stream of tiles per XCD)
  bool track_ = false;  // C_MAXD2 is maintained (enable_distance_tracking)
  int bound_scan_ = 1;  // bound the delete scan by the delete queue's box + the largest stored distance (FIESTA_HIP_BOUND_SCAN=0: whole grid)
  static constexpr uint32_t kCountOnDevice = 0xFFFFFFFFu;  // run_rounds: the first list's length was never read
  int chain_hint_ = 4;             // rounds per chain of the next small update (the previous one's count + 1)
  hipEvent_t last_chain_event_ = nullptr;  // recorded after the last round of the last chain
  bool h_counters_fresh_ = false;  // h_counters_ holds the device counters as of the end of the last chain
  int small_update_ = 4096;  // updates with at most this many inserts + deletes skip that read (FIESTA_HIP_SMALL_UPDATE)
  int list_threshold_ = 1024;  // updates that start with fewer active tiles use the compact list + paired rounds
  int spatial_ = 1;  // walk the tiles in XCD-chunked spatial order (FIESTA_HIP_SPATIAL=0: compact list order)
  uint32_t serial_ = 0;                         // relaxation rounds launched so far (all updates)
  uint32_t *tile_flag_[2] = {nullptr, nullptr};
  uint32_t *tile_list_[2] = {nullptr, nullptr};
  uint32_t epoch_ = 0;

  // queues
  DevBuf<uint32_t> touched_, ins_, del_;
  int64_t touched_upper_ = 0;  // host-side upper bound of C_TOUCHED
  unsigned long long host_counts_[4] = {0, 0, 0, 0};  // C_INSERT, C_DELETE, C_OBSERVED, C_NOCC as last read
  bool host_counts_valid_ = false;
  unsigned long long *counters_ = nullptr;
  unsigned long long *h_counters_ = nullptr;  // pinned

  // staging
  DevBuf<unsigned char> stage_a_, stage_b_, stage_c_;
  // raycast front-end state (per-frame stamp arrays = Fiesta::set_occ_/set_free_, include/Fiesta.h:107-110;
  // per-ray traversal lists), lazily allocated by raycast.hip
  struct RaycastState;
  RaycastState *rc_ = nullptr;
  void free_raycast_state();
  friend struct RaycastAccess;

  Snapshot snaps_[4];

  // scalar queries (n <= kHostQueries positions per call, host pointers): a host-side cache of 16^3-voxel bricks of the
  // field, see dense_map.hip (HostBricks).  field_epoch_ is bumped by everything that may change the field.
  static constexpr int64_t kHostQueries = 8;
  struct HostBricks;
  struct HostWords;
  HostBricks *bricks_ = nullptr;
  uint64_t field_epoch_ = 1;
  const uint32_t *host_brick(int x, int y, int z);
  int host_occ(int x, int y, int z);
};

}  // namespace fiesta
