/* fiesta_hip.h -- the drop-in boundary of the MI355X-native incremental ESDF engine.
 *
 * A plain C ABI (no C++ types, no HIP types, no torch types) exported by fiesta_amd/libfiesta_hip.so.
 * The reference (HKUST-Aerial-Robotics/FIESTA) has no FFI/plugin layer: its operator API for this path IS
 * the public section of `class fiesta::ESDFMap` (include/ESDFMap.h:111-166) plus the free function
 * `Raycast` (include/raycast.h:16-18) and the per-frame driver `Fiesta::RaycastProcess`
 * (include/Fiesta.h:194-278). Every entry point below names the reference interface it replaces; the C++
 * facade include/fiesta/ESDFMap.h re-creates the reference's class on top of these calls, so a maintainer
 * swaps the implementation file, not the callers.
 *
 * Conventions
 *   - every function returns an int status: 0 = FIESTA_HIP_OK, otherwise an error code; the message of
 *     the last failure on the calling thread is fiesta_hip_last_error(). Nothing throws across the ABI.
 *   - "vox" arrays are n x 3 int32 (x,y,z voxel coordinates), "pos" arrays are n x 3 double (metres);
 *     linear voxel index = x*Ny*Nz + y*Nz + z (src/ESDFMap.cpp:91), z fastest.
 *   - pointers are HOST pointers unless the function name ends in _dev (device pointers on the map's GPU).
 *   - one host thread drives one map (the reference is single-threaded by contract). All work of a map is
 *     ordered on one HIP stream; calls return after the device work they need has completed unless stated.
 *   - there is NO CPU fallback: if no gfx950 device is usable, fiesta_hip_create fails.
 */
#ifndef FIESTA_HIP_H
#define FIESTA_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FIESTA_HIP_OK 0
#define FIESTA_HIP_ERR_INVALID 1  /* bad argument / unsupported configuration */
#define FIESTA_HIP_ERR_DEVICE 2   /* a HIP runtime call failed */
#define FIESTA_HIP_ERR_NOMEM 3    /* device allocation failed */
#define FIESTA_HIP_ERR_STATE 4    /* call sequence error */

/* Reference sentinels (src/ESDFMap.cpp:181-182). */
#define FIESTA_HIP_UNDEFINED (-10000)
#define FIESTA_HIP_INFINITY 10000

#define FIESTA_HIP_MODE_ARRAY 0 /* ESDFMap(origin,res,map_size)  src/ESDFMap.cpp:171  */
#define FIESTA_HIP_MODE_HASH 1  /* ESDFMap(origin,res,reserve)   src/ESDFMap.cpp:130  (HASH_TABLE+BLOCK+BITWISE) */

typedef struct fiesta_hip_map fiesta_hip_map; /* opaque */

typedef struct fiesta_hip_config {
  int32_t mode;         /* FIESTA_HIP_MODE_* (array vs hash is a RUNTIME choice here, a macro upstream) */
  int32_t device;       /* HIP device ordinal */
  double origin[3];     /* l_cornor_ / origin_ */
  double resolution;    /* metres per voxel */
  double map_size[3];   /* array mode: grid = ceil(map_size/resolution) (src/ESDFMap.cpp:175-176) */
  int32_t reserve_size; /* hash mode: initial voxel reserve (src/ESDFMap.cpp:141-145) */
  int32_t update_engine; /* UpdateESDF engine: 0 = chosen per update (default), 1 = frontier rounds only, 2 = bulk
                            feature transform whenever the map is fully observed, 3 = level engine for
                            every update its lists can hold, 4 = as 2 with the envelope passes only, 5 = as 2
                            with the cell transform wherever it applies (2 chooses between the two), 6 = on
                            partially observed maps the masked transform for every update the map's history allows it for
                            (0, 2, 4 and 5 take it for deltas too large for the level engine), else as 0; on
                            fully observed maps the distances do not depend on it */
  /* Spatial sharding. A map may be one shard of a larger global grid: it owns the global
   * voxel box [shard_lo, shard_lo + grid) and stores closest-obstacle ids in GLOBAL coordinates. For an
   * unsharded map leave these zero. */
  int32_t shard_lo[3];
  int32_t global_grid[3];
} fiesta_hip_config;

/* Counters of one UpdateESDF; the reference only prints its counters (src/ESDFMap.cpp:277,394). */
typedef struct fiesta_hip_stats {
  int64_t inserted;      /* insert-queue length at entry */
  int64_t deleted;       /* delete-queue length at entry */
  int64_t invalidated;   /* voxels reset because their closest obstacle vanished */
  int64_t rounds;        /* level-synchronous frontier rounds */
  int64_t tile_visits;   /* tiles relaxed, summed over rounds */
  int64_t sweeps;        /* in-LDS relaxation sweeps, summed over tile visits */
  int64_t voxel_writes;  /* voxel states written back to HBM (a voxel may be written in several rounds) */
  double device_ms;      /* HIP-event time from first to last kernel of the update */
  double host_ms;        /* wall time of the call */
  double relax_ms;       /* sum of the HIP-event durations of the relaxation launches (k_relax) */
  int64_t relax_launches;
  int64_t prof[8];       /* engine profiling counters (only with FIESTA_HIP_PROF=1 in the environment) */
  int64_t bulk;          /* 1: this update ran the bulk feature transform (rounds == 0), relax_ms = its kernels */
  double ft_rows_ms, ft_plane_ms, ft_x_ms; /* bulk path: HIP-event time of k_ft_rows / pass A / pass B */
  int64_t ft_overflow[6];/* bulk path: column groups that moved ring entries to the backing store (deques deeper than
                            their LDS ring): [0] pass A, [3] pass B; the other entries are unused (0) */
  int64_t observed_voxels, occupied_voxels; /* map totals at entry (array mode): observed at least once / Exist() */
  int64_t ft_max_d2;     /* bulk path on a shard: largest squared distance written (decides whether the margin sufficed) */
  int64_t dropped_observations; /* hash mode, cumulative: observations that fell outside the window even after it moved
                                   to their batch (a single batch or frame spanning more than 1024 voxels on an axis,
                                   non-finite positions) and were ignored -- a non-zero value means lost map data */
  int64_t levels;        /* 1: this update ran the level engine from start to end (rounds = its levels, one per layer of the
                            reference's FIFO); 0 with bulk == 0: the frontier rounds ran (possibly after the level engine's
                            lists overflowed) */
  int64_t grid_levels;   /* of `rounds` with levels == 1: levels that ran on many CUs (k_level_grid) rather than one */
  int64_t cells;         /* with bulk == 1: the transform was the cell transform (nn_kernels.hpp: per-cell obstacle lists), not
                            the envelope passes */
  double nn_cells_ms, nn_lists_ms, nn_fill_ms; /* cell transform: HIP-event time of k_nn_cells / k_nn_lists / k_nn_fill */
  int64_t nn_entries;    /* cell transform: list entries over all cells */
  int64_t nn_failed;     /* cells that got no list when the cell transform was tried (> 0: the envelope passes served the
                            update instead, cells == 0).  Once a cell has failed the launch stops early: nn_failed and
                            nn_entries are then LOWER BOUNDS (how many work-groups were already running depends on scheduling) */
  int64_t nn_incremental; /* with cells == 1: only the cells whose search window held a changed voxel were redone (the lists of the
                             last transform were still valid), nn_dirty_cells of them */
  int64_t nn_dirty_cells;
  int64_t nn_brute_cells; /* with cells == 1: cells that got no list (nothing within reach, more candidates than a list holds) and were
                             served one by one by brute force instead of failing the transform (r06) */
  int64_t masked;        /* with bulk == 1: a PARTIALLY observed map -- the transform ran masked (mask_kernels.hpp): its result kept
                            on the observed voxels whose segment to their obstacle is observed, the others repaired by pulls */
  int64_t mask_uncertified, mask_iterations, mask_walks, mask_quads; /* masked: voxels under repair, repair iterations, segment
                                                                        walks, quads (8 x 8 x 32 voxels) under repair */
  double mask_certify_ms, mask_repair_ms;               /* masked: HIP-event time of k_mask_certify / of the repair launches */
  int64_t path_notes;    /* dense-array maps: WHY this update took the path it took -- FIESTA_HIP_NOTE_* bits (0: nothing stood in the
                            way of the exact transform, or there was nothing to do) */
} fiesta_hip_stats;

/* fiesta_hip_stats.path_notes: the gates and back-offs that decided an UpdateESDF's engine (each is a place where the latency of a
   call changes by a factor; `bulk`, `cells`, `masked`, `levels`, `rounds` say WHAT ran, these say why) */
#define FIESTA_HIP_NOTE_PARTLY_OBSERVED 0x0001    /* voxels never observed: the exact transforms are gated (src/ESDFMap.cpp:345,382) */
#define FIESTA_HIP_NOTE_PARTIAL_WINDOW 0x0002     /* the update window is not the whole array (SetUpdateRange) */
#define FIESTA_HIP_NOTE_WINDOW_HISTORY 0x0004     /* an earlier update ran under a partial window: the field depends on that history */
#define FIESTA_HIP_NOTE_LATE_OBSERVATION 0x0008   /* voxels first observed free while obstacles stood still wait for a wave (:246-249) */
#define FIESTA_HIP_NOTE_FIRST_WAVE_PENDING 0x0010 /* ... the same on a fully observed map: the exact transform waits for them too */
#define FIESTA_HIP_NOTE_DENSITY 0x0020            /* obstacle density outside the cell transform's range: envelope passes */
#define FIESTA_HIP_NOTE_CELLS_BACKOFF 0x0040      /* the cell transform failed or lost recently and is not retried yet */
#define FIESTA_HIP_NOTE_CELLS_FAILED 0x0080       /* the cell transform was tried, a cell got no list: the envelope passes served the update */
#define FIESTA_HIP_NOTE_INCREMENTAL_REDONE 0x0100 /* the incremental cell transform failed and the same call ran it in full */
#define FIESTA_HIP_NOTE_SMALL_DELTA 0x0200        /* too few changes for a whole-grid transform to pay: level engine / rounds */
#define FIESTA_HIP_NOTE_MASKED_GAVE_UP 0x0400     /* the masked transform gave the update back (lists or repair outgrew their buffers) */
#define FIESTA_HIP_NOTE_LEVELS_GAVE_UP 0x0800     /* the level engine handed the update on to the rounds */
#define FIESTA_HIP_NOTE_SHARDED 0x1000            /* a shard of a larger map: level engine and masked transform excluded */
#define FIESTA_HIP_NOTE_ID_WRAP 0x2000            /* an extent beyond 1024 voxels: ids wrap, masked transform and incremental lists excluded */
#define FIESTA_HIP_NOTE_ENGINE_PINNED 0x4000      /* update_engine is not "auto" */

const char *fiesta_hip_last_error(void);
int fiesta_hip_version(void);
/* Number of usable gfx950 devices (0 on a box without a GPU; never an error). */
int fiesta_hip_device_count(void);

/* ---- life cycle: ESDFMap::ESDFMap / ~ESDFMap (include/ESDFMap.h:112-121) ---- */
int fiesta_hip_create(const fiesta_hip_config *cfg, fiesta_hip_map **out);
int fiesta_hip_destroy(fiesta_hip_map *m);
/* array mode: grid_size_ and grid_total_size_ (include/ESDFMap.h:78,115); hash mode: allocated voxels. */
int fiesta_hip_grid_size(fiesta_hip_map *m, int32_t out[3]);
int fiesta_hip_grid_total_size(fiesta_hip_map *m, int64_t *out);
/* What SetOccupancy(Vector3i) returns for each voxel WITHOUT observing it (host arithmetic only): the reference's
 * callers test the value against -10000 and use it as the per-frame de-duplication key (include/Fiesta.h:221-232,
 * 253-273), so it must identify the voxel.  Array mode: Vox2Idx = x*Ny*Nz + y*Nz + z (src/ESDFMap.cpp:84-93; no range
 * check, like the reference).  Hash mode: the reference returns an allocation-order slot number; here the voxel's map
 * coordinates modulo 1024, packed (30 bits, never negative): unique among the voxels of one window position, which is
 * all a frame can observe. */
int fiesta_hip_voxel_key(fiesta_hip_map *m, const int32_t *vox, int64_t n, int32_t *out);

/* ---- hash-block map: the moving window ----
 * The hash-block map is unbounded like the reference's (src/ESDFMap.cpp:46-48: PosInMap/VoxInMap are always true); the
 * part of it that queries, observations and UpdateESDF work on is a WINDOW of 1024^3 voxels that starts centred on map
 * voxel (0,0,0) and FOLLOWS THE OBSERVATIONS: a SetOccupancy batch or ray-cast frame whose bounding box does not fit
 * the window recentres it on that box (per axis, in whole tiles of 16 x 16 x 32 voxels).  Pages that leave the window
 * are parked -- kept, listed by fiesta_hip_download_hash, still ANSWERING every query (GetDistance, GetOccupancy,
 * GetDistWithGradTrilinear go through a map-wide page table outside the window: a planner may ask about a goal far from
 * the sensor) with the field they held when the window left, but taking no part in observations or UpdateESDF -- and
 * rejoin, with their distance field rebuilt at the next UpdateESDF, when the window returns.  Inside the window the field
 * is the ESDF of the obstacles inside the window (reach of a closest-obstacle id: 512 voxels).
 *   fiesta_hip_hash_window    origin = map voxel of the window's lowest corner; moves (nullable) = moves so far
 *   fiesta_hip_hash_recentre  move the window so that `centre` is at its middle (e.g. to query around a goal pose) */
int fiesta_hip_hash_window(fiesta_hip_map *m, int32_t origin[3], int64_t *moves);
int fiesta_hip_hash_recentre(fiesta_hip_map *m, const int32_t centre[3]);

/* ---- parameters and window ---- */
/* ESDFMap::SetParameters (src/ESDFMap.cpp:218-224). */
int fiesta_hip_set_prob_params(fiesta_hip_map *m, double p_hit, double p_miss, double p_min, double p_max,
                               double p_occ);
/* ESDFMap::SetUpdateRange (src/ESDFMap.cpp:792-810) / SetOriginalRange (:812-824). */
int fiesta_hip_set_update_range(fiesta_hip_map *m, const double min_pos[3], const double max_pos[3],
                                int new_vec);
int fiesta_hip_set_original_range(fiesta_hip_map *m);
/* fiesta_hip_config.update_engine, changed on a live map (takes effect with the next UpdateESDF).  Array maps take 0-6;
 * hash-block maps take 0, 1 and 3 (2, 4, 5 and 6 behave as 0 there: the transforms need a dense array).  No reference
 * counterpart: the reference has one engine. */
int fiesta_hip_set_update_engine(fiesta_hip_map *m, int32_t engine);
/* Diagnostics of the last UpdateESDF the level engine served (fiesta_hip_stats.levels): for each of its first 48 levels
 * (= layers of the reference's FIFO, src/ESDFMap.cpp:339-392) the number of frontier entries (high 16 bits) and the time
 * the level took inside the one-work-group kernel in units of 10 ns (low 16 bits).  *n_levels = levels of that update
 * (may exceed 48).  No reference counterpart. */
int fiesta_hip_level_trace(fiesta_hip_map *m, uint32_t out[48], int32_t *n_levels);
/* Diagnostics of the level engine's wide levels (k_level_grid): `grid_groups` work-groups of one XCD take part
 * (0..32; 0: never launched -- wide frontiers go to the frontier rounds as before); a barrier among them waits `spin_limit`
 * polls (~1 us each) before the update is given up and repaired by the frontier rounds (0: gives up at its first barrier --
 * how the tests reach that path).  Negative values leave a setting as it is.  Defaults: 32, 262144. */
int fiesta_hip_level_tuning(fiesta_hip_map *m, int32_t grid_groups, int64_t spin_limit);

/* ---- occupancy ingest: ESDFMap::SetOccupancy x2 (src/ESDFMap.cpp:401-437), batched ----
 * Observations are applied as if SetOccupancy had been called once per entry; hit/total counters are
 * accumulated with atomics so the order inside a batch is irrelevant. ret (nullable, host) receives what
 * each individual call would have returned (the linear index, or FIESTA_HIP_UNDEFINED). */
int fiesta_hip_set_occupancy_vox(fiesta_hip_map *m, const int32_t *vox, const int32_t *occ, int64_t n,
                                 int32_t *ret);
int fiesta_hip_set_occupancy_pos(fiesta_hip_map *m, const double *pos, const int32_t *occ, int64_t n,
                                 int32_t *ret);
/* Same, inputs already resident in HBM; no return values, no host synchronisation. */
int fiesta_hip_set_occupancy_vox_dev(fiesta_hip_map *m, const int32_t *vox_dev, const int32_t *occ_dev,
                                     int64_t n);

/* SetOccupancy(Vector3i, occ) for every voxel of the inclusive box [lo, hi] (map voxel coordinates), device
 * side: the usual way to mark a whole region observed-free (the reference's callers loop over voxels). */
int fiesta_hip_set_occupancy_box(fiesta_hip_map *m, const int32_t lo[3], const int32_t hi[3], int32_t occ);

/* ---- ray casting: Raycast (src/raycast.cpp:56-158) + Fiesta::RaycastProcess (include/Fiesta.h:194-278) ---- */
typedef struct fiesta_hip_raycast_params {
  double min_ray_length, max_ray_length; /* parameters_.min/max_ray_length_ (src/parameters.cpp:9-10) */
  double l_cornor[3], r_cornor[3];       /* ray clipping box in metres (parameters_.l_cornor_/r_cornor_) */
  int32_t dedup;                         /* 1: per-frame de-dup of end points and free-space voxels with
                                            the reference's early ray termination made order-independent
                                            (see DESIGN.md); 0: every ray marks every voxel it crosses */
  int32_t inverse;                       /* 1: this map is the -DSIGNED_NEEDED companion inv_esdf_map_ (include/Fiesta.h:39-41,
                                            216-218, 249-251): the frame's end points are counted as FREE and the voxels
                                            the rays cross as OCCUPIED; same rays, same de-duplication.  A caller that
                                            wants the signed field keeps two maps of equal geometry, feeds every frame
                                            to both (inverse = 0 / 1), updates both, and subtracts the inverse map's
                                            distance (distance to the nearest free voxel) from the map's own */
} fiesta_hip_raycast_params;
/* One sensor frame: points are n x 3 float (sensor frame), transform the row-major 4x4 transform_,
 * origin the raycast_origin_. Equivalent to RaycastMultithread with ray_cast_num_thread_ == 0. */
int fiesta_hip_raycast_frame(fiesta_hip_map *m, const float *points, int64_t n, const double transform[16],
                             const double origin[3], const fiesta_hip_raycast_params *p);
int fiesta_hip_raycast_frame_dev(fiesta_hip_map *m, const float *points_dev, int64_t n,
                                 const double transform[16], const double origin[3],
                                 const fiesta_hip_raycast_params *p);
/* Depth-image front end (pinhole part of Fiesta::DepthConversion, include/Fiesta.h:341-351): uint16
 * millimetre depth, rows x cols, converted to sensor-frame points on the device, then ray cast. */
int fiesta_hip_raycast_depth(fiesta_hip_map *m, const uint16_t *depth, int32_t rows, int32_t cols,
                             double fx, double fy, double cx, double cy, const double transform[16],
                             const double origin[3], const fiesta_hip_raycast_params *p);
/* The same front end with the temporal depth-consistency filter of Fiesta::DepthConversion (use_depth_filter_,
 * include/Fiesta.h:352-379): a pixel casts a ray only if it lies inside the margin, its depth within [min_dist, max_dist],
 * and its re-projection into the PREVIOUS depth image agrees with the depth stored there within `tolerance`. The map
 * keeps the previous image on the device; the first image of a run (none stored yet, or reset != 0) casts nothing, as
 * upstream (image_cnt_ == 1). rel_transform = last_transform_^-1 * transform_, row-major 4x4, supplied by the caller
 * (the node has both poses). Defaults upstream: tolerance 0.1, max 10, min 0.1, margin 0 (src/parameters.cpp:38-42). */
typedef struct fiesta_hip_depth_filter {
  double tolerance, max_dist, min_dist;
  int32_t margin;
  int32_t reset;
  double rel_transform[16];
} fiesta_hip_depth_filter;
int fiesta_hip_raycast_depth_filtered(fiesta_hip_map *m, const uint16_t *depth, int32_t rows, int32_t cols, double fx,
                                      double fy, double cx, double cy, const double transform[16], const double origin[3],
                                      const fiesta_hip_raycast_params *p, const fiesta_hip_depth_filter *filter);
/* Fiesta::DepthConversion alone (array mode): the frame's point cloud to a host buffer of rows x cols x 3 floats in
 * pixel order; a pixel the filter rejects (filter nullable: none) reads NaN, NaN -- such points are skipped by the ray
 * cast like upstream's NaN points (include/Fiesta.h:202). *n_valid = points that survived. Advances the stored previous
 * image like the call above. */
int fiesta_hip_depth_conversion(fiesta_hip_map *m, const uint16_t *depth, int32_t rows, int32_t cols, double fx, double fy,
                                double cx, double cy, const fiesta_hip_depth_filter *filter, float *points_out,
                                int64_t *n_valid);
/* The free function Raycast itself, for one ray (voxel units); returns the voxel count in *n_out
 * (FIESTA_HIP_ERR_INVALID if the reference would throw: more than 1500 voxels). out is cap x 3 doubles. */
int fiesta_hip_raycast_single(const double start[3], const double end[3], const double minv[3],
                              const double maxv[3], double *out, int32_t cap, int32_t *n_out, int32_t device);

/* ---- occupancy fusion and the ESDF update ---- */
/* ESDFMap::CheckUpdate (src/ESDFMap.cpp:227-233): *out = 1 iff some voxel was observed since the last
 * UpdateOccupancy. */
int fiesta_hip_check_update(fiesta_hip_map *m, int32_t *out);
/* ESDFMap::UpdateOccupancy (src/ESDFMap.cpp:235-271). n_insert/n_delete (nullable) receive the lengths of
 * the insert and delete queues after the call, *any (nullable) the reference's bool return value. */
int fiesta_hip_update_occupancy(fiesta_hip_map *m, int32_t global_map, int64_t *n_insert, int64_t *n_delete,
                                int32_t *any);
/* ESDFMap::UpdateESDF (src/ESDFMap.cpp:273-398). stats is nullable. */
int fiesta_hip_update_esdf(fiesta_hip_map *m, fiesta_hip_stats *stats);

/* ---- queries, batched ---- */
/* ESDFMap::GetDistance(Vector3i) (src/ESDFMap.cpp:477-479); out-of-grid voxels read as +10000. */
int fiesta_hip_get_distance_vox(fiesta_hip_map *m, const int32_t *vox, int64_t n, double *out);
/* ESDFMap::GetDistance(Vector3d) (:467-475): -10000 outside the map. */
int fiesta_hip_get_distance_pos(fiesta_hip_map *m, const double *pos, int64_t n, double *out);
/* ESDFMap::GetDistWithGradTrilinear (:481-540): dist -1 outside the map; grad is n x 3. */
int fiesta_hip_get_dist_grad(fiesta_hip_map *m, const double *pos, int64_t n, double *dist, double *grad);
/* ESDFMap::GetOccupancy x2 (:452-465). */
int fiesta_hip_get_occupancy_vox(fiesta_hip_map *m, const int32_t *vox, int64_t n, int32_t *out);
int fiesta_hip_get_occupancy_pos(fiesta_hip_map *m, const double *pos, int64_t n, int32_t *out);
/* Device-resident query: pos_dev n x 3 double, dist_dev n double, grad_dev n x 3 double (nullable). */
int fiesta_hip_get_dist_grad_dev(fiesta_hip_map *m, const double *pos_dev, int64_t n, double *dist_dev,
                                 double *grad_dev);
/* The five host-pointer queries above, called with n <= 8 on an array map (what fiesta::ESDFMap::GetDistance & co. do:
 * one position per call, as the reference's callers -- planners, 10^4-10^6 calls a second, src/ESDFMap.cpp:467-540 is an
 * array read there), are answered from a host-side cache of 16^3-voxel bricks of the field: the first query into a brick
 * fetches it (one small kernel writing into pinned host memory, one synchronisation), every further one is a host read
 * with the same arithmetic, bit for bit.  UpdateOccupancy, UpdateESDF, a restore or load and the ghost exchange of a
 * shard invalidate the cache.  *fetches = bricks fetched so far (a statistic for tests and the benchmark). */
int fiesta_hip_host_cache_fetches(fiesta_hip_map *m, int64_t *fetches);

/* ---- whole-field access (tests, visualisation, checkpoints) ----
 * Dense dump in the reference's linear order; each output is nullable.
 *   d2      int32  squared voxel distance to the closest obstacle; -1 never observed; INT32_MAX observed
 *                  but no obstacle reachable. distance_buffer_ == sqrt(d2) * resolution.
 *   coc     3 x int32 closest_obstacle_ (global voxel coordinates; -10000 undefined)
 *   occ     uint8  Exist(idx)
 *   logodds double occupancy_buffer_ */
int fiesta_hip_download_field(fiesta_hip_map *m, int32_t *d2, int32_t *coc, uint8_t *occ, double *logodds);
/* Pending observation counters num_hit_ / num_miss_ (include/ESDFMap.h:89; num_miss_ counts ALL observations
 * since the last UpdateOccupancy), dense order (hash-block maps: the order of fiesta_hip_download_hash); each output
 * nullable. */
int fiesta_hip_download_counts(fiesta_hip_map *m, int32_t *num_hit, int32_t *num_miss);
/* Checkpoint: the whole map state (closest-obstacle field, log-odds, pending observation counters, occupancy bits,
 * insert/delete queues, update ranges; hash-block maps: the page pool, the directory and the window position) written
 * to / read from a raw file, streamed through a pinned buffer.  A file loads only into a map created with the same
 * mode, origin, resolution and grid (array mode: same shard); afterwards the map behaves exactly like the one that
 * was saved.  The reference has no counterpart (its state dies with the node). */
int fiesta_hip_save(fiesta_hip_map *m, const char *path);
int fiesta_hip_load(fiesta_hip_map *m, const char *path);
/* Visualisation exports, compacted / sliced on the device (reference: ESDFMap::GetPointCloud and GetSliceMarker,
 * src/ESDFMap.cpp:544-699, which fill ROS messages -- a ROS adapter wraps these two calls).
 * get_occupied_voxels: map voxel coordinates of every occupied voxel, at most `capacity` triples are written,
 * *n_out is the total (call with vox NULL / capacity 0 to size the buffer). Order is unspecified.
 * get_slice: GetDistance(Vector3i) for every (x, y) of the plane z = z_vox, nx * ny doubles, x-major. */
int fiesta_hip_get_occupied_voxels(fiesta_hip_map *m, int32_t *vox, int64_t capacity, int64_t *n_out);
/* Observed voxels (of the owned box of a shard) whose distance reads +10000, "no obstacle" (src/ESDFMap.cpp:246-249,
 * 306, 328): voxels observed late that no wave has reached yet, everything while the map holds no obstacle -- and, on a
 * grid beyond 1024 voxels per axis ONLY, voxels farther than 512 voxels from every obstacle: the reach of a stored id there
 * (the reference's closest_obstacle_ is a full Vector3i, include/ESDFMap.h:90, but it cannot hold such a grid).  A
 * deployment on a large grid watches this number: it is the count of distances truncated to "none".  Array maps only. */
int fiesta_hip_count_no_obstacle(fiesta_hip_map *m, int64_t *n_out);
int fiesta_hip_get_slice(fiesta_hip_map *m, int32_t z_vox, double *out);
/* The two getters themselves, filtered and converted on the device exactly as the reference fills its messages (array
 * and hash-block maps; the C++ class include/fiesta/ESDFMap.h fills any message type with the reference's field names):
 * get_point_cloud   ESDFMap::GetPointCloud(m, vis_lower_bound, vis_upper_bound), src/ESDFMap.cpp:544-582: voxel centres
 *                   (float xyz, like geometry_msgs::Point32) of the occupied voxels inside the update range whose z
 *                   INDEX lies in [vis_lower_bound, vis_upper_bound];
 * get_slice_marker  ESDFMap::GetSliceMarker(m, slice, id, color, max_dist), :639-699: centres (double xyz) and colours
 *                   (float rgba, the rainbow of :584-636) of the voxels of plane z = slice inside the x/y update range
 *                   that hold a defined, finite distance.
 * At most `capacity` entries are written, *n_out is the total (size the buffers with capacity 0). The reference's
 * message order (x, y, z lexicographic / allocation order) is not reproduced: a point set is unordered. */
int fiesta_hip_get_point_cloud(fiesta_hip_map *m, int32_t vis_lower_bound, int32_t vis_upper_bound, float *xyz,
                               int64_t capacity, int64_t *n_out);
int fiesta_hip_get_slice_marker(fiesta_hip_map *m, int32_t slice, double max_dist, double *xyz, float *rgba,
                                int64_t capacity, int64_t *n_out);
/* Hash mode: allocated voxels in allocation order (vox n x 3); with all outputs NULL only *n_out is set. */
int fiesta_hip_download_hash(fiesta_hip_map *m, int64_t *n_out, int32_t *vox, int32_t *d2, int32_t *coc,
                             uint8_t *occ);

/* Device-side snapshots of the complete map state (benchmark repetitions, tests). slot in [0,3].
 * Hash-block maps: slot 0 only, save + count_updated only (one copy of the state words; restore is an error). */
int fiesta_hip_snapshot_save(fiesta_hip_map *m, int32_t slot);
int fiesta_hip_snapshot_restore(fiesta_hip_map *m, int32_t slot);
/* Number of voxels whose (d2, closest obstacle) differs between snapshot `slot` and the current state,
 * counted as SURVEY.md 8d defines an "updated voxel": d2 differs, or the old closest obstacle is no longer
 * occupied. */
int fiesta_hip_snapshot_count_updated(fiesta_hip_map *m, int32_t slot, int64_t *updated);

/* ---- multi-GPU: one map = one SHARD of a larger grid; used by the sharded driver ----
 * Create the shard with cfg.global_grid = the global extents, cfg.shard_lo = the global voxel origin of the box it
 * OWNS, cfg.map_size = the owned extents in metres and cfg.origin = the GLOBAL map origin. The shard allocates a
 * 2-voxel ghost layer (the stencil radius) on every side that has a neighbour; closest-obstacle ids are global.
 * One UpdateESDF of the whole grid is, on every shard (all _dev pointers are on the shard's GPU):
 *     update_occupancy -> export_transitions -> [all-gather] -> apply_transitions          (occupancy in sync)
 *     esdf_seed -> { pack ghosts-to-be / [send,recv] / apply -> relax_pending } until no shard changed anything
 * The [..] steps are RCCL collectives issued by the host driver (fiesta_amd/sharded.py over torch.distributed). */
typedef struct fiesta_hip_shard_info {
  int32_t local_dims[3];    /* extents of the local array (owned box + ghost layers) */
  int32_t local_origin[3];  /* global voxel coordinates of local voxel (0,0,0) */
  int32_t owned_lo[3];      /* owned box in LOCAL coordinates, inclusive */
  int32_t owned_hi[3];
  int32_t global_grid[3];
} fiesta_hip_shard_info;
int fiesta_hip_shard_info_get(fiesta_hip_map *m, fiesta_hip_shard_info *out);
/* Copies the words of the inclusive LOCAL box [lo,hi] into out_dev (dense, z fastest); blocks until done. */
int fiesta_hip_halo_pack_dev(fiesta_hip_map *m, const int32_t box_lo[3], const int32_t box_hi[3], uint32_t *out_dev);
/* Overwrites the ghost cells of the inclusive LOCAL box with a neighbour's words; a cell that changed and carries
 * an obstacle becomes a frontier source and wakes its tile. *n_changed (nullable) = cells that differed. */
int fiesta_hip_halo_apply_dev(fiesta_hip_map *m, const int32_t box_lo[3], const int32_t box_hi[3],
                              const uint32_t *in_dev, int64_t *n_changed);
/* Occupancy transitions queued on this shard: TWO uint32 words per entry, x | y << 16 and z | occupied-now << 31
 * (global voxel coordinates); capacity and *n_out count ENTRIES. out_dev NULL: only the count. Does not consume the
 * queues. */
int fiesta_hip_export_transitions_dev(fiesta_hip_map *m, uint32_t *out_dev, int64_t capacity, int64_t *n_out);
/* Applies transitions (of any shard, own ones included) to this shard's replica of the global occupancy bitmap. */
int fiesta_hip_apply_transitions_dev(fiesta_hip_map *m, const uint32_t *entries_dev, int64_t n);
/* Host-buffer forms of the four calls above (for transports that are not GPU-aware, and for tests). */
int fiesta_hip_halo_pack(fiesta_hip_map *m, const int32_t box_lo[3], const int32_t box_hi[3], uint32_t *out);
int fiesta_hip_halo_apply(fiesta_hip_map *m, const int32_t box_lo[3], const int32_t box_hi[3], const uint32_t *in,
                          int64_t *n_changed);
int fiesta_hip_export_transitions(fiesta_hip_map *m, uint32_t *out, int64_t capacity, int64_t *n_out);
int fiesta_hip_apply_transitions(fiesta_hip_map *m, const uint32_t *entries, int64_t n);
/* The seeding half of UpdateESDF (insert drain + delete invalidation, src/ESDFMap.cpp:278-337): consumes the
 * queues and leaves the seeded tiles pending. */
int fiesta_hip_esdf_seed(fiesta_hip_map *m, fiesta_hip_stats *stats);
/* The relaxation half (src/ESDFMap.cpp:339-392): relaxes every pending tile to quiescence (no queues consumed).
 * *pending_tiles (nullable) = tiles that were pending at entry. */
int fiesta_hip_relax_pending(fiesta_hip_map *m, fiesta_hip_stats *stats, int64_t *pending_tiles);

/* ---- the shard protocol itself, native (fiesta_amd/csrc/shard_group.hip): C++ host code over RCCL ----
 * A group drives UpdateOccupancy / UpdateESDF of ONE map cut into `world` shards (1, 2, 4 or 8: 1x1x1, 2x1x1, 2x2x1,
 * 2x2x2; shard r owns box r of the regular cut, see fiesta_hip_shard_box).  Two set-ups:
 *   - one rank per GPU: n_local = 1, local_ranks[0] = this process's rank, rccl_id = the 128 bytes rank 0 obtained from
 *     fiesta_hip_rccl_unique_id and handed to every rank out of band (e.g. a torch.distributed / MPI broadcast);
 *   - every shard in this process (tests; N shards multiplexed on one GPU): n_local = world, rccl_id = NULL.
 * Per sweep of UpdateESDF each shard sends only the boundary cells that CHANGED since it last sent them
 * ({receiver cell index, word} entries) to its <= 26 neighbours in one ncclGroup; one small all-gather per sweep carries
 * the message sizes and the convergence test. */
typedef struct fiesta_hip_shard_group fiesta_hip_shard_group;
int fiesta_hip_rccl_unique_id(uint8_t id[128]);
int fiesta_hip_shard_box(const int32_t global_grid[3], int32_t world, int32_t rank, int32_t lo[3], int32_t size[3]);
int fiesta_hip_shard_group_create(fiesta_hip_map *const *local_shards, const int32_t *local_ranks, int32_t n_local,
                                  int32_t world, const uint8_t *rccl_id, fiesta_hip_shard_group **out);
/* A third transport for the same protocol: the caller's own messaging, through HOST buffers.  One shard per process (like
 * RCCL); the library stages what it sends and receives through host memory and calls
 *   all_gather(ctx, send, recv, bytes)      every rank contributes `bytes` bytes; recv = world x bytes, in rank order
 *   exchange(ctx, n, peers, send, send_bytes, recv, recv_bytes)   for k < n: send_bytes[k] bytes from send[k] go to rank
 *                                           peers[k], recv_bytes[k] bytes from that rank arrive in recv[k] (either may be 0;
 *                                           a pair of ranks exchanges at most one message each way per call)
 * Both are collective over the group's ranks and return 0 on success.  Slower than RCCL by the staging copies; it exists
 * so that the C++ sweep loop, sparse diff / apply and convergence test can run ACROSS PROCESSES where RCCL cannot (two ranks
 * on one GPU) -- the multi-process tests bind it to torch.distributed over gloo -- or over a fabric RCCL does not know. */
typedef struct fiesta_hip_shard_transport {
  void *ctx;
  int32_t (*all_gather)(void *ctx, const void *send, void *recv, int64_t bytes);
  int32_t (*exchange)(void *ctx, int32_t n, const int32_t *peers, const void *const *send, const int64_t *send_bytes,
                      void *const *recv, const int64_t *recv_bytes);
} fiesta_hip_shard_transport;
int fiesta_hip_shard_group_create_hosted(fiesta_hip_map *local_shard, int32_t local_rank, int32_t world,
                                         const fiesta_hip_shard_transport *transport, fiesta_hip_shard_group **out);
/* The LOCAL half of _create's checks (shard boxes against the regular cut, set-up rules, librccl loadable when
 * use_rccl) without the collective communicator set-up: ranks exchange the outcome of this first (out of band), so that
 * one rank's local failure cannot leave the others blocked inside ncclCommInitRank. */
int fiesta_hip_shard_group_precheck(fiesta_hip_map *const *local_shards, const int32_t *local_ranks, int32_t n_local,
                                    int32_t world, int32_t use_rccl);
/* What the RCCL communicator itself reports: *nranks = ncclCommCount (0: local transport, no communicator),
 * *rank = ncclCommUserRank.  For self-verifying multi-GPU runs (bench.py prints it). */
int fiesta_hip_shard_group_comm_info(fiesta_hip_shard_group *g, int32_t *nranks, int32_t *rank);
int fiesta_hip_shard_group_destroy(fiesta_hip_shard_group *g);
/* ESDFMap::UpdateOccupancy of the whole map: *n_insert / *n_delete are the queue sizes summed over all shards. */
int fiesta_hip_shard_group_update_occupancy(fiesta_hip_shard_group *g, int32_t global_map, int64_t *n_insert,
                                            int64_t *n_delete, int32_t *any);
/* ESDFMap::UpdateESDF of the whole map. stats: summed over this process's shards; *sweeps: ghost exchanges;
 * *entries_sent: boundary entries this process sent (8 bytes each). */
int fiesta_hip_shard_group_update_esdf(fiesta_hip_shard_group *g, fiesta_hip_stats *stats, int32_t *sweeps,
                                       int64_t *entries_sent);

/* Blocks until all device work of the map has finished. */
int fiesta_hip_synchronize(fiesta_hip_map *m);

#ifdef __cplusplus
}
#endif
#endif /* FIESTA_HIP_H */
