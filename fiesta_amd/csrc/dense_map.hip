// fiesta_amd/csrc/dense_map.hip -- gfx950 kernels + host driver of the dense-array incremental ESDF map.
//
// What replaces what (reference = HKUST-Aerial-Robotics/FIESTA, paths into its tree):
//   SetOccupancy x2            src/ESDFMap.cpp:401-437  -> k_observe_vox / k_observe_pos (atomic counters,
//                                                          first toucher appends to the touched list)
//   UpdateOccupancy            src/ESDFMap.cpp:235-271  -> k_fuse (one lane per touched voxel)
//   UpdateESDF                 src/ESDFMap.cpp:273-398  -> k_seed_insert, k_invalidate, k_relax rounds
//   GetDistance/Trilinear/...  src/ESDFMap.cpp:452-540  -> k_query_*
//
// UpdateESDF, restated for a GPU.  The reference is a FIFO work-list algorithm: a voxel is put on the
// queue when it is seeded (inserted obstacle, or orphaned by a delete) or when its distance improves; a
// queued voxel first PULLS the closest obstacles of its 24 stencil neighbours and, if that did not help,
// PUSHES its own closest obstacle to them.  Voxels that are never queued never change -- in particular a
// freshly observed free voxel keeps distance "infinity" until a wave passes by.  The fixed point it
// reaches is characterised by: every voxel v that was ever queued ("the frontier set R") ends with
//     d(v) <= |v - coc(n)|   and   d(n) <= |n - coc(v)|      for all observed stencil neighbours n,
// and voxels outside R keep their old state unless a member of R improves them.
// Here the grid is cut into TX x TY x 32 tiles (z, the fastest axis of the reference's layout, is the
// 32-lane axis).  A work-group stages one tile plus its 2-voxel halo (the stencil radius) in LDS,
// carries one "frontier" bit per voxel next to the 30-bit packed obstacle, and runs Jacobi sweeps in
// which voxel n evaluates candidate coc(v) only if v or n is on the frontier; an improved voxel is on
// the next sweep's frontier.  When the tile is quiescent the changed words are written back, the
// tile's members of R are recorded in a bitmap, and neighbouring tiles whose halo saw a frontier voxel
// are appended to the next round's tile list (level-synchronous rounds, one launch per round).
#include "dense_map.hpp"
#include "checkpoint.hpp"
#include "ft_kernels.hpp"
#include "nn_kernels.hpp"
#include "mask_kernels.hpp"
#include "level_kernels.hpp"
#include "relax_kernels.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace fiesta {

// =====================================================================================================
// small kernels
// =====================================================================================================
template <typename T>
__global__ void k_fill(T *p, T v, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    p[i] = v;
}

// Is the obstacle at GLOBAL voxel (cx,cy,cz) still occupied?  Unsharded: the map's own bitmap.  Sharded: the
// replicated global bitmap (kept in sync by export_transitions / apply_transitions), because a closest
// obstacle may live on any shard.
__device__ inline bool obstacle_alive(const Geom &g, const uint32_t *occbits, const uint32_t *gocc, int cx, int cy,
                                      int cz) {
  if (g.sharded) return (gocc[g.gbitword(cx, cy, cz)] >> (cz & 31)) & 1u;
  const int x = cx - g.gx0, y = cy - g.gy0, z = cz - g.gz0;
  return g.in_grid(x, y, z) && occ_test(occbits, g, x, y, z);
}

// Appends `value` to list[] for every thread of the WORK-GROUP whose `pred` holds with ONE atomicAdd on the counter (every
// thread of the group must call it, in uniform control flow): a returning atomic on one hot address costs ~10-50 ns and they
// serialise -- 780 of them (one per wave of a 50 k-voxel batch) were most of k_observe_vox's 12 us and of k_fuse's 29 us.
// s: two words of LDS per wave + one.  Returns the number appended by the group.
__device__ inline uint32_t block_append(bool pred, uint32_t value, uint32_t *list, unsigned long long *counter, uint32_t *s) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = (blockDim.x + 63) >> 6;
  const unsigned long long m = __ballot(pred);
  if (lane == 0) s[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t sum = 0;
    for (int w = 0; w < nwave; ++w) {
      const uint32_t c = s[w];
      s[w] = sum, sum += c;
    }
    s[nwave] = sum;
    s[nwave + 1] = sum ? (uint32_t)atomicAdd(counter, (unsigned long long)sum) : 0u;
  }
  __syncthreads();
  const uint32_t total = s[nwave];
  if (pred) list[s[nwave + 1] + s[wave] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = value;
  __syncthreads();  // (s is free again)
  return total;
}

// ---- SetOccupancy(Vector3i,int), PROBABILISTIC branch (src/ESDFMap.cpp:417-437) ----
// vox are map (global) voxel coordinates. No validation of occ here: the reference's Vector3i overload
// does none either.
__global__ __launch_bounds__(256) void k_observe_vox(Geom g, const int32_t *vox, const int32_t *occ, int64_t n,
                                                      unsigned long long *cnt, uint32_t *touched, unsigned long long *counters) {
  __shared__ uint32_t s_app[18];
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  bool first = false;
  int64_t idx = 0;
  if (i < n) {
    const int x = vox[3 * i] - g.gx0, y = vox[3 * i + 1] - g.gy0, z = vox[3 * i + 2] - g.gz0;
    if (g.in_grid(x, y, z) && g.in_window(x, y, z) && g.owned(x, y, z)) {  // VoxInRange (:420)
      idx = g.idx(x, y, z);
      const unsigned long long add = ((unsigned long long)(uint32_t)occ[i] << 32) | 1ull;
      const unsigned long long old = atomicAdd(&cnt[idx], add);
      first = (uint32_t)old == 0;  // num_miss_ became 1: first touch since the last fusion -> occupancy_queue_ (:426)
    }
  }
  block_append(first, (uint32_t)idx, touched, &counters[C_TOUCHED], s_app);
}

// SetOccupancy(Vector3i, occ) for EVERY voxel of a box (map coordinates, inclusive), e.g. "observe the whole
// grid free once": same effect as one call per voxel, without materialising the coordinate list.
__global__ __launch_bounds__(1024) void k_observe_box(Geom g, int bx0, int by0, int bz0, int ex, int ey, int ez, int occ,
                                                      unsigned long long *cnt, uint32_t *touched,
                                                      unsigned long long *counters) {
  // 16 waves per work-group, one z-run of 64 voxels of a box row per wave and step. Every voxel is visited exactly
  // once, so the counter update needs no atomic (kernels of one map are stream-ordered). The first-touch appends of a
  // step are aggregated per WORK-GROUP (LDS) into one global atomicAdd: a whole-grid box makes 134 M of them, and even
  // one atomic per wave on the single list counter took 20 ms.
  __shared__ uint32_t blk_count, blk_base;
  const int zchunks = (ez + 63) >> 6;
  const int64_t nitems = (int64_t)ex * ey * zchunks;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  const int64_t steps = (nitems + (int64_t)gridDim.x * nwave - 1) / ((int64_t)gridDim.x * nwave);
  for (int64_t st = 0; st < steps; ++st) {
    const int64_t item = (st * gridDim.x + blockIdx.x) * nwave + wave;
    if (threadIdx.x == 0) blk_count = 0;
    __syncthreads();
    bool first = false;
    int64_t idx = 0;
    if (item < nitems) {
      const int zc = (int)(item % zchunks);
      const int64_t row = item / zchunks;
      const int y = by0 + (int)(row % ey) - g.gy0, x = bx0 + (int)(row / ey) - g.gx0;
      const int zi = zc * 64 + lane;
      const int z = bz0 + zi - g.gz0;
      if (zi < ez && g.in_grid(x, y, z) && g.in_window(x, y, z) && g.owned(x, y, z)) {
        idx = g.idx(x, y, z);
        const unsigned long long old = cnt[idx];
        cnt[idx] = old + (((unsigned long long)(uint32_t)occ << 32) | 1ull);
        first = (uint32_t)old == 0;
      }
    }
    const unsigned long long m = __ballot(first);
    uint32_t woff = 0;
    if (lane == 0 && m) woff = atomicAdd(&blk_count, (uint32_t)__popcll(m));
    woff = __shfl(woff, 0);
    __syncthreads();
    if (threadIdx.x == 0 && blk_count) blk_base = (uint32_t)atomicAdd(&counters[C_TOUCHED], (unsigned long long)blk_count);
    __syncthreads();
    if (first) touched[blk_base + woff + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)idx;
  }
}

// ---- SetOccupancy(Vector3d,int) (src/ESDFMap.cpp:401-415) ----
__global__ void k_observe_pos(Geom g, const double *pos, const int32_t *occ, int64_t n,
                              unsigned long long *cnt, uint32_t *touched, unsigned long long *counters) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int o = occ[i];
  if (o != 0 && o != 1) return;  // "occ value error!" (:402-405)
  const double px = pos[3 * i], py = pos[3 * i + 1], pz = pos[3 * i + 2];
  if (px < g.lo[0] || py < g.lo[1] || pz < g.lo[2] || px > g.hi[0] || py > g.hi[1] || pz > g.hi[2]) return;
  const int x = (int)floor((px - g.org[0]) / g.res) - g.gx0;  // Pos2Vox (:74-77)
  const int y = (int)floor((py - g.org[1]) / g.res) - g.gy0;
  const int z = (int)floor((pz - g.org[2]) / g.res) - g.gz0;
  if (!g.in_grid(x, y, z) || !g.in_window(x, y, z) || !g.owned(x, y, z)) return;
  const int64_t idx = g.idx(x, y, z);
  const unsigned long long old = atomicAdd(&cnt[idx], ((unsigned long long)(uint32_t)o << 32) | 1ull);
  wave_append((uint32_t)old == 0, (uint32_t)idx, touched, &counters[C_TOUCHED]);
}

// ---- UpdateOccupancy (src/ESDFMap.cpp:235-271): one lane per touched voxel ----
// UpdateOccupancy for one touched voxel (src/ESDFMap.cpp:239-267); reports a transition, does not queue it.
__device__ inline void fuse_one(const Geom &g, const ProbParams &pp, int global_map, uint32_t idx,
                                     unsigned long long *cnt, double *logodds, vox_t *coc, uint32_t *occbits,
                                     uint32_t *gocc, uint32_t *obsbits, bool &to_ins, bool &to_del, bool &first_obs) {
  const int z = idx % g.nz, y = (idx / g.nz) % g.ny, x = idx / (g.nz * g.ny);
  const unsigned long long c = cnt[idx];
  cnt[idx] = 0;  // num_hit_ = num_miss_ = 0 (:245)
  const int64_t hits = (int64_t)(int32_t)(c >> 32), seen = (int64_t)(uint32_t)c;
  const double step = (hits >= seen - hits) ? pp.l_hit : pp.l_miss;  // majority vote (:243)
  double L = logodds[idx];
  const bool was = L > pp.l_occ;  // Exist (:16-22)
  if (coc[idx] == kUnobserved) {  // first observation: -10000 -> +10000 (:246-249)
    coc[idx] = kInf;
    first_obs = true;
    atomicOr(&obsbits[g.bitword(x, y, z)], 1u << (z & 31));
  }
  if ((step >= 0 && L >= pp.l_max) || (step <= 0 && L <= pp.l_min)) return;  // already clamped (:250-255)
  if (!global_map && !g.in_prev_window(x, y, z)) {  // local-map reset (:256-259): distance = infinity, the link stays
    L = 0;
    coc[idx] = stale_link(coc[idx]);
  }
  L = fmin(fmax(L + step, pp.l_min), pp.l_max);  // (:260-262)
  logodds[idx] = L;
  const bool now = L > pp.l_occ;
  const uint32_t bit = 1u << (z & 31);
  if (now && !was) {  // free -> occupied: insert_queue_ (:263-264)
    atomicOr(&occbits[g.bitword(x, y, z)], bit);
    if (g.sharded) atomicOr(&gocc[g.gbitword(x + g.gx0, y + g.gy0, z + g.gz0)], 1u << ((z + g.gz0) & 31));
    to_ins = true;
  } else if (!now && was) {  // occupied -> free: delete_queue_ (:265-266)
    atomicAnd(&occbits[g.bitword(x, y, z)], ~bit);
    if (g.sharded) atomicAnd(&gocc[g.gbitword(x + g.gx0, y + g.gy0, z + g.gz0)], ~(1u << ((z + g.gz0) & 31)));
    to_del = true;
  }
}

// ---- UpdateOccupancy (src/ESDFMap.cpp:235-271): one lane per touched voxel ----
// `result` (nullable, pinned host memory the device can write): the LAST work-group to finish copies the four counters the
// host wants after a fusion -- insert / delete queue lengths, observed and occupied voxels -- there and clears the touched
// list's counter: UpdateOccupancy is then two launches and one synchronisation (r04: + a counter-reset kernel + a copy).
__global__ __launch_bounds__(256) void k_fuse(Geom g, ProbParams pp, int global_map, const uint32_t *touched, int64_t n,
                                              unsigned long long *cnt, double *logodds, vox_t *coc, uint32_t *occbits, uint32_t *gocc,
                                              uint32_t *obsbits, uint32_t *latebits, int late_matters,
                                              uint32_t *ins, uint32_t *del, unsigned long long *counters, unsigned long long *result) {
  __shared__ uint32_t s_app[18];
  __shared__ uint32_t s_obs;
  __shared__ int s_late;
  if (n < 0) n = (int64_t)counters[C_TOUCHED];  // the host only knows an upper bound (it sized the grid with it)
  // work-groups beyond the list leave at once and take no ticket (a depth frame's upper bound is millions of voxels, its list
  // tens of thousands: 8192 idle groups queueing for the ticket cost 0.16 ms)
  const unsigned long long nwork = (unsigned long long)max((int64_t)1, min((int64_t)gridDim.x, (n + blockDim.x - 1) / blockDim.x));
  if (blockIdx.x >= nwork) return;
  if (threadIdx.x == 0) s_obs = 0, s_late = 0;
  long long nocc = 0;  // (thread 0's: inserts - deletes of the group)
  __syncthreads();
  // whole work-groups stride over the list (the appends below need every thread of the group in the same iteration)
  for (int64_t i0 = blockIdx.x * (int64_t)blockDim.x; i0 < n; i0 += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = i0 + threadIdx.x;
    bool to_ins = false, to_del = false, first_obs = false;
    uint32_t idx = 0;
    if (i < n) {
      idx = touched[i];
      fuse_one(g, pp, global_map, idx, cnt, logodds, coc, occbits, gocc, obsbits, to_ins, to_del, first_obs);
      // late observations (see C_LATE): marked when first seen while obstacles exist, healed when the voxel becomes one
      const int z = idx % g.nz, y = (idx / g.nz) % g.ny, x = idx / (g.nz * g.ny);
      const uint32_t bit = 1u << (z & 31);
      if (first_obs && late_matters && !to_ins) {
        atomicOr(&latebits[g.bitword(x, y, z)], bit);
        atomicAdd(&s_late, 1);
      } else if (to_ins && !first_obs && (latebits[g.bitword(x, y, z)] & bit)) {
        atomicAnd(&latebits[g.bitword(x, y, z)], ~bit);
        atomicAdd(&s_late, -1);
      }
    }
    // queue appends: ONE atomic per work-group, pass and queue (block_append)
    const uint32_t ni = block_append(to_ins, idx, ins, &counters[C_INSERT], s_app);
    const uint32_t nd = block_append(to_del, idx, del, &counters[C_DELETE], s_app);
    nocc += (long long)ni - (long long)nd;
    // bookkeeping for the choice of the UpdateESDF engine: observed voxels and occupied voxels of the map
    const unsigned long long mo = __ballot(first_obs);
    if ((threadIdx.x & 63) == 0 && mo) atomicAdd(&s_obs, (uint32_t)__popcll(mo));
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_obs) atomicAdd(&counters[C_OBSERVED], (unsigned long long)s_obs);
    if (nocc) atomicAdd(&counters[C_NOCC], (unsigned long long)nocc);
    if (s_late) atomicAdd(&counters[C_LATE], (unsigned long long)(long long)s_late);
    if (result) {
      __threadfence();
      if (atomicAdd(&counters[C_FUSE_TICKET], 1ull) == nwork - 1ull) {  // everybody else is done
        __threadfence();
        for (int k = 0; k < 4; ++k) result[k] = atomicAdd(&counters[C_INSERT + k], 0ull);
        result[C_LATE - C_INSERT] = atomicAdd(&counters[C_LATE], 0ull);
        counters[C_TOUCHED] = 0;
        counters[C_FUSE_TICKET] = 0;
        __threadfence_system();
      }
    }
  }
}

// =====================================================================================================
// UpdateESDF
// =====================================================================================================
// Insert drain (src/ESDFMap.cpp:278-291): a queued voxel that is still occupied becomes its own closest
// obstacle at distance 0 and joins the frontier (ACT tag, consumed by the first relaxation round).
__global__ void k_seed_insert(Geom g, TileGrid tg, const uint32_t *ins, int64_t n, vox_t *coc,
                              const uint32_t *occbits, uint32_t *flag, uint32_t *list, unsigned long long *count) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t idx = ins[i];
  const int z = idx % g.nz, y = (idx / g.nz) % g.ny, x = idx / (g.nz * g.ny);
  if (!occ_test(occbits, g, x, y, z)) return;  // "Exist after a whole bunch of updates" (:282)
  coc[idx] = pack_coc(x + g.gx0, y + g.gy0, z + g.gz0) | kAct;
  activate_tile(tg.tile_of(x, y, z), flag, list, count);
}

// Delete drain (src/ESDFMap.cpp:292-337). The reference walks the vanished obstacle's linked list; with
// no lists here, every voxel whose closest obstacle is no longer occupied is found by one coalesced scan
// of the 4-byte state (the obstacle's occupancy bit is an L2-resident gather: neighbouring voxels point
// at the same obstacle). Such a voxel is reset to "no obstacle" and tagged as frontier seed; its re-seed
// from the neighbourhood (:308-321) is simply its first pull in k_relax.
// Largest finite d^2 currently stored (one pass over the grid when distance tracking is switched on).
__global__ void k_maxd2_scan(Geom g, const vox_t *coc, unsigned long long *counters) {
  __shared__ uint32_t blk;
  if (threadIdx.x == 0) blk = 0;
  __syncthreads();
  uint32_t mx = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < g.n; i += (int64_t)gridDim.x * blockDim.x) {
    const vox_t w = coc[i];
    if (w & kNoCoc) continue;
    const int z = (int)(i % g.nz), y = (int)((i / g.nz) % g.ny), x = (int)(i / ((int64_t)g.nz * g.ny));
    mx = max(mx, (uint32_t)dist2(g.wrap, x + g.gx0, y + g.gy0, z + g.gz0, w & ~kAct));
  }
  for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, off));
  if ((threadIdx.x & 63) == 0) atomicMax(&blk, mx);
  __syncthreads();
  if (threadIdx.x == 0 && blk) atomicMax(&counters[C_MAXD2], (unsigned long long)blk);
}

// Bounding box of the delete queue (ONE work-group: a few thousand entries at most per update in practice).
__global__ __launch_bounds__(1024) void k_del_bbox(Geom g, const uint32_t *del, int64_t n, unsigned long long *counters) {
  __shared__ int lo[3], hi[3];
  if (threadIdx.x < 3) lo[threadIdx.x] = 0x7FFFFFFF, hi[threadIdx.x] = -1;
  __syncthreads();
  int mn[3] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF}, mx[3] = {-1, -1, -1};
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
    const uint32_t idx = del[i];
    const int c[3] = {(int)(idx / ((uint32_t)g.nz * (uint32_t)g.ny)), (int)((idx / (uint32_t)g.nz) % (uint32_t)g.ny),
                      (int)(idx % (uint32_t)g.nz)};
#pragma unroll
    for (int k = 0; k < 3; ++k) mn[k] = min(mn[k], c[k]), mx[k] = max(mx[k], c[k]);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    for (int off = 32; off > 0; off >>= 1) {
      mn[k] = min(mn[k], __shfl_xor(mn[k], off));
      mx[k] = max(mx[k], __shfl_xor(mx[k], off));
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMin(&lo[k], mn[k]);
      atomicMax(&hi[k], mx[k]);
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    counters[C_DBOX0 + threadIdx.x] = (unsigned long long)(long long)lo[threadIdx.x];
    counters[C_DBOX0 + 3 + threadIdx.x] = (unsigned long long)(long long)hi[threadIdx.x];
  }
}

// A voxel OUTSIDE the update window whose obstacle vanished (local / sliding-window maps only).  In the reference such an
// orphan is re-seeded from the first stencil neighbour IN the window that holds a live obstacle (src/ESDFMap.cpp:308-321:
// dirs_ order, VoxInRange gates the neighbour, not the orphan -- and the neighbours include orphans re-seeded a moment
// earlier in the same list walk, which runs from the rim of the dead cell inwards), is then put on the update queue like
// any other orphan, PULLS the best obstacle of its in-window neighbours when it is popped (:345-366: only the neighbour is
// range-checked) -- and is never pushed into while it stays outside (:378).  So it ends with the best of what its
// in-window neighbours hold once they have been re-seeded themselves.  The frontier rounds only stage voxels of the
// window; the orphans outside keep their reset tag through the rounds and take that pull afterwards (k_reseed_outside),
// from the relaxed field.  The tag is gone after that: when the window later covers the voxel it holds what the
// reference holds, not a stale seed.
__global__ __launch_bounds__(256) void k_reseed_outside(Geom g, vox_t *coc, const uint32_t *occbits, const uint32_t *gocc,
                                                        const unsigned long long *counters, int bounded) {
  int bx0 = 0, by0 = 0, bz0 = 0, bx1 = g.nx - 1, by1 = g.ny - 1, bz1 = g.nz - 1;
  if (bounded) {  // the box k_invalidate scanned
    const int r = (int)ceil(sqrt((double)counters[C_MAXD2])) + 1;
    bx0 = max(bx0, (int)(long long)counters[C_DBOX0 + 0] - r), bx1 = min(bx1, (int)(long long)counters[C_DBOX0 + 3] + r);
    by0 = max(by0, (int)(long long)counters[C_DBOX0 + 1] - r), by1 = min(by1, (int)(long long)counters[C_DBOX0 + 4] + r);
    bz0 = max(bz0, (int)(long long)counters[C_DBOX0 + 2] - r), bz1 = min(bz1, (int)(long long)counters[C_DBOX0 + 5] + r);
    if (bx0 > bx1 || by0 > by1 || bz0 > bz1) return;
  }
  const int ez = bz1 - bz0 + 1, ey = by1 - by0 + 1;
  const int64_t nbox = (int64_t)(bx1 - bx0 + 1) * ey * ez;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nbox; i += (int64_t)gridDim.x * blockDim.x) {
    const int z = bz0 + (int)(i % ez), y = by0 + (int)((i / ez) % ey), x = bx0 + (int)(i / ((int64_t)ez * ey));
    const int64_t idx = g.idx(x, y, z);
    if (coc[idx] != kReset || g.in_window(x, y, z)) continue;
    vox_t best = kInf;
    int32_t bestd = kD2Inf;
#define FIESTA_RESEED(DX, DY, DZ)                                                                   \
  {                                                                                                 \
    const int ux = x + (DX), uy = y + (DY), uz = z + (DZ);                                          \
    if (g.in_grid(ux, uy, uz) && g.in_window(ux, uy, uz)) {                                         \
      const vox_t w = coc[g.idx(ux, uy, uz)];                                                       \
      if (!(w & kNoCoc)) {                                                                          \
        int cx, cy, cz;                                                                             \
        unpack_coc(g.wrap, ux + g.gx0, uy + g.gy0, uz + g.gz0, w & ~kAct, cx, cy, cz);                \
        const int32_t d = dist2(g.wrap, x + g.gx0, y + g.gy0, z + g.gz0, pack_coc(cx, cy, cz));     \
        if (d < bestd && (!g.wrap || d < kD2Cap) && obstacle_alive(g, occbits, gocc, cx, cy, cz)) bestd = d, best = w & ~kAct; \
      }                                                                                             \
    }                                                                                               \
  }
    FIESTA_STENCIL24(FIESTA_RESEED)
#undef FIESTA_RESEED
    coc[idx] = best;
  }
}

__device__ inline vox_t reseed_first_neighbour(const Geom &g, const vox_t *coc, const uint32_t *occbits, const uint32_t *gocc,
                                               int x, int y, int z) {
#define FIESTA_RESEED(DX, DY, DZ)                                                                   \
  {                                                                                                 \
    const int ux = x + (DX), uy = y + (DY), uz = z + (DZ);                                          \
    if (g.in_grid(ux, uy, uz) && g.in_window(ux, uy, uz)) {                                         \
      const vox_t w = coc[g.idx(ux, uy, uz)];                                                       \
      if (!(w & kNoCoc)) {                                                                          \
        int cx, cy, cz;                                                                             \
        unpack_coc(g.wrap, ux + g.gx0, uy + g.gy0, uz + g.gz0, w & ~kAct, cx, cy, cz);                \
        if (obstacle_alive(g, occbits, gocc, cx, cy, cz)) return w & ~kAct;                         \
      }                                                                                             \
    }                                                                                               \
  }
  FIESTA_STENCIL24(FIESTA_RESEED)
#undef FIESTA_RESEED
  return kInf;
}

// LEVELS (the level engine seeds itself from this scan, level_kernels.hpp): an orphan inside the window is reset with the
// dead id still in the word (kReset | id: k_level_outside wants to know whose orphan it was) and appended to level 0's
// frontier; one outside the window is only listed (lv.outside) and keeps its word until k_level_outside has judged it.
template <bool LEVELS>
__global__ __launch_bounds__(256) void k_invalidate(Geom g, TileGrid tg, vox_t *coc, const uint32_t *occbits,
                                                    const uint32_t *gocc, uint32_t *flag, uint32_t *list,
                                                    unsigned long long *count, unsigned long long *counters, int bounded,
                                                    LevelArgs lv) {
  // one wave per z-row, 16-byte loads: a lane owns 8 consecutive voxels (512 voxels = 2 KiB per wave step); its
  // obstacle-occupancy gathers are independent of each other instead of one dependent load -> gather -> store chain per
  // voxel. Row and tile arithmetic is wave-uniform 32-bit math. (nz % 4 != 0: the same loop with scalar loads.)
  // bounded: a voxel whose closest obstacle was deleted lies within max-stored-distance of that obstacle, so only the
  // delete queue's bounding box grown by that radius is scanned (a handful of deletes per depth frame no longer cost a
  // pass over the whole grid). Both bounds are read from device memory: no host round trip.
  int bx0 = 0, by0 = 0, bz0 = 0, bx1 = g.nx - 1, by1 = g.ny - 1, bz1 = g.nz - 1;
  if (bounded) {
    const int r = (int)ceil(sqrt((double)counters[C_MAXD2])) + 1;
    bx0 = max(bx0, (int)(long long)counters[C_DBOX0 + 0] - r), bx1 = min(bx1, (int)(long long)counters[C_DBOX0 + 3] + r);
    by0 = max(by0, (int)(long long)counters[C_DBOX0 + 1] - r), by1 = min(by1, (int)(long long)counters[C_DBOX0 + 4] + r);
    bz0 = max(bz0, (int)(long long)counters[C_DBOX0 + 2] - r), bz1 = min(bz1, (int)(long long)counters[C_DBOX0 + 5] + r);
    if (bx0 > bx1 || by0 > by1 || bz0 > bz1) return;
    bz0 &= ~31;  // whole tiles along z (the lane groups below map to tiles)
  }
  const uint32_t nry = (uint32_t)(by1 - by0 + 1), nrows = (uint32_t)(bx1 - bx0 + 1) * nry;
  const int lane = threadIdx.x & 63;
  const bool vec = (g.nz & 3) == 0;
  const bool win_all = g.wx0 <= 0 && g.wy0 <= 0 && g.wz0 <= 0 && g.wx1 >= g.nx - 1 && g.wy1 >= g.ny - 1 && g.wz1 >= g.nz - 1;
  unsigned long long local = 0;
  constexpr int V = 8;  // consecutive voxels per lane: two 16-byte loads in flight, and longer same-obstacle runs per lane
  // XCD-aware row order (work-group b runs on XCD b % 8; used for speed only): every XCD scans one contiguous eighth
  // of the rows, so the slice of the occupancy bitmap its gathers hit (the obstacles NEAR its voxels, ~2 MB of the
  // 16 MB at 512^3) stays in that XCD's 4 MB L2 instead of missing to the fabric.
  const uint32_t per_xcd = (nrows + 7u) / 8u, xcd = blockIdx.x & 7u;
  const uint32_t row_end = min(nrows, (xcd + 1u) * per_xcd);
  // R rows per wave and step: R * 2 KiB of loads in flight per wave (the scan needs ~16 MB in flight device-wide to
  // cover the HBM latency at full bandwidth)
  constexpr int R = 2;
  const uint32_t rstride = (gridDim.x >> 3) * 4u;
  for (uint32_t row0 = xcd * per_xcd + (blockIdx.x >> 3) * 4u + (threadIdx.x >> 6); row0 < row_end; row0 += R * rstride) {
    for (int zb = bz0; zb <= bz1; zb += 64 * V) {
      const int z8 = zb + V * lane;
      vox_t w[R][V];
      int xs[R], ys[R];
      int64_t bases[R];
      bool live[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t row = row0 + r * rstride;
        live[r] = row < row_end;
        const uint32_t rr = live[r] ? row : row0;
        xs[r] = bx0 + (int)(rr / nry), ys[r] = by0 + (int)(rr % nry);
        bases[r] = ((int64_t)xs[r] * g.ny + ys[r]) * g.nz;
#pragma unroll
        for (int k = 0; k < V; ++k) w[r][k] = kUnobserved;
        if (!live[r]) continue;
        if (vec) {
#pragma unroll
          for (int u = 0; u < V / 4; ++u)
            if (z8 + 4 * u < g.nz) {
              const uint4 q = *reinterpret_cast<const uint4 *>(coc + bases[r] + z8 + 4 * u);
              w[r][4 * u] = q.x, w[r][4 * u + 1] = q.y, w[r][4 * u + 2] = q.z, w[r][4 * u + 3] = q.w;
            }
        } else {
#pragma unroll
          for (int k = 0; k < V; ++k)
            if (z8 + k < g.nz) w[r][k] = coc[bases[r] + z8 + k];
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (!live[r]) continue;  // (wave-uniform)
        const int x = xs[r], y = ys[r];
        const int64_t base = bases[r];
        // neighbours along z mostly share their obstacle: one occupancy gather per RUN inside the lane's voxels
        uint32_t rmask = 0;
        bool dead_prev = false;
#pragma unroll
        for (int k = 0; k < V; ++k) {
          if (has_link(w[r][k])) {  // (a valid id, or the stale link of a voxel reset by the local-map rule)
            const vox_t c = w[r][k] & kIdMask;
            bool dead;
            if (k > 0 && has_link(w[r][k - 1]) && c == (w[r][k - 1] & kIdMask)) {
              dead = dead_prev;
            } else {
              int cx, cy, cz;
              unpack_coc(g.wrap, x + g.gx0, y + g.gy0, z8 + k + g.gz0, c, cx, cy, cz);
              dead = !obstacle_alive(g, occbits, gocc, cx, cy, cz);
            }
            dead_prev = dead;
            if (dead && (!g.sharded || g.owned(x, y, z8 + k))) rmask |= 1u << k;
          }
        }
        // (orphans outside the update window are reset and tagged like the others; the rounds never stage them -- they
        //  are re-seeded after the rounds, k_reseed_outside.  A shard has no such pass -- its rounds are driven from outside,
        //  relax_pending -- and settles them here, from the first in-window neighbour of the pre-delete field.)
        if (rmask && !win_all && g.sharded) {
#pragma unroll
          for (int k = 0; k < V; ++k)
            if (((rmask >> k) & 1u) && !g.in_window(x, y, z8 + k)) {
              coc[base + z8 + k] = reseed_first_neighbour(g, coc, occbits, gocc, x, y, z8 + k);
              rmask &= ~(1u << k);
              ++local;
            }
        }
        if (LEVELS) {
          if (__ballot(rmask != 0)) {
            uint32_t inside = 0, outside = 0;
#pragma unroll
            for (int k = 0; k < V; ++k) {
              const bool dead = (rmask >> k) & 1u;
              const bool inw = win_all || g.in_window(x, y, z8 + k);
              if (dead && inw) coc[base + z8 + k] = kReset | (w[r][k] & kIdMask);
              inside |= (dead && inw) ? 1u << k : 0u;
              outside |= (dead && !inw) ? 1u << k : 0u;
            }
            auto entry = [&](int k) { return lv_pack(x, y, z8 + k); };
            bool fits = lv_append_many<V>(inside, entry, lv.list[0], &lv.ctl->n[0], lv.cap);
            if (!win_all) fits &= lv_append_many<V>(outside, entry, lv.outside, &lv.ctl->nout, lv.cap);
            if (!fits) lv.ctl->overflow = 1;
            local += __popc(rmask);
          }
          continue;
        }
#pragma unroll
        for (int u = 0; u < V / 4; ++u) {
          const uint32_t nib = (rmask >> (4 * u)) & 15u;
          if (vec && nib == 15u) {
            *reinterpret_cast<uint4 *>(coc + base + z8 + 4 * u) = make_uint4(kReset, kReset, kReset, kReset);
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if ((nib >> k) & 1u) coc[base + z8 + 4 * u + k] = kReset;
          }
        }
        const unsigned long long m = __ballot(rmask != 0);
        if (m) {
          local += __popc(rmask);
          // lanes 4j .. 4j+3 cover the 32 voxels of one tile along z
          constexpr int LPT = 32 / V;
          const int grp = lane / LPT;
          if ((lane % LPT) == 0 && ((m >> (grp * LPT)) & ((1ull << LPT) - 1ull)) && z8 < g.nz) {
            const uint32_t t = tg.tile_of(x, y, z8);
            if (flag[t] == 0u) activate_tile(t, flag, list, count);
          }
        }
      }
    }
  }
  // one atomic per work-group: ~10 ns each on a single hot address, one per wave was most of this kernel's run time
  __shared__ unsigned long long blk_local;
  if (threadIdx.x == 0) blk_local = 0;
  __syncthreads();
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off);
  if (lane == 0 && local) atomicAdd(&blk_local, local);
  __syncthreads();
  if (threadIdx.x == 0 && blk_local) {
    if (LEVELS)
      atomicAdd(&lv.ctl->invalidated, (uint32_t)blk_local);  // (the level engine keeps its statistics in its control block)
    else
      atomicAdd(&counters[C_INVALIDATED], blk_local);
  }
}

// =====================================================================================================
// queries
// =====================================================================================================
// The arithmetic of the queries is written once, over any SOURCE of voxel words: the batch kernels read the field in HBM, the
// scalar host calls of the drop-in class (one position per call: fiesta::ESDFMap::GetDistance & co., include/fiesta/ESDFMap.h)
// read a host-side cache of 16^3-voxel bricks (DenseMap::HostBricks below) -- same code, same -ffp-contract=off, same bits.
__host__ __device__ inline double word_distance(const Geom &g, vox_t w, int x, int y, int z) {
  // GetDistance(Vector3i) (src/ESDFMap.cpp:477-479): unobserved (-10000) reads as +10000
  if (w & kNoCoc) return (double)FIESTA_HIP_INFINITY;
  const int32_t d2 = dist2(g.wrap, x + g.gx0, y + g.gy0, z + g.gz0, w);
  return sqrt((double)d2) * g.res;  // Dist (:122-124)
}
struct FieldWords {  // the field itself
  const Geom &g;
  const vox_t *coc;
  __device__ inline vox_t operator()(int x, int y, int z) const { return coc[g.idx(x, y, z)]; }
};
template <class Words>
__host__ __device__ inline double vox_distance(const Geom &g, Words &wd, int x, int y, int z) {
  if (!g.in_grid(x, y, z)) return (double)FIESTA_HIP_INFINITY;  // the reference reads out of bounds here
  return word_distance(g, wd(x, y, z) & ~kAct, x, y, z);
}
__host__ __device__ inline bool pos_in_map(const Geom &g, double px, double py, double pz) {  // PosInMap (:46-61)
  return !(px < g.lo[0] || py < g.lo[1] || pz < g.lo[2] || px > g.hi[0] || py > g.hi[1] || pz > g.hi[2]);
}
template <class Words>
__host__ __device__ inline double query_dist_pos(const Geom &g, Words &wd, double px, double py, double pz) {  // (:467-475)
  if (!pos_in_map(g, px, py, pz)) return (double)FIESTA_HIP_UNDEFINED;
  return vox_distance(g, wd, (int)floor((px - g.org[0]) / g.res) - g.gx0, (int)floor((py - g.org[1]) / g.res) - g.gy0,
                      (int)floor((pz - g.org[2]) / g.res) - g.gz0);
}
// GetDistWithGradTrilinear (src/ESDFMap.cpp:481-540), same operation order in f64 (compiled with -ffp-contract=off so no
// FMA contraction changes the last bit).  grad: three doubles or null.
template <class Words>
__host__ __device__ inline double query_trilinear(const Geom &g, Words &wd, const double *p, double *grad) {
  if (!pos_in_map(g, p[0], p[1], p[2])) {
    if (grad) grad[0] = grad[1] = grad[2] = 0;
    return -1;
  }
  int b[3];
  double f[3];
  for (int k = 0; k < 3; ++k) {
    const double pm = p[k] - 0.5 * g.res * 1.0;
    b[k] = (int)floor((pm - g.org[k]) / g.res);
    const double c = (b[k] + 0.5) * g.res + g.org[k];
    f[k] = (p[k] - c) * g.res_inv;
  }
  double v[2][2][2];
  for (int ix = 0; ix < 2; ++ix)
    for (int iy = 0; iy < 2; ++iy)
      for (int iz = 0; iz < 2; ++iz)
        v[ix][iy][iz] = vox_distance(g, wd, b[0] + ix - g.gx0, b[1] + iy - g.gy0, b[2] + iz - g.gz0);
  const double v00 = (1 - f[0]) * v[0][0][0] + f[0] * v[1][0][0];
  const double v01 = (1 - f[0]) * v[0][0][1] + f[0] * v[1][0][1];
  const double v10 = (1 - f[0]) * v[0][1][0] + f[0] * v[1][1][0];
  const double v11 = (1 - f[0]) * v[0][1][1] + f[0] * v[1][1][1];
  const double v0 = (1 - f[1]) * v00 + f[1] * v10;
  const double v1 = (1 - f[1]) * v01 + f[1] * v11;
  if (grad) {
    grad[2] = (v1 - v0) * g.res_inv;
    grad[1] = ((1 - f[2]) * (v10 - v00) + f[2] * (v11 - v01)) * g.res_inv;
    double gx = (1 - f[2]) * (1 - f[1]) * (v[1][0][0] - v[0][0][0]);
    gx += (1 - f[2]) * f[1] * (v[1][1][0] - v[0][1][0]);
    gx += f[2] * (1 - f[1]) * (v[1][0][1] - v[0][0][1]);
    gx += f[2] * f[1] * (v[1][1][1] - v[0][1][1]);
    grad[0] = gx * g.res_inv;
  }
  return (1 - f[2]) * v0 + f[2] * v1;
}

__global__ void k_query_dist_vox(Geom g, const vox_t *coc, const int32_t *vox, int64_t n, double *out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  FieldWords wd{g, coc};
  out[i] = vox_distance(g, wd, vox[3 * i] - g.gx0, vox[3 * i + 1] - g.gy0, vox[3 * i + 2] - g.gz0);
}
__global__ void k_query_dist_pos(Geom g, const vox_t *coc, const double *pos, int64_t n, double *out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  FieldWords wd{g, coc};
  out[i] = query_dist_pos(g, wd, pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
}
__global__ void k_query_trilinear(Geom g, const vox_t *coc, const double *pos, int64_t n, double *dist,
                                  double *grad) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double p[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
  FieldWords wd{g, coc};
  double gr[3];
  dist[i] = query_trilinear(g, wd, p, grad ? gr : nullptr);
  if (grad) grad[3 * i] = gr[0], grad[3 * i + 1] = gr[1], grad[3 * i + 2] = gr[2];
}
// GetOccupancy x2 (src/ESDFMap.cpp:452-465)
__global__ void k_query_occ_vox(Geom g, const uint32_t *occbits, const int32_t *vox, int64_t n, int32_t *out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = vox[3 * i] - g.gx0, y = vox[3 * i + 1] - g.gy0, z = vox[3 * i + 2] - g.gz0;
  out[i] = g.in_grid(x, y, z) ? (int)occ_test(occbits, g, x, y, z) : 0;
}
__global__ void k_query_occ_pos(Geom g, const uint32_t *occbits, const double *pos, int64_t n, int32_t *out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double px = pos[3 * i], py = pos[3 * i + 1], pz = pos[3 * i + 2];
  if (!pos_in_map(g, px, py, pz)) {
    out[i] = FIESTA_HIP_UNDEFINED;
    return;
  }
  const int x = (int)floor((px - g.org[0]) / g.res) - g.gx0, y = (int)floor((py - g.org[1]) / g.res) - g.gy0,
            z = (int)floor((pz - g.org[2]) / g.res) - g.gz0;
  out[i] = g.in_grid(x, y, z) ? (int)occ_test(occbits, g, x, y, z) : 0;
}

// ---- host-side brick cache of the scalar queries (DenseMap::HostBricks) ------------------------------------------------------
// The reference's GetDistance is an array read (src/ESDFMap.cpp:467-479); a planner calls it 10^4-10^6 times a second, one
// position per call.  Through copy-in / launch / copy-out / synchronise such a call cost tens of microseconds (VERDICT r4
// weak #10).  Now: the field is cached on the host in bricks of 16^3 voxels, fetched on first touch -- ONE small kernel that
// writes the brick's 4096 words and 4096 occupancy bits straight into pinned host memory, one synchronisation -- and every
// further scalar query into that brick is a host read.  Whatever can change the field bumps an epoch that invalidates
// every brick (UpdateOccupancy, UpdateESDF, restore / load, ghost exchange).
__global__ __launch_bounds__(256) void k_fetch_brick(Geom g, const vox_t *coc, const uint32_t *occbits, int bx, int by, int bz, uint32_t *dst) {
  const int t = threadIdx.x, x = 16 * bx + (t >> 4), y = 16 * by + (t & 15), z0 = 16 * bz;
  uint32_t bits = 0;
  for (int k = 0; k < 16; ++k) {
    const bool in = g.in_grid(x, y, z0 + k);
    dst[t * 16 + k] = in ? coc[g.idx(x, y, z0 + k)] : kUnobserved;
    if (in && occ_test(occbits, g, x, y, z0 + k)) bits |= 1u << k;
  }
  reinterpret_cast<uint16_t *>(dst + 4096)[t] = (uint16_t)bits;
}
struct DenseMap::HostBricks {
  static constexpr int kSlots = 2048, kWords = 4096 + 128;  // direct-mapped by a HASH of the brick's coordinates (35 MB of pinned memory)
  uint32_t *pool = nullptr;
  std::vector<int64_t> tag;
  std::vector<uint64_t> stamp;
  int64_t fetches = 0;
  ~HostBricks() {
    if (pool) (void)hipHostFree(pool);
  }
};
const uint32_t *DenseMap::host_brick(int x, int y, int z) {
  if (!bricks_) {
    bricks_ = new HostBricks;
    FIESTA_HIP_CHECK(hipHostMalloc((void **)&bricks_->pool, (size_t)HostBricks::kSlots * HostBricks::kWords * sizeof(uint32_t)));
    bricks_->tag.assign(HostBricks::kSlots, -1);
    bricks_->stamp.assign(HostBricks::kSlots, 0);
  }
  const int bx = x >> 4, by = y >> 4, bz = z >> 4;
  const int64_t id = ((int64_t)bx * ((g_.ny + 15) >> 4) + by) * ((g_.nz + 15) >> 4) + bz;
  // (id modulo 2048 put every brick of a planner's path along x on a 512^3 map -- (ny/16)(nz/16) = 1024 bricks per x-slab
  //  -- into two slots; a multiplicative hash of the three coordinates spreads any axis-aligned run over the table.  The scalar
  //  queries of one map are single-threaded, like every other call on it: include/fiesta_hip.h.)
  const uint32_t hsh = ((uint32_t)bx * 0x9E3779B1u) ^ ((uint32_t)by * 0x85EBCA77u) ^ ((uint32_t)bz * 0xC2B2AE3Du);
  const int slot = (int)((hsh ^ (hsh >> 15)) % (uint32_t)HostBricks::kSlots);
  uint32_t *b = bricks_->pool + (size_t)slot * HostBricks::kWords;
  if (bricks_->tag[slot] != id || bricks_->stamp[slot] != field_epoch_) {
    use_device();
    hipLaunchKernelGGL(k_fetch_brick, dim3(1), dim3(256), 0, stream_, g_, (const vox_t *)coc_, (const uint32_t *)occbits_, bx, by, bz, b);
    FIESTA_HIP_CHECK(hipGetLastError());
    FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
    bricks_->tag[slot] = id, bricks_->stamp[slot] = field_epoch_;
    ++bricks_->fetches;
  }
  return b;
}
struct DenseMap::HostWords {  // a source of voxel words for the query templates
  DenseMap *m;
  vox_t operator()(int x, int y, int z) { return m->host_brick(x, y, z)[((x & 15) << 8) | ((y & 15) << 4) | (z & 15)]; }
};
int64_t DenseMap::host_brick_fetches() const { return bricks_ ? bricks_->fetches : 0; }
int DenseMap::host_occ(int x, int y, int z) {
  const uint32_t *b = host_brick(x, y, z);
  const int row = ((x & 15) << 4) | (y & 15);
  return (int)((b[4096 + (row >> 1)] >> (16 * (row & 1) + (z & 15))) & 1u);
}

// ---- whole-field export ----
__global__ void k_export(Geom g, const vox_t *coc, const uint32_t *occbits, int32_t *d2, int32_t *cxyz,
                         uint8_t *occ) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < g.n; i += stride) {
    const int z = i % g.nz, y = (i / g.nz) % g.ny, x = i / ((int64_t)g.nz * g.ny);
    const vox_t w = coc[i];
    if (d2) d2[i] = (w == kUnobserved) ? -1 : ((w & kNoCoc) ? kD2Inf : dist2(g.wrap, x + g.gx0, y + g.gy0, z + g.gz0, w));
    if (cxyz) {
      int cx = FIESTA_HIP_UNDEFINED, cy = FIESTA_HIP_UNDEFINED, cz = FIESTA_HIP_UNDEFINED;
      if (!(w & kNoCoc)) unpack_coc(g.wrap, x + g.gx0, y + g.gy0, z + g.gz0, w, cx, cy, cz);
      cxyz[3 * i] = cx;
      cxyz[3 * i + 1] = cy;
      cxyz[3 * i + 2] = cz;
    }
    if (occ) occ[i] = occ_test(occbits, g, x, y, z);
  }
}

// ---- visualisation exports (reference: GetPointCloud / GetSliceMarker, src/ESDFMap.cpp:544-699) ----
// Occupied voxels, compacted on the device (wave-aggregated append); coordinates are map voxel coordinates.
__global__ void k_occupied_list(Geom g, const uint32_t *occbits, int32_t *out, unsigned long long cap,
                                unsigned long long *count) {
  const int64_t nwords = (int64_t)g.nx * g.ny * g.nzw;
  for (int64_t wi = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; wi < nwords; wi += (int64_t)gridDim.x * blockDim.x) {
    uint32_t bits = occbits[wi];
    if (!bits) continue;
    const int zw = (int)(wi % g.nzw), y = (int)((wi / g.nzw) % g.ny), x = (int)(wi / ((int64_t)g.nzw * g.ny));
    const unsigned long long base = atomicAdd(count, (unsigned long long)__popc(bits));
    unsigned long long k = base;
    while (bits) {
      const int b = __ffs(bits) - 1;
      bits &= bits - 1;
      if (k < cap) {
        out[3 * k] = x + g.gx0;
        out[3 * k + 1] = y + g.gy0;
        out[3 * k + 2] = zw * 32 + b + g.gz0;
      }
      ++k;
    }
  }
}
// One z-slice of the distance field (GetDistance(Vector3i) per cell: unobserved / no obstacle read +10000).
__global__ void k_slice(Geom g, const vox_t *coc, int z, double *out) {
  const int64_t n = (int64_t)g.nx * g.ny;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int y = (int)(i % g.ny), x = (int)(i / g.ny);
    FieldWords wd{g, coc};
    out[i] = vox_distance(g, wd, x, y, z);
  }
}

// GetPointCloud (src/ESDFMap.cpp:544-582): centres (Vox2Pos, narrowed to float like geometry_msgs::Point32) of the
// occupied voxels inside the update range whose z INDEX lies within the visualisation bounds.  Order unspecified.
__global__ void k_point_cloud(Geom g, const uint32_t *occbits, int zlo, int zhi, float *out, unsigned long long cap,
                              unsigned long long *count) {
  const int64_t nwords = (int64_t)g.nx * g.ny * g.nzw;
  for (int64_t wi = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; wi < nwords; wi += (int64_t)gridDim.x * blockDim.x) {
    uint32_t bits = occbits[wi];
    if (!bits) continue;
    const int zw = (int)(wi % g.nzw), y = (int)((wi / g.nzw) % g.ny), x = (int)(wi / ((int64_t)g.nzw * g.ny));
    if (x < g.wx0 || x > g.wx1 || y < g.wy0 || y > g.wy1) continue;
    uint32_t keep = 0;
    for (uint32_t b = bits; b; b &= b - 1) {
      const int k = __ffs(b) - 1, z = zw * 32 + k, gz = z + g.gz0;
      if (z >= g.wz0 && z <= g.wz1 && gz >= zlo && gz <= zhi) keep |= 1u << k;
    }
    if (!keep) continue;
    unsigned long long k = atomicAdd(count, (unsigned long long)__popc(keep));
    for (; keep; keep &= keep - 1, ++k) {
      if (k >= cap) continue;
      const int z = zw * 32 + __ffs(keep) - 1;
      out[3 * k] = (float)((x + g.gx0 + 0.5) * g.res + g.org[0]);
      out[3 * k + 1] = (float)((y + g.gy0 + 0.5) * g.res + g.org[1]);
      out[3 * k + 2] = (float)((z + g.gz0 + 0.5) * g.res + g.org[2]);
    }
  }
}
// GetSliceMarker (src/ESDFMap.cpp:639-699): voxels of the plane z inside the x/y update range with a defined, finite
// distance: centre (double, geometry_msgs::Point) and colour rainbow(min(d / max_dist, 1)).  Order unspecified.
__global__ void k_slice_marker(Geom g, const vox_t *coc, int z, double max_dist, double *xyz, float *rgba,
                               unsigned long long cap, unsigned long long *count) {
  const int ex = g.wx1 - g.wx0 + 1, ey = g.wy1 - g.wy0 + 1;
  const int64_t n = (int64_t)max(ex, 0) * max(ey, 0);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int y = g.wy0 + (int)(i % ey), x = g.wx0 + (int)(i / ey);
    const vox_t w = coc[g.idx(x, y, z)];
    if (w == kUnobserved || (w & kNoCoc)) continue;  // distance -10000 / +10000
    FieldWords wd{g, coc};
    const double d = vox_distance(g, wd, x, y, z);
    const unsigned long long k = atomicAdd(count, 1ull);
    if (k >= cap) continue;
    xyz[3 * k] = (x + g.gx0 + 0.5) * g.res + g.org[0];
    xyz[3 * k + 1] = (y + g.gy0 + 0.5) * g.res + g.org[1];
    xyz[3 * k + 2] = (z + g.gz0 + 0.5) * g.res + g.org[2];
    rainbow_rgba(d <= max_dist ? d / max_dist : 1, rgba + 4 * k);
  }
}

// "updated voxel" as SURVEY.md 8d defines it: d^2 differs, or the old closest obstacle vanished.
__global__ void k_count_updated(Geom g, const vox_t *before, const vox_t *now, const uint32_t *occbits,
                                const uint32_t *gocc, unsigned long long *out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long local = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < g.n; i += stride) {
    const vox_t a = before[i] == kUnobserved ? kUnobserved : (before[i] & ~kAct);
    const vox_t b = now[i] == kUnobserved ? kUnobserved : (now[i] & ~kAct);
    if (a == b) continue;
    const int z = i % g.nz, y = (i / g.nz) % g.ny, x = i / ((int64_t)g.nz * g.ny);
    const int gx = x + g.gx0, gy = y + g.gy0, gz = z + g.gz0;
    const int32_t da = (a == kUnobserved) ? -1 : ((a & kNoCoc) ? kD2Inf : dist2(g.wrap, gx, gy, gz, a));
    const int32_t db = (b == kUnobserved) ? -1 : ((b & kNoCoc) ? kD2Inf : dist2(g.wrap, gx, gy, gz, b));
    bool upd = da != db;
    if (!upd && !(a & kNoCoc)) {
      int cx, cy, cz;
      unpack_coc(g.wrap, gx, gy, gz, a, cx, cy, cz);
      upd = !obstacle_alive(g, occbits, gocc, cx, cy, cz);
    }
    local += upd && g.owned(x, y, z);
  }
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off);
  if ((threadIdx.x & 63) == 0 && local) atomicAdd(out, local);
}

// Observed voxels that carry no obstacle (bulk-path precondition, see DenseMap::stale_inf_).
__global__ void k_count_stale(Geom g, const vox_t *coc, unsigned long long *out) {
  unsigned long long local = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < g.n; i += (int64_t)gridDim.x * blockDim.x) {
    const vox_t w = coc[i];
    const int z = (int)(i % g.nz), y = (int)((i / g.nz) % g.ny), x = (int)(i / ((int64_t)g.nz * g.ny));
    local += (w != kUnobserved) && (w & kNoCoc) && g.owned(x, y, z);
  }
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off);
  if ((threadIdx.x & 63) == 0 && local) atomicAdd(out, local);
}

// =====================================================================================================
// host side
// =====================================================================================================
// one thread per element unless the caller names a cap (= the kernel strides over the grid): the default must cover the
// largest arrays (a 1024^3 shard touches 10^9 voxels at once -- a cap of 2^20 blocks silently dropped three quarters
// of them, found by tools/c5_smoke.py)
static inline int grid_for(int64_t n, int block = 256, int cap = 0x7FFFFFFF) {
  int64_t b = (n + block - 1) / block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

void DenseMap::use_device() const { FIESTA_HIP_CHECK(hipSetDevice(device_)); }

DenseMap::DenseMap(const fiesta_hip_config &cfg) {
  device_ = cfg.device;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    throw Error(FIESTA_HIP_ERR_DEVICE, "no HIP device available (this engine has no CPU fallback)");
  if (device_ < 0 || device_ >= ndev) throw Error(FIESTA_HIP_ERR_INVALID, "device ordinal out of range");
  require_gfx950(device_);
  use_device();
  if (!(cfg.resolution > 0)) throw Error(FIESTA_HIP_ERR_INVALID, "resolution must be positive");
  Geom &g = g_;
  memset(&g, 0, sizeof(g));
  g.res = cfg.resolution;
  g.res_inv = 1 / cfg.resolution;  // src/ESDFMap.cpp:173
  int gs[3];
  for (int i = 0; i < 3; ++i) {
    g.org[i] = cfg.origin[i];
    gs[i] = (int)std::ceil(cfg.map_size[i] / cfg.resolution);  // src/ESDFMap.cpp:175-176
    g.lo[i] = cfg.origin[i];
    g.hi[i] = cfg.origin[i] + cfg.map_size[i];
    if (gs[i] <= 0) throw Error(FIESTA_HIP_ERR_INVALID, "map_size must be positive");
  }
  const bool sharded = cfg.global_grid[0] > 0;
  int gg[3];
  for (int i = 0; i < 3; ++i) gg[i] = sharded ? cfg.global_grid[i] : gs[i];
  // closest-obstacle ids are coordinates modulo 1024 (common.hpp: pack_coc): plain up to 1024 voxels per axis, decoded
  // relative to the voxel ("wrap", reach 512 voxels) beyond that -- up to the 16-bit coordinates of the shard protocol
  g.wrap = 0;
  for (int i = 0; i < 3; ++i) {
    if (gg[i] > 32768) throw Error(FIESTA_HIP_ERR_INVALID, "grid extent exceeds 32768 voxels per axis");
    if (gg[i] > kMaxDim) g.wrap = 1;
  }
  // Sharded: map_size is the OWNED box, shard_lo its global voxel origin; a 2-voxel ghost layer (the stencil
  // radius) is added on every side that has a neighbour shard. origin stays the GLOBAL map origin.
  int glo[3] = {0, 0, 0}, ghi[3] = {0, 0, 0};
  if (sharded) {
    for (int i = 0; i < 3; ++i) {
      if (cfg.shard_lo[i] < 0 || cfg.shard_lo[i] + gs[i] > gg[i]) throw Error(FIESTA_HIP_ERR_INVALID, "shard box outside the global grid");
      glo[i] = cfg.shard_lo[i] > 0 ? 2 : 0;
      ghi[i] = cfg.shard_lo[i] + gs[i] < gg[i] ? 2 : 0;
      g.hi[i] = cfg.origin[i] + gg[i] * cfg.resolution;
    }
  }
  g.sharded = sharded ? 1 : 0;
  g.GX = gg[0], g.GY = gg[1], g.GZ = gg[2];
  g.GZW = (gg[2] + 31) / 32;
  g.nx = gs[0] + glo[0] + ghi[0];
  g.ny = gs[1] + glo[1] + ghi[1];
  g.nz = gs[2] + glo[2] + ghi[2];
  g.gx0 = sharded ? cfg.shard_lo[0] - glo[0] : 0;
  g.gy0 = sharded ? cfg.shard_lo[1] - glo[1] : 0;
  g.gz0 = sharded ? cfg.shard_lo[2] - glo[2] : 0;
  g.n = (int64_t)g.nx * g.ny * g.nz;
  if (g.n >= (1ll << 32)) throw Error(FIESTA_HIP_ERR_INVALID, "grid too large for 32-bit voxel indices");
  g.nzw = (g.nz + 31) / 32;
  g.ox0 = glo[0], g.oy0 = glo[1], g.oz0 = glo[2];
  g.ox1 = glo[0] + gs[0] - 1;
  g.oy1 = glo[1] + gs[1] - 1;
  g.oz1 = glo[2] + gs[2] - 1;
  nbitwords_ = (int64_t)g.nx * g.ny * g.nzw;

  if (cfg.update_engine < 0 || cfg.update_engine > 6) throw Error(FIESTA_HIP_ERR_INVALID, "unknown update_engine");
  update_engine_ = cfg.update_engine;
  ntx_ = (g.nx + tx_ - 1) / tx_;
  nty_ = (g.ny + ty_ - 1) / ty_;
  ntz_ = (g.nz + 31) / 32;
  ntiles_ = ntx_ * nty_ * ntz_;

  FIESTA_HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  FIESTA_HIP_CHECK(hipEventCreate(&ev0_));
  FIESTA_HIP_CHECK(hipEventCreate(&ev1_));
  FIESTA_HIP_CHECK(hipMalloc((void **)&coc_, g.n * sizeof(vox_t)));
  FIESTA_HIP_CHECK(hipMalloc((void **)&logodds_, g.n * sizeof(double)));
  FIESTA_HIP_CHECK(hipMalloc((void **)&cnt_, g.n * sizeof(unsigned long long)));
  FIESTA_HIP_CHECK(hipMalloc((void **)&occbits_, nbitwords_ * sizeof(uint32_t)));
  FIESTA_HIP_CHECK(hipMalloc((void **)&rbits_, nbitwords_ * sizeof(uint32_t)));
  FIESTA_HIP_CHECK(hipMalloc((void **)&obsbits_, nbitwords_ * sizeof(uint32_t)));
  FIESTA_HIP_CHECK(hipMalloc((void **)&latebits_, nbitwords_ * sizeof(uint32_t)));
  FIESTA_HIP_CHECK(hipMemsetAsync(obsbits_, 0, nbitwords_ * sizeof(uint32_t), stream_));
  FIESTA_HIP_CHECK(hipMemsetAsync(latebits_, 0, nbitwords_ * sizeof(uint32_t), stream_));
  if (sharded) {
    ngoccwords_ = (int64_t)g.GX * g.GY * g.GZW;
    FIESTA_HIP_CHECK(hipMalloc((void **)&gocc_, ngoccwords_ * sizeof(uint32_t)));
    FIESTA_HIP_CHECK(hipMemsetAsync(gocc_, 0, ngoccwords_ * sizeof(uint32_t), stream_));
  }
  FIESTA_HIP_CHECK(hipMalloc((void **)&tile_epoch_, ntiles_ * sizeof(uint32_t)));
  for (int k = 0; k < 2; ++k) {
    FIESTA_HIP_CHECK(hipMalloc((void **)&cbits_[k], nbitwords_ * sizeof(uint32_t)));
    FIESTA_HIP_CHECK(hipMalloc((void **)&cstamp_[k], ntiles_ * sizeof(uint32_t)));
    FIESTA_HIP_CHECK(hipMemsetAsync(cbits_[k], 0, nbitwords_ * sizeof(uint32_t), stream_));
    FIESTA_HIP_CHECK(hipMemsetAsync(cstamp_[k], 0, ntiles_ * sizeof(uint32_t), stream_));
  }
  for (int k = 0; k < 2; ++k) {
    FIESTA_HIP_CHECK(hipMalloc((void **)&tile_flag_[k], ntiles_ * sizeof(uint32_t)));
    FIESTA_HIP_CHECK(hipMalloc((void **)&tile_list_[k], ntiles_ * sizeof(uint32_t)));
    FIESTA_HIP_CHECK(hipMemsetAsync(tile_flag_[k], 0, ntiles_ * sizeof(uint32_t), stream_));
  }
  FIESTA_HIP_CHECK(hipMalloc((void **)&counters_, C_COUNT * sizeof(unsigned long long)));
  FIESTA_HIP_CHECK(hipHostMalloc((void **)&h_counters_, C_COUNT * sizeof(unsigned long long)));
  FIESTA_HIP_CHECK(hipMemsetAsync(counters_, 0, C_COUNT * sizeof(unsigned long long), stream_));
  hipLaunchKernelGGL(k_fill<vox_t>, dim3(grid_for(g.n, 256, 4096)), dim3(256), 0, stream_, coc_, kUnobserved, g.n);
  FIESTA_HIP_CHECK(hipGetLastError());
  FIESTA_HIP_CHECK(hipMemsetAsync(logodds_, 0, g.n * sizeof(double), stream_));
  FIESTA_HIP_CHECK(hipMemsetAsync(cnt_, 0, g.n * sizeof(unsigned long long), stream_));
  FIESTA_HIP_CHECK(hipMemsetAsync(occbits_, 0, nbitwords_ * sizeof(uint32_t), stream_));
  FIESTA_HIP_CHECK(hipMemsetAsync(rbits_, 0, nbitwords_ * sizeof(uint32_t), stream_));
  FIESTA_HIP_CHECK(hipMemsetAsync(tile_epoch_, 0, ntiles_ * sizeof(uint32_t), stream_));
  set_original_range();
  pp_ = ProbParams{0, 0, 0, 0, 0};
#ifdef FIESTA_HIP_TUNING  // a developer's build (tools/dev): the shipped library reads no environment
  if (const char *e = getenv("FIESTA_HIP_PROF")) prof_ = atoi(e);
  if (const char *e = getenv("FIESTA_HIP_SPATIAL")) spatial_ = atoi(e);
  if (const char *e = getenv("FIESTA_HIP_LIST_THRESHOLD")) list_threshold_ = std::max(0, atoi(e));
  if (const char *e = getenv("FIESTA_HIP_SMALL_UPDATE")) small_update_ = std::max(0, atoi(e));
  if (const char *e = getenv("FIESTA_HIP_BOUND_SCAN")) bound_scan_ = atoi(e);
  if (const char *e = getenv("FIESTA_HIP_BLOCKS")) spatial_blocks_ = std::max(8, atoi(e) / 8 * 8);
  if (const char *e = getenv("FIESTA_HIP_BULK_RATIO")) bulk_ratio_ = atof(e);
  if (const char *e = getenv("FIESTA_HIP_FT_S0")) ft_s0_ = atoi(e);
#endif
  if (ft_s0_ != 16 && ft_s0_ != 32) throw Error(FIESTA_HIP_ERR_INVALID, "FIESTA_HIP_FT_S0 must be 16 or 32");
  for (auto &e : ft_ev_) FIESTA_HIP_CHECK(hipEventCreate(&e));
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
}

DenseMap::~DenseMap() {
  (void)hipSetDevice(device_);
  if (stream_) (void)hipStreamSynchronize(stream_);
  free_raycast_state();
  void *ptrs[] = {coc_,          logodds_,      cnt_,          occbits_,      rbits_,     tile_epoch_,
                  tile_flag_[0], tile_flag_[1], tile_list_[0], tile_list_[1], counters_,  cbits_[0],
                  cbits_[1],     cstamp_[0],    cstamp_[1],    gocc_,         obsbits_,   latebits_,
                  mask_ctr_};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  if (h_counters_) (void)hipHostFree(h_counters_);
  if (h_mask_ctr_) (void)hipHostFree(h_mask_ctr_);
  delete bricks_;
  delete lv_;
  if (lv_done_) (void)hipEventDestroy(lv_done_);
  for (hipEvent_t e : evpool_) (void)hipEventDestroy(e);
  for (hipEvent_t e : ft_ev_)
    if (e) (void)hipEventDestroy(e);
  if (ev0_) (void)hipEventDestroy(ev0_);
  if (ev1_) (void)hipEventDestroy(ev1_);
  if (stream_) (void)hipStreamDestroy(stream_);
}

void DenseMap::set_prob_params(double p_hit, double p_miss, double p_min, double p_max, double p_occ) {
  auto logit = [](double x) { return std::log(x / (1 - x)); };  // Logit (src/ESDFMap.cpp:12-14)
  pp_.l_hit = logit(p_hit);
  pp_.l_miss = logit(p_miss);
  pp_.l_min = logit(p_min);
  pp_.l_max = logit(p_max);
  pp_.l_occ = logit(p_occ);
}

void DenseMap::set_original_range() {  // SetOriginalRange (src/ESDFMap.cpp:812-824), array flavour
  Geom &g = g_;
  g.wx0 = g.wy0 = g.wz0 = 0;
  g.wx1 = g.nx - 1;
  g.wy1 = g.ny - 1;
  g.wz1 = g.nz - 1;
  g.px0 = g.wx0, g.py0 = g.wy0, g.pz0 = g.wz0;
  g.px1 = g.wx1, g.py1 = g.wy1, g.pz1 = g.wz1;
}

void DenseMap::set_update_range(const double *mn, const double *mx, bool new_vec) {  // SetUpdateRange (:792-810)
  Geom &g = g_;
  double a[3], b[3];
  for (int i = 0; i < 3; ++i) {
    a[i] = std::max(mn[i], g.lo[i]);
    b[i] = std::min(mx[i], g.hi[i]);
  }
  if (new_vec) {
    g.px0 = g.wx0, g.py0 = g.wy0, g.pz0 = g.wz0;
    g.px1 = g.wx1, g.py1 = g.wy1, g.pz1 = g.wz1;
  }
  auto p2v = [&](double p, int i) { return (int)std::floor((p - g.org[i]) / g.res); };
  const int g0[3] = {g.gx0, g.gy0, g.gz0};
  int lo[3], hi[3];
  for (int i = 0; i < 3; ++i) {
    lo[i] = p2v(a[i], i) - g0[i];
    hi[i] = p2v(b[i] - g.res / 2, i) - g0[i];
  }
  g.wx0 = lo[0], g.wy0 = lo[1], g.wz0 = lo[2];
  g.wx1 = hi[0], g.wy1 = hi[1], g.wz1 = hi[2];
}

unsigned long long DenseMap::read_counter(int which) {
  FIESTA_HIP_CHECK(hipMemcpyAsync(&h_counters_[which], &counters_[which], sizeof(unsigned long long),
                                  hipMemcpyDeviceToHost, stream_));
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
  return h_counters_[which];
}
void DenseMap::zero_counters(int first, int n) {
  hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, stream_, &counters_[first], n);
  FIESTA_HIP_CHECK(hipGetLastError());
}
void DenseMap::zero_counter(int which) { zero_counters(which, 1); }

void DenseMap::ensure_touched_capacity(int64_t extra) {
  touched_upper_ = std::min<int64_t>(g_.n, touched_upper_ + extra);
  if ((size_t)touched_upper_ > touched_.cap) {
    // entries beyond the true count are garbage but harmless to copy
    const size_t keep = touched_.cap;
    touched_.ensure(std::min<size_t>((size_t)g_.n, std::max<size_t>((size_t)touched_upper_, 2 * touched_.cap)),
                    stream_, keep);
  }
}

void DenseMap::observe_vox(const int32_t *vox, const int32_t *occ, int64_t n, int32_t *ret, bool dev) {
  use_device();
  if (n <= 0) return;
  const Geom &g = g_;
  const int32_t *dv = vox, *docc_in = occ;
  if (!dev) {
    stage_a_.ensure(n * 3 * sizeof(int32_t), stream_);
    stage_b_.ensure(n * sizeof(int32_t), stream_);
    FIESTA_HIP_CHECK(hipMemcpyAsync(stage_a_.p, vox, n * 3 * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
    FIESTA_HIP_CHECK(hipMemcpyAsync(stage_b_.p, occ, n * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
    dv = (const int32_t *)stage_a_.p;
    docc_in = (const int32_t *)stage_b_.p;
  }
  ensure_touched_capacity(n);
  hipLaunchKernelGGL(k_observe_vox, dim3(grid_for(n, 256)), dim3(256), 0, stream_, g, dv, docc_in, n, cnt_, touched_.p,
                     counters_);
  FIESTA_HIP_CHECK(hipGetLastError());
  if (ret && !dev) {  // what each SetOccupancy(Vector3i,int) call returns: Vox2Idx(vox) (:418,421,437)
    const int gny = g.ny, gnz = g.nz;  // (shards report indices in their local array)
    for (int64_t i = 0; i < n; ++i)
      ret[i] = (vox[3 * i] - g.gx0) * (gny * gnz) + (vox[3 * i + 1] - g.gy0) * gnz + (vox[3 * i + 2] - g.gz0);
  }
  if (!dev) FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));  // staging buffers are reused
}

void DenseMap::observe_box(const int32_t *lo, const int32_t *hi, int occ) {
  use_device();
  const int64_t ex = (int64_t)hi[0] - lo[0] + 1, ey = (int64_t)hi[1] - lo[1] + 1, ez = (int64_t)hi[2] - lo[2] + 1;
  if (ex <= 0 || ey <= 0 || ez <= 0) return;
  if (ex > 4096 || ey > 4096 || ez > 4096) throw Error(FIESTA_HIP_ERR_INVALID, "box too large");
  ensure_touched_capacity(ex * ey * ez);
  hipLaunchKernelGGL(k_observe_box, dim3(grid_for(ex * ey * ((ez + 63) / 64), 16, 4096)), dim3(1024), 0, stream_, g_, lo[0], lo[1],
                     lo[2], (int)ex, (int)ey, (int)ez, occ, cnt_, touched_.p, counters_);
  FIESTA_HIP_CHECK(hipGetLastError());
}

void DenseMap::observe_pos(const double *pos, const int32_t *occ, int64_t n, int32_t *ret) {
  use_device();
  if (n <= 0) return;
  const Geom &g = g_;
  stage_a_.ensure(n * 3 * sizeof(double), stream_);
  stage_b_.ensure(n * sizeof(int32_t), stream_);
  FIESTA_HIP_CHECK(hipMemcpyAsync(stage_a_.p, pos, n * 3 * sizeof(double), hipMemcpyHostToDevice, stream_));
  FIESTA_HIP_CHECK(hipMemcpyAsync(stage_b_.p, occ, n * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
  ensure_touched_capacity(n);
  hipLaunchKernelGGL(k_observe_pos, dim3(grid_for(n)), dim3(256), 0, stream_, g, (const double *)stage_a_.p,
                     (const int32_t *)stage_b_.p, n, cnt_, touched_.p, counters_);
  FIESTA_HIP_CHECK(hipGetLastError());
  if (ret) {  // SetOccupancy(Vector3d,int) return value (:401-415)
    for (int64_t i = 0; i < n; ++i) {
      const double *p = pos + 3 * i;
      if ((occ[i] != 0 && occ[i] != 1) || p[0] < g.lo[0] || p[1] < g.lo[1] || p[2] < g.lo[2] || p[0] > g.hi[0] ||
          p[1] > g.hi[1] || p[2] > g.hi[2]) {
        ret[i] = FIESTA_HIP_UNDEFINED;
        continue;
      }
      const int x = (int)std::floor((p[0] - g.org[0]) / g.res) - g.gx0;
      const int y = (int)std::floor((p[1] - g.org[1]) / g.res) - g.gy0;
      const int z = (int)std::floor((p[2] - g.org[2]) / g.res) - g.gz0;
      ret[i] = x * (g.ny * g.nz) + y * g.nz + z;
    }
  }
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
}

bool DenseMap::check_update() {  // CheckUpdate (src/ESDFMap.cpp:227-233)
  use_device();
  if (touched_upper_ == 0) return false;
  return read_counter(C_TOUCHED) != 0;
}

bool DenseMap::update_occupancy(bool global_map, int64_t *n_ins, int64_t *n_del) {
  use_device();
  // ONE host synchronisation per call (none when nothing was observed): the host keeps the queue sizes / map totals of
  // its last read and an upper bound of the touched list, so the fusion kernel is launched without asking the device
  // first (it reads the exact length itself); only the result is read back.
  if (!host_counts_valid_) {
    FIESTA_HIP_CHECK(hipMemcpyAsync(&h_counters_[C_INSERT], &counters_[C_INSERT], 4 * sizeof(unsigned long long),
                                    hipMemcpyDeviceToHost, stream_));  // C_INSERT, C_DELETE, C_OBSERVED, C_NOCC
    FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
    for (int k = 0; k < 4; ++k) host_counts_[k] = h_counters_[C_INSERT + k];
    host_counts_valid_ = true;
  }
  unsigned long long ni = host_counts_[0], nd = host_counts_[1];
  const unsigned long long obs_before = host_counts_[2];
  const long long nocc_before = (long long)host_counts_[3];
  const unsigned long long nt = (unsigned long long)touched_upper_;
  if (nt) {
    ++field_epoch_;  // (occupancy and first observations may change: the host-side brick cache of the scalar queries is stale)
    ins_.ensure(ni + nt, stream_, ni);
    del_.ensure(nd + nt, stream_, nd);
    hipLaunchKernelGGL(k_fuse, dim3(grid_for((int64_t)nt, 256, 8192)), dim3(256), 0, stream_, g_, pp_, global_map ? 1 : 0,
                       (const uint32_t *)touched_.p, (int64_t)-1, cnt_, logodds_, coc_, occbits_, gocc_, obsbits_, latebits_,
                       (nocc_before > 0 || ni > 0 || nd > 0) ? 1 : 0, ins_.p, del_.p, counters_, &h_counters_[C_INSERT]);  // (its last work-group writes the four results into h_counters_)
    FIESTA_HIP_CHECK(hipGetLastError());
    touched_upper_ = 0;
    FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
    for (int k = 0; k < 4; ++k) host_counts_[k] = h_counters_[C_INSERT + k];
    ni = host_counts_[0];
    nd = host_counts_[1];
    // voxels observed for the first time while obstacles (or pending deletes of obstacles) exist: see stale_inf_
    if (host_counts_[2] != obs_before && (nocc_before > 0 || nd > 0)) stale_inf_ = true;
    if (!global_map) stale_inf_ = true;  // (the local-map reset may have put observed voxels back to +10000: same rule)
  }
  if (n_ins) *n_ins = (int64_t)ni;
  if (n_del) *n_del = (int64_t)nd;
  return ni != 0 || nd != 0;  // (:270)
}

// Maps that take many small updates (the ray-cast front end switches this on): keep an upper bound of the stored
// distances so that the delete drain scans (box of the deleted obstacles) + (that radius) instead of the whole grid.
// Default work-queue engine on an unsharded map only (ghost cells / remote deletes of a shard are not covered).
void DenseMap::enable_distance_tracking() {
  if (track_ || !bound_scan_ || g_.sharded) return;
  hipLaunchKernelGGL(k_maxd2_scan, dim3(grid_for(g_.n, 256, 4096)), dim3(256), 0, stream_, g_, (const vox_t *)coc_, counters_);
  FIESTA_HIP_CHECK(hipGetLastError());
  track_ = true;
}

// ONE memset: the statistics, the transform's spill counters and -- at the start of an update (lists) -- the tile-list counters
void DenseMap::reset_stats_counters(bool lists, bool queues) {
  static_assert(C_LIST2 == C_LIST0 + 2 && C_INVALIDATED == C_LIST2 + 1 && C_FT_MAXD2 < C_COUNT, "counter layout");
  const int first = lists ? C_LIST0 : C_INVALIDATED;
  if (queues) {  // ... and the two queue lengths in the same launch (a transform drains both queues whatever happens)
    static_assert(C_DELETE == C_INSERT + 1, "counter layout");
    hipLaunchKernelGGL(k_zero_words2, dim3(1), dim3(64), 0, stream_, &counters_[first], C_COUNT - first, &counters_[C_INSERT], 2);
    FIESTA_HIP_CHECK(hipGetLastError());
  } else {
    zero_counters(first, C_COUNT - first);
  }
  ft_counters_clean_ = true;
}

void DenseMap::collect_stats(fiesta_hip_stats *st) {
  if (!h_counters_fresh_) {  // (the last chain of rounds brought the counters with it and nothing ran since)
    FIESTA_HIP_CHECK(hipMemcpyAsync(h_counters_, counters_, C_COUNT * sizeof(unsigned long long),
                                    hipMemcpyDeviceToHost, stream_));
    FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
  }
  h_counters_fresh_ = false;
  if (st) {
    st->invalidated = (int64_t)h_counters_[C_INVALIDATED];
    st->sweeps = (int64_t)h_counters_[C_SWEEPS];
    st->voxel_writes = (int64_t)h_counters_[C_WRITES];
    st->tile_visits = (int64_t)h_counters_[C_VISITS];
    for (int k = 0; k < 8; ++k) st->prof[k] = (int64_t)h_counters_[C_PROF0 + k];
  }
}

hipEvent_t DenseMap::pool_event(size_t i) {
  while (evpool_.size() <= i) {
    hipEvent_t e;
    FIESTA_HIP_CHECK(hipEventCreate(&e));
    evpool_.push_back(e);
  }
  return evpool_[i];
}

void DenseMap::run_rounds(fiesta_hip_stats *st, uint32_t first_count, int first_list) {
  int cur = first_list;
  uint32_t ncur = first_count;
  int64_t launches = 0, device_rounds = -1, spatial_rounds = 0;  // rounds that found work are counted on the device
  size_t nev = 0;  // event pairs: one per spatial round, one per chain
  bool lists_dirty = false;
  TileGrid tg{tx_, ty_, ntx_, nty_, ntz_};
  serial_ += 2;  // no stamp of an earlier update may validate this update's first round
  // one round of the work-queue engine: active tiles of list/flags `cur_list` -> `cur_list ^ 1`
  auto launch_q = [&](const int cur_list, const uint32_t n_host, const bool n_on_device, const int spatial) {
    const int nxt = cur_list ^ 1;
    const int c_in = C_LIST0 + (int)(launches % 3), c_out = C_LIST0 + (int)((launches + 1) % 3), c_zero = C_LIST0 + (int)((launches + 2) % 3);
    const unsigned long long *n_dev = n_on_device ? &counters_[c_in] : nullptr;
    ++serial_;
    RelaxQArgs a;
    a.g = g_;
    a.tg = tg;
    a.coc = coc_;
    a.rbits = rbits_;
    a.tile_epoch = tile_epoch_;
    a.epoch = epoch_;
    a.cbits_prev = cbits_[(serial_ - 1) & 1];
    a.cbits_cur = cbits_[serial_ & 1];
    a.cstamp_prev = cstamp_[(serial_ - 1) & 1];
    a.cstamp_cur = cstamp_[serial_ & 1];
    a.serial = serial_;
    a.list_cur = tile_list_[cur_list];
    a.n_cur = n_host;
    a.n_cur_dev = n_dev;
    a.flag_cur = tile_flag_[cur_list];
    a.flag_next = tile_flag_[nxt];
    a.list_next = tile_list_[nxt];
    a.count_next = &counters_[c_out];
    a.count_zero = &counters_[c_zero];
    a.counters = counters_;
    a.prof = prof_;
    a.dir = nullptr;
    a.spatial = spatial;
    // spatial walk: a multiple of 8 blocks (one stream per XCD), a few per CU for load balance.  A round of a chain: one
    // work-group per CU striding over the list -- the tile's keys fill a CU's LDS, so more work-groups than CUs only
    // queue, and a round that finds its list empty should cost as little as a launch can.
    const int blocks = spatial ? (int)std::min<uint32_t>((uint32_t)((ntiles_ + 7) / 8 * 8), (uint32_t)spatial_blocks_)
                       : n_dev ? 256
                               : (int)std::min<uint32_t>(n_host, 16384u);
    if (track_)
      hipLaunchKernelGGL((k_relax_q<16, 16, 1024, false, true>), dim3(blocks), dim3(1024), 0, stream_, a);
    else
      hipLaunchKernelGGL((k_relax_q<16, 16, 1024>), dim3(blocks), dim3(1024), 0, stream_, a);
    FIESTA_HIP_CHECK(hipGetLastError());
    ++launches;
  };
  const bool unknown = first_count == kCountOnDevice;  // small update: nobody has read the length of the first list
  while (ncur) {
    // Large updates walk all tiles in XCD-chunked spatial order, one round per host round trip.  Small ones (few active
    // tiles: depth frames) use the compact list and go out in chains, each round reading the length of its list on the
    // device and doing nothing once a predecessor activated no tile: ONE host round trip per chain instead of one per
    // round (a depth frame's update is a handful of short kernels; the round trips were most of its time).  A chain is
    // as long as the previous update's rounds plus one: consecutive frames of a sensor need about the same number.
    FIESTA_HIP_CHECK(hipEventRecord(pool_event(2 * nev), stream_));
    if (!unknown && spatial_ && ncur >= (uint32_t)list_threshold_) {
      launch_q(cur, ncur, false, 1);
      FIESTA_HIP_CHECK(hipEventRecord(pool_event(2 * nev + 1), stream_));
      ++nev;
      ++spatial_rounds;
      cur ^= 1;
      ncur = (uint32_t)read_counter(C_LIST0 + (int)(launches % 3));
      lists_dirty = true;  // (the round's own input counter is still set)
    } else {
      const int chain = device_rounds < 0 ? std::min(std::max(chain_hint_, 2), 12) : 4;
      for (int k = 0; k < chain; ++k) {
        launch_q(cur, 0, true, 0);
        cur ^= 1;
      }
      FIESTA_HIP_CHECK(hipEventRecord(pool_event(2 * nev + 1), stream_));
      last_chain_event_ = evpool_[2 * nev + 1];
      ++nev;
      // (everything collect_stats wants is in this copy too: an update that ends here needs no second round trip)
      FIESTA_HIP_CHECK(hipMemcpyAsync(h_counters_, counters_, C_COUNT * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream_));
      FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
      ncur = (uint32_t)h_counters_[C_LIST0 + (int)(launches % 3)];
      device_rounds = (int64_t)h_counters_[C_ROUNDS];
      h_counters_fresh_ = ncur == 0;
      // (a chain that ends on a round with work leaves that round's input counter set; trailing idle rounds clear it)
      lists_dirty = h_counters_[C_LIST0] || h_counters_[C_LIST1] || h_counters_[C_LIST2];
    }
  }
  if (lists_dirty) zero_counters(C_LIST0, 3);  // between updates the three list counters are zero
  if (device_rounds >= 0) chain_hint_ = (int)device_rounds + 1;
  if (st) {
    st->rounds = device_rounds >= 0 ? device_rounds + spatial_rounds : launches;
    double sum = 0;
    for (size_t r = 0; r < nev; ++r) {
      float ms = 0;
      FIESTA_HIP_CHECK(hipEventElapsedTime(&ms, evpool_[2 * r], evpool_[2 * r + 1]));
      sum += ms;
    }
    st->relax_ms = sum;
    st->relax_launches = launches;
  }
}

// ---- the bulk path: whole-grid exact feature transform (ft_core.hpp / ft_kernels.hpp) -------------------------------
// Valid when the reference's propagation has nothing to be gated by: every voxel of the array observed, the update
// window = the whole array, one unsharded map.  Then the fixed point of src/ESDFMap.cpp:339-392 IS the Euclidean
// feature transform of the occupied set, whatever the previous state was.
template <int S, int LANES, int WAVES, bool WIDE>
static void launch_ft_plane(const FtArgs &a, int blocks, hipStream_t s) {
  hipLaunchKernelGGL((k_ft_plane<S, LANES, WAVES, WIDE>), dim3(blocks), dim3(64 * WAVES), 0, s, a);
}
template <int S, int LANES, int WAVES, bool WIDE>
static void launch_ft_x(const FtArgs &a, int blocks, hipStream_t s) {
  if (a.maxd2)  // (the largest distance written is only tracked where somebody asks for it: shards, distance-bounded scans)
    hipLaunchKernelGGL((k_ft_x<S, LANES, WAVES, WIDE, true>), dim3(blocks), dim3(64 * WAVES), 0, s, a);
  else
    hipLaunchKernelGGL((k_ft_x<S, LANES, WAVES, WIDE, false>), dim3(blocks), dim3(64 * WAVES), 0, s, a);
}

// The transform of this map's array.  Unsharded: the region is the array, the bitmap the map's own.  Sharded: the
// region is the local array (owned box + ghost layers) grown by `margin` voxels towards every neighbour shard and read
// from this shard's replica of the GLOBAL occupancy bitmap -- no communication at all.  The result is exact iff every
// voxel written found its obstacle within the margin (an obstacle outside the region is farther than the margin from
// every voxel of the array); *exact reports that, from the largest distance written.  Returns false (nothing done) if the
// region exceeds the 1024 voxels per axis the transform's site packing allows.
bool DenseMap::run_bulk(fiesta_hip_stats *st, int margin, bool *exact) {
  const Geom &g = g_;
  FtArgs a;
  memset(&a, 0, sizeof(a));
  int rlo[3], rhi[3], mlo[3], mhi[3];  // region in GLOBAL coordinates (inclusive); margins actually obtained per side
  const int l0[3] = {g.gx0, g.gy0, g.gz0}, ln[3] = {g.nx, g.ny, g.nz}, G[3] = {g.GX, g.GY, g.GZ};
  bool open_side = false;  // some side of the region does not reach the global boundary (needs the margin test)
  bool wide = g.wrap != 0;  // site packing of the transform (ft_kernels.hpp: FtPack)
  for (int k = 0; k < 3; ++k) {
    rlo[k] = g.sharded ? std::max(0, l0[k] - margin) : l0[k];
    rhi[k] = g.sharded ? std::min(G[k] - 1, l0[k] + ln[k] - 1 + margin) : l0[k] + ln[k] - 1;
  }
  if (g.sharded) {  // whole bitmap words along z
    rlo[2] &= ~31;
    rhi[2] = std::min(G[2] - 1, rhi[2] | 31);
  }
  for (int k = 0; k < 3; ++k) {
    mlo[k] = l0[k] - rlo[k], mhi[k] = rhi[k] - (l0[k] + ln[k] - 1);
    if (g.sharded && ((rlo[k] > 0) || (rhi[k] < G[k] - 1))) open_side = true;
    if (rhi[k] - rlo[k] + 1 > 2048) return false;
    if (rhi[k] - rlo[k] + 1 > 1024) wide = true;
  }
  a.nx = rhi[0] - rlo[0] + 1, a.ny = rhi[1] - rlo[1] + 1, a.nz = rhi[2] - rlo[2] + 1;
  a.nzw = (a.nz + 31) / 32, a.nzc = (a.nz + 63) / 64;
  a.gx0 = rlo[0], a.gy0 = rlo[1], a.gz0 = rlo[2];
  if (g.sharded) {
    a.src = gocc_, a.sx0 = rlo[0], a.sy0 = rlo[1], a.sw0 = rlo[2] / 32, a.sny = g.GY, a.snzw = g.GZW;
  } else {
    a.src = tr_occ_ ? tr_occ_ : occbits_, a.sx0 = 0, a.sy0 = 0, a.sw0 = 0, a.sny = g.ny, a.snzw = g.nzw;
  }
  a.ox0 = mlo[0], a.oy0 = mlo[1], a.oz0 = mlo[2];
  a.onx = g.nx, a.ony = g.ny, a.onz = g.nz;
  const int64_t rn = (int64_t)a.nx * a.ny * a.nz;
  const uint32_t items_a = (uint32_t)(a.nx * a.nzc), items_b = (uint32_t)(a.ny * a.nzc);
  const uint32_t cap = std::max(items_a, items_b);
  ft_inter_.ensure((size_t)rn, stream_);
  ft_rowlist_.ensure((size_t)a.nx * a.ny, stream_);
  ft_rowcnt_.ensure((size_t)a.nx + 64, stream_);  // + the 2048-bit plane mask
  // backing store of the rings: one slice per wave of the passes' grid (at most kFtBlocks work-groups of 4 waves), one
  // 512-byte slot row per counter value of the longest column (ft_core.hpp: a deque deeper than its ring)
  const uint32_t spill_stride = (uint32_t)(std::max(a.nx, a.ny) + 2) * 512u;
  const uint32_t spill_blocks = std::min<uint32_t>((cap + 3) / 4, (uint32_t)kFtBlocks);  // (the larger pass's grid)
  ft_spill_.ensure_exact((size_t)spill_blocks * 4 * (spill_stride / sizeof(unsigned long long)), stream_);
  a.rowlist = ft_rowlist_.p;
  a.rowcnt = ft_rowcnt_.p;
  a.inter = ft_inter_.p;
  // a shard's transform lands in a side buffer first: it only replaces the field once every shard has confirmed that
  // its margin sufficed (bulk_commit); if not, the frontier rounds take over from the untouched field
  // ... unless the margin test cannot fail: a region that reaches the global boundary on every side (a single shard, or
  // margins grown to the whole grid) holds every obstacle there is -- the result is final and goes in place
  // -- and only in a group of ONE: with several shards the commit waits for every shard's verdict, and an error on another
  // shard between try and commit (an allocation failing there) must find this shard's field untouched
  ft_in_place_ = !g.sharded || (!open_side && alone_in_group_);
  if (!ft_in_place_) ft_out_.ensure_exact((size_t)g.n, stream_);
  a.coc = tr_out_ ? tr_out_ : ft_in_place_ ? coc_ : ft_out_.p;  // (tr_out_: a masked transform's side buffer, run_masked)
  const bool want_max = (track_ && !tr_out_) || open_side;
  a.maxd2 = want_max ? &counters_[C_FT_MAXD2] : nullptr;
  if (!ft_counters_clean_)  // (a second run within one update: the spill counters of the first are still there)
    FIESTA_HIP_CHECK(hipMemsetAsync(&counters_[C_FT_OVF0], 0, 7 * sizeof(unsigned long long), stream_));  // + C_FT_MAXD2
  ft_counters_clean_ = false;
  FIESTA_HIP_CHECK(hipEventRecord(ft_ev_[0], stream_));
  hipLaunchKernelGGL(k_ft_rows, dim3(a.nx), dim3(256), 0, stream_, a);
  FIESTA_HIP_CHECK(hipGetLastError());
  FIESTA_HIP_CHECK(hipEventRecord(ft_ev_[1], stream_));
  // ONE kernel per pass, whatever the scene: rings of ft_s0_ entries in LDS, and a deque that outgrows its ring goes on
  // in the backing store (r02/r03a: overflowing column groups were redone by further launches with bigger rings -- a
  // pass is one round of waves, as long as one item takes, so every such launch cost a whole pass however few items
  // it held: 0.45 ms for 11 % of pass B's items on the surfaces scene).  The grid is capped at what is resident at once.
  a.spill = reinterpret_cast<char *>(ft_spill_.p);
  a.spill_stride = spill_stride;
  auto pass = [&](const bool pass_a, const uint32_t n0, const int counter) {
    FtArgs t = a;
    t.n_items = n0;
    t.spill_count = &counters_[counter];
    const int blocks = (int)std::min<uint32_t>((n0 + 3) / 4, (uint32_t)kFtBlocks);
    if (pass_a) {
      if (wide) {
        if (ft_s0_ == 16) launch_ft_plane<16, 64, 4, true>(t, blocks, stream_);
        else launch_ft_plane<32, 64, 4, true>(t, blocks, stream_);
      } else {
        if (ft_s0_ == 16) launch_ft_plane<16, 64, 4, false>(t, blocks, stream_);
        else launch_ft_plane<32, 64, 4, false>(t, blocks, stream_);
      }
    } else {
      if (wide) {
        if (ft_s0_ == 16) launch_ft_x<16, 64, 4, true>(t, blocks, stream_);
        else launch_ft_x<32, 64, 4, true>(t, blocks, stream_);
      } else {
        if (ft_s0_ == 16) launch_ft_x<16, 64, 4, false>(t, blocks, stream_);
        else launch_ft_x<32, 64, 4, false>(t, blocks, stream_);
      }
    }
    FIESTA_HIP_CHECK(hipGetLastError());
  };
  pass(true, items_a, C_FT_OVF0);
  FIESTA_HIP_CHECK(hipEventRecord(ft_ev_[2], stream_));
  pass(false, items_b, C_FT_OVF0 + 3);
  const int launches = 3;
  FIESTA_HIP_CHECK(hipEventRecord(ft_ev_[3], stream_));
  if (st) {
    st->bulk = 1;
    st->relax_launches = launches;
  }
  if (want_max) {
    const unsigned long long dmax2 = read_counter(C_FT_MAXD2);
    if (track_ && !tr_out_) {
      h_counters_[C_MAXD2] = dmax2;
      FIESTA_HIP_CHECK(hipMemcpyAsync(&counters_[C_MAXD2], &h_counters_[C_MAXD2], sizeof(unsigned long long), hipMemcpyHostToDevice, stream_));
    }
    // (grids beyond 1024 per axis: nothing farther than an id's reach of 512 voxels is ever stored, so a margin of
    //  512 always suffices there)
    const unsigned long long need2 = g.wrap ? std::min<unsigned long long>(dmax2, (unsigned long long)kD2Cap) : dmax2;
    bool ok = true;
    for (int k = 0; k < 3; ++k) {
      if (rlo[k] > 0 && (unsigned long long)mlo[k] * mlo[k] < need2) ok = false;
      if (rhi[k] < G[k] - 1 && (unsigned long long)mhi[k] * mhi[k] < need2) ok = false;
    }
    if (exact) *exact = ok || !g.sharded;
    if (st) st->ft_max_d2 = (int64_t)dmax2;
  } else if (exact) {
    *exact = true;
  }
  return true;
}

// ---- the cell transform (nn_core.hpp / nn_kernels.hpp): the same fixed point for a sparse obstacle set ----------------------
// Applies to maps, and to the shards of any grid, whose region -- the array, a shard's array plus margin -- stays within
// nn::kRegionMax voxels per axis (beyond 1024 the sites are stored modulo 1024, as the voxel words of such grids are).  Whether
// it is worth trying: the obstacle density must be in the range where every cell finds an obstacle within its search window and
// lists stay short (measured on scatter scenes, profiles/r05h_density_range.json: 8e-5 ... 2.5e-3 of the voxels; config 2's
// scene is 3.7e-4), it must not have failed at about this obstacle count, and it must not have been slower than the envelope passes.
bool DenseMap::cells_wanted() {
  const Geom &g = g_;
  if (update_engine_ == 4 || g.nx > nn::kRegionMax || g.ny > nn::kRegionMax || g.nz > nn::kRegionMax) return false;
  if (update_engine_ == 5) return true;
  const long long nocc = (long long)h_counters_[C_NOCC];
  // (measured with the cells that get no list served one by one, profiles/r06_density_range.json: 7.5e-5 -- 0.36 against 0.57 ms on
  //  the envelope passes; 5.2e-5 -- two cells scan every site, 0.89 against 0.56; 2.5e-3 -- 0.74 against 0.98; 3.7e-3 -- 1.14 against 1.04)
  if (nocc * 15000 < g.n || nocc * 400 > g.n) {
    notes_ |= FIESTA_HIP_NOTE_DENSITY;
    return false;
  }
  // a failed attempt (a cell without a list: ~0.2 ms lost before the envelope passes take over) is not repeated at once: the
  // next 8, 16, ... 256 eligible updates go straight to the envelope passes, then it is tried again -- a scene that cannot be
  // served costs 1 % in the long run, one unlucky cell in a scene that can does not switch the transform off for good
  if (nn_skip_ > 0) {
    --nn_skip_;
    notes_ |= FIESTA_HIP_NOTE_CELLS_BACKOFF;
    return false;
  }
  if (nn_last_ms_ > 0 && ft_last_ms_ > 0 && nn_last_ms_ > ft_last_ms_ && std::llabs(nocc - nn_last_nocc_) * 4 <= nn_last_nocc_) {
    notes_ |= FIESTA_HIP_NOTE_CELLS_BACKOFF;
    return false;
  }
  return true;
}

// The cell transform of this map's array.  Unsharded: the region is the array, the bitmap the map's own.  Sharded: the
// region is the array grown by `margin` (+ what a search window adds to a distance: a cell's window reaches
// |competitor - centre| + 14 voxels) towards the neighbour shards, cut out of the replica of the GLOBAL bitmap
// (nn_core.hpp: region_geom) -- no communication, like run_bulk.  Exactness is decided cell by cell on the device: a cell
// whose window touches an open face of the region fails, and a failed cell fails the transform (the caller reads
// C_NN_FAILED and takes the envelope passes).  Returns false (nothing launched) if the transform does not apply to this map.
bool DenseMap::run_cells(fiesta_hip_stats *st, int margin, bool publish, bool incremental, unsigned long long ni, unsigned long long nd) {
  const Geom &g = g_;
  if (g.nx > nn::kRegionMax || g.ny > nn::kRegionMax || g.nz > nn::kRegionMax) return false;
  NnArgs a;
  memset(&a, 0, sizeof(a));
  int rlo[3] = {0, 0, 0};
  bool open_side = false;
  if (g.sharded) {
    const int G[3] = {g.GX, g.GY, g.GZ}, l0[3] = {g.gx0, g.gy0, g.gz0}, ln[3] = {g.nx, g.ny, g.nz};
    const int mc = (margin + 24 + nn::kB - 1) / nn::kB * nn::kB;
    if (!nn::region_geom(G, l0, ln, mc, a.g, rlo)) return false;
    open_side = a.g.open != 0;
    a.occ = gocc_, a.sx0 = rlo[0], a.sy0 = rlo[1], a.szb = rlo[2] / 8, a.sny = g.GY, a.snzw = g.GZW;
  } else {
    if ((g.gx0 | g.gy0 | g.gz0) != 0) return false;
    a.g = nn::whole_geom(g.nx, g.ny, g.nz);
    a.occ = tr_occ_ ? tr_occ_ : occbits_, a.sny = g.ny, a.snzw = g.nzw;
    a.cellobs = tr_cellobs_;  // (a masked transform: cells nobody ever observed get no list)
  }
  const int64_t nrows = (int64_t)a.g.ncx * a.g.ncy, ncells = nrows * a.g.ncz;
  // sites: the obstacles of the region -- an unsharded map knows the count (k_fuse keeps it exact); a shard's region also
  // holds obstacles of its neighbours: room for the region's share of a scene at the densest the transform is tried at,
  // and k_nn_cells fails the transform if that does not suffice
  const size_t sites_cap = g.sharded ? (size_t)std::max<long long>(4 * (long long)h_counters_[C_NOCC] + 4096,
                                                                   (long long)a.g.nx * a.g.ny * a.g.nz / 256)
                                     : (size_t)std::max<long long>((long long)h_counters_[C_NOCC], 0) + 64;
  nn_ctab_.ensure_exact((size_t)nrows * (a.g.ncz + 1), stream_);
  nn_sites_.ensure(sites_cap, stream_);
  nn_lists_.ensure_exact((size_t)ncells * nn::kStride + kListPad, stream_);
  a.ctab = nn_ctab_.p, a.sites = nn_sites_.p, a.sites_cap = (uint32_t)std::min<size_t>(nn_sites_.cap, 0xFFFFFFFFu);
  a.lists = nn_lists_.p;
  a.dump = nn_lists_.p + (size_t)ncells * nn::kStride + (kListPad - 128);
  if (incremental) {
    if (nn_dirty_flag_.cap < (size_t)ncells) {
      nn_dirty_flag_.ensure_exact((size_t)ncells, stream_);
      FIESTA_HIP_CHECK(hipMemsetAsync(nn_dirty_flag_.p, 0, (size_t)ncells * sizeof(uint32_t), stream_));
    }
    nn_dirty_list_.ensure_exact((size_t)ncells / 2 + 64, stream_);
    a.chg[0] = ins_.p, a.chg[1] = del_.p, a.nchg[0] = (uint32_t)ni, a.nchg[1] = (uint32_t)nd;
    a.dirty_flag = nn_dirty_flag_.p, a.dirty_list = nn_dirty_list_.p, a.dirty_cap = (uint32_t)(ncells / 2);
  }
  a.dirty_count = &counters_[C_NN_DIRTY];
  a.ticket = &counters_[C_FUSE_TICKET];
  if (!g.sharded && !a.g.big()) {  // cells without a list are served one by one (k_nn_close), up to a 64th of the cells
    a.fail_cap = (uint32_t)std::min<int64_t>(std::max<int64_t>(ncells / 64, 16), 4096);
    nn_fail_list_.ensure((size_t)a.fail_cap, stream_);
    a.fail_list = nn_fail_list_.p, a.nfail = &counters_[C_NN_BRUTE];
  }
  const unsigned close_blocks = nn_last_brute_ > 0 ? (unsigned)std::min<long long>(nn_last_brute_ + 8, 512) : 1u;
  a.cursor = &counters_[C_NN_CURSOR], a.failed = &counters_[C_NN_FAILED], a.entries = &counters_[C_NN_ENTRIES];
  // (the largest distance written: maps that track it, and shards -- the group sizes the next margin from it)
  const bool want_max = (track_ && !tr_out_) || open_side;
  a.maxd2 = want_max ? &counters_[C_FT_MAXD2] : nullptr;
  // a shard's transform lands in a side buffer unless nobody else has a say (run_bulk)
  ft_in_place_ = !g.sharded || (!open_side && alone_in_group_);
  if (!ft_in_place_) ft_out_.ensure_exact((size_t)g.n, stream_);
  a.coc = tr_out_ ? tr_out_ : ft_in_place_ ? coc_ : ft_out_.p;  // (tr_out_: a masked transform's side buffer, run_masked)
  if (publish) {  // a last one-thread launch reports into h_counters_ and cleans up (nn_kernels.hpp: k_nn_close)
    a.pub = h_counters_, a.queues = &counters_[C_INSERT], a.track_dst = track_ ? &counters_[C_MAXD2] : nullptr;
    a.tag = ++nn_tag_;
    a.pub_failed = C_NN_FAILED, a.pub_entries = C_NN_ENTRIES, a.pub_maxd2 = C_FT_MAXD2, a.pub_tag = C_NN_CURSOR, a.pub_dirty = C_NN_DIRTY;
    a.pub_brute = C_NN_BRUTE;
  }
  if (!ft_counters_clean_)
    FIESTA_HIP_CHECK(hipMemsetAsync(&counters_[C_FT_OVF0], 0, 7 * sizeof(unsigned long long), stream_));  // + C_FT_MAXD2
  ft_counters_clean_ = false;
  FIESTA_HIP_CHECK(hipEventRecord(ft_ev_[0], stream_));
  // (the region's rows start dword-aligned: four wide loads per lane and a turn in LDS instead of 64 byte loads)
  if (a.szb % 4 == 0) hipLaunchKernelGGL(k_nn_cells<true>, dim3((unsigned)((nrows + 15) / 16)), dim3(1024), 0, stream_, a);
  else hipLaunchKernelGGL(k_nn_cells<false>, dim3((unsigned)((nrows + 15) / 16)), dim3(1024), 0, stream_, a);
  FIESTA_HIP_CHECK(hipGetLastError());
  FIESTA_HIP_CHECK(hipEventRecord(ft_ev_[1], stream_));
  const int lcx = a.g.lx1 - a.g.lx0, lcy = a.g.ly1 - a.g.ly0, lcz = a.g.lz1 - a.g.lz0;  // the cells that get a list
  if (incremental) {
    // which cells a changed voxel can reach, their lists, their fill: three short launches that find their work on the device
    hipLaunchKernelGGL(k_nn_mark, dim3((unsigned)std::min<unsigned long long>((ni + nd) * 3375 / 1024 + 1, 256)), dim3(1024), 0, stream_, a);
    const unsigned est = (unsigned)std::min<unsigned long long>((ni + nd) * 200 + 256, (unsigned long long)ncells / 2);
    hipLaunchKernelGGL(k_nn_lists_dirty, dim3(std::min(std::max(est / 16u, 64u), 8192u)), dim3(256), 0, stream_, a);
    FIESTA_HIP_CHECK(hipGetLastError());
    FIESTA_HIP_CHECK(hipEventRecord(ft_ev_[2], stream_));
    if (want_max) hipLaunchKernelGGL(k_nn_fill_dirty<true>, dim3(std::min(std::max(est / 4u, 64u), 8192u)), dim3(256), 0, stream_, a);
    else hipLaunchKernelGGL(k_nn_fill_dirty<false>, dim3(std::min(std::max(est / 4u, 64u), 8192u)), dim3(256), 0, stream_, a);
    FIESTA_HIP_CHECK(hipGetLastError());
    FIESTA_HIP_CHECK(hipEventRecord(ft_ev_[3], stream_));
    if (publish) {
      NnArgs c = a;
      c.nfail = nullptr;  // (an incremental transform fails on a cell without a list: the full one serves it)
      hipLaunchKernelGGL(k_nn_close, dim3(1), dim3(256), 0, stream_, c);
      FIESTA_HIP_CHECK(hipGetLastError());
    }
    if (st) {
      st->bulk = 1;
      st->cells = 1;
      st->nn_incremental = 1;
      st->relax_launches = 4;
    }
    return true;
  }
  if (a.g.big()) hipLaunchKernelGGL(k_nn_lists<true>, dim3((lcz + 63) / 64, (lcy + 3) / 4, lcx), dim3(1024), 0, stream_, a);
  else hipLaunchKernelGGL(k_nn_lists<false>, dim3((lcz + 63) / 64, (lcy + 3) / 4, lcx), dim3(1024), 0, stream_, a);
  FIESTA_HIP_CHECK(hipGetLastError());
  FIESTA_HIP_CHECK(hipEventRecord(ft_ev_[2], stream_));
  // the unpredicated variant: the region IS the array, whole cells everywhere and the same number of quads for every wave
  // (k_nn_fill_full)
  const int64_t nq = nrows * ((a.g.ncz + 3) / 4);
  unsigned fill_blocks = 0;
  bool full = (a.g.fx | a.g.fy | a.g.fz) == 0 && a.g.nx == g.nx && a.g.ny == g.ny && a.g.nz == g.nz && g.nx % nn::kB == 0 &&
              g.ny % nn::kB == 0 && g.nz % (4 * nn::kB) == 0;
  if (full) {
    unsigned b = (unsigned)std::min<int64_t>(nq / 4, kFillBlocks);
    while (b >= 64 && nq % (4 * (int64_t)b) != 0) --b;
    if (b >= 64) fill_blocks = b; else full = false;
  }
  const dim3 cell_grid((lcz + 3) / 4, lcy, lcx);
  // arrays at an offset of their region, pairs of voxels aligned: the persistent fill with 8-byte stores (k_nn_fill_full<., true>)
  const bool pairs = !full && (a.g.az % 2) == 0 && (a.g.fz % 2) == 0;
  // (persistent waves there too: runs of at least four quads where the map has them)
  const int64_t nq_off = (int64_t)lcx * lcy * ((lcz + 3) / 4);
  const unsigned off_blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(kFillBlocks, (nq_off + 15) / 16));
  if (want_max) {
    if (full) hipLaunchKernelGGL((k_nn_fill_full<true>), dim3(fill_blocks), dim3(256), 0, stream_, a);
    else if (pairs) hipLaunchKernelGGL((k_nn_fill_full<true, true>), dim3(off_blocks), dim3(256), 0, stream_, a);
    else hipLaunchKernelGGL((k_nn_fill<true>), cell_grid, dim3(256), 0, stream_, a);
  } else {
    if (full) hipLaunchKernelGGL((k_nn_fill_full<false>), dim3(fill_blocks), dim3(256), 0, stream_, a);
    else if (pairs) hipLaunchKernelGGL((k_nn_fill_full<false, true>), dim3(off_blocks), dim3(256), 0, stream_, a);
    else hipLaunchKernelGGL((k_nn_fill<false>), cell_grid, dim3(256), 0, stream_, a);
  }
  FIESTA_HIP_CHECK(hipGetLastError());
  FIESTA_HIP_CHECK(hipEventRecord(ft_ev_[3], stream_));
  if (publish || a.nfail) {  // the cells without a list, and (publish) the report
    hipLaunchKernelGGL(k_nn_close, dim3(close_blocks), dim3(256), 0, stream_, a);
    FIESTA_HIP_CHECK(hipGetLastError());
  }
  if (track_ && !publish && !tr_out_)  // (a failed transform leaves 0 here; the envelope passes that follow it set the bound themselves)
    FIESTA_HIP_CHECK(hipMemcpyAsync(&counters_[C_MAXD2], &counters_[C_FT_MAXD2], sizeof(unsigned long long), hipMemcpyDeviceToDevice, stream_));
  if (st) {
    st->bulk = 1;
    st->cells = 1;
    st->relax_launches = 3;
  }
  return true;
}

// The map-local half of the engine choice: may this update be served by the bulk transform at all?
bool DenseMap::bulk_eligible(unsigned long long ni, unsigned long long nd) {
  const Geom &g = g_;
  // (a map that held no obstacle before this update has no history left: update_esdf clears the flag ahead of recording
  //  this update's own window; here only when the window is the whole array -- the sharded driver's probe comes this way)
  const bool full_win = g.wx0 <= 0 && g.wy0 <= 0 && g.wz0 <= 0 && g.wx1 >= g.nx - 1 && g.wy1 >= g.ny - 1 && g.wz1 >= g.nz - 1;
  if (win_dirty_ && full_win && !g.sharded && (long long)h_counters_[C_NOCC] - (long long)ni + (long long)nd <= 0) win_dirty_ = false;
  if (update_engine_ == 1 || update_engine_ == 3) return false;
  if (!(g.wx0 <= 0 && g.wy0 <= 0 && g.wz0 <= 0 && g.wx1 >= g.nx - 1 && g.wy1 >= g.ny - 1 && g.wz1 >= g.nz - 1)) return false;
  const long long owned = (long long)(g.ox1 - g.ox0 + 1) * (g.oy1 - g.oy0 + 1) * (g.oz1 - g.oz0 + 1);
  if ((long long)h_counters_[C_OBSERVED] != owned) return false;
  // An update that ran under a partial window left the voxels outside it as they were (inserts never reach them, the
  // orphans of a delete keep what one pull gave them, src/ESDFMap.cpp:351,378): from then on the reference's field is a
  // function of its history, not the transform of the occupied set -- until the map holds no obstacle again.
  if (win_dirty_) {
    const long long before = (long long)h_counters_[C_NOCC] - (long long)ni + (long long)nd;
    if (!g.sharded && before <= 0) win_dirty_ = false;
  }
  if (win_dirty_) return false;
  if (stale_inf_) {  // re-validate: does any observed voxel still wait for its first wave?
    zero_counter(C_SCRATCH);
    hipLaunchKernelGGL(k_count_stale, dim3(grid_for(g_.n, 256, 4096)), dim3(256), 0, stream_, g_, (const vox_t *)coc_,
                       &counters_[C_SCRATCH]);
    FIESTA_HIP_CHECK(hipGetLastError());
    // (with no obstacle before this update every voxel legitimately reads "no obstacle"; a shard cannot tell from its own
    //  counters whether another shard held one, so it only trusts the scan)
    const long long before = (long long)h_counters_[C_NOCC] - (long long)ni + (long long)nd;
    if (read_counter(C_SCRATCH) == 0 || (!g.sharded && before <= 0)) stale_inf_ = false;
  }
  if (stale_inf_) notes_ |= FIESTA_HIP_NOTE_FIRST_WAVE_PENDING;
  return !stale_inf_;
}

// ---- the masked transform (mask_kernels.hpp): large deltas on partially observed maps -------------------------------------------
// May this update be served by it?  The same history conditions as the transform of a fully observed map (bulk_eligible: whole
// window, no update under a partial window since the map last held no obstacle) -- and no voxel still waiting for its first
// wave: a voxel first observed while obstacles existed reads "no obstacle" in the reference until a wave reaches it
// (src/ESDFMap.cpp:246-249), which no function of (occupied set, observed set) reproduces.  k_fuse marks and counts such voxels
// (C_LATE, latebits_); the count falls when one of them becomes an obstacle (the insert drain seeds it) and, found by a rescan
// here, when a wave has given it an obstacle.
bool DenseMap::masked_eligible(unsigned long long ni, unsigned long long nd) {
  const Geom &g = g_;
  if (g.sharded || g.wrap || update_engine_ == 1 || update_engine_ == 3) return false;
  if (g.nx > nn::kRegionMax || g.ny > nn::kRegionMax || g.nz > nn::kRegionMax) return false;
  if (!(g.wx0 <= 0 && g.wy0 <= 0 && g.wz0 <= 0 && g.wx1 >= g.nx - 1 && g.wy1 >= g.ny - 1 && g.wz1 >= g.nz - 1)) return false;
  if (win_dirty_) return false;  // (update_esdf has cleared it if the map held no obstacle before this update)
  const long long before = (long long)h_counters_[C_NOCC] - (long long)ni + (long long)nd;
  unsigned long long late = h_counters_[C_LATE];
  if (late != 0 && before <= 0) {
    // no obstacle before this update: every voxel legitimately reads "no obstacle", nobody waits for anything
    FIESTA_HIP_CHECK(hipMemsetAsync(latebits_, 0, nbitwords_ * sizeof(uint32_t), stream_));
    zero_counter(C_LATE);
    h_counters_[C_LATE] = late = 0;
  }
  if (late != 0) {  // has a wave reached them since?
    hipLaunchKernelGGL(k_late_rescan, dim3(grid_for(nbitwords_, 256, 4096)), dim3(256), 0, stream_, g_, (const vox_t *)coc_,
                       (const uint32_t *)occbits_, latebits_, nbitwords_, &counters_[C_LATE]);
    FIESTA_HIP_CHECK(hipGetLastError());
    h_counters_[C_LATE] = late = read_counter(C_LATE);
  }
  if (late != 0) notes_ |= FIESTA_HIP_NOTE_LATE_OBSERVATION;
  return late == 0;
}

// One UpdateESDF by the masked transform.  Everything is launched behind one another -- summary of the observed bitmap, the
// sites, the transform into the side buffer, the certificate, a first chain of repair iterations -- and read back ONCE; further
// chains only if the repair has not settled.  Returns false with the field untouched (the side buffer is simply dropped) if the
// repair list outgrew its buffer: the caller's frontier rounds then serve the update.
bool DenseMap::run_masked(fiesta_hip_stats *st, std::chrono::steady_clock::time_point h0) {
  const Geom &g = g_;
  const int ncx = (g.nx + 7) / 8, ncy = (g.ny + 7) / 8, ncz = (g.nz + 7) / 8;
  const int64_t ncells = (int64_t)ncx * ncy * ncz, nquads = (int64_t)ncx * ncy * g.nzw;
  if (!mask_ctr_) {
    FIESTA_HIP_CHECK(hipMalloc((void **)&mask_ctr_, MC_COUNT * sizeof(unsigned long long)));
    FIESTA_HIP_CHECK(hipHostMalloc((void **)&h_mask_ctr_, MC_COUNT * sizeof(unsigned long long)));
  }
  // the walk list: room for a quarter of the voxels (measured on config 2's partially observed scene: 13 % of the observed
  // voxels need the walk), in kMaskSegs segments that the quads feed in turn
  if (mask_seg_cap_ == 0) mask_seg_cap_ = (size_t)std::max<int64_t>(g.n / 4 / kMaskSegs, 8192);
  const size_t seg_cap = mask_seg_cap_;
  effocc_.ensure_exact((size_t)nbitwords_, stream_);
  mask_ubits_.ensure_exact((size_t)nbitwords_, stream_);
  cellobs_.ensure_exact((size_t)ncells, stream_);
  celldist_.ensure_exact((size_t)ncells, stream_);
  cellnb_.ensure_exact((size_t)ncells, stream_);
  cellst_.ensure_exact((size_t)ncells, stream_);
  mask_out_.ensure_exact((size_t)g.n, stream_);
  mask_walks_.ensure_exact(seg_cap * kMaskSegs, stream_);
  mask_uq_.ensure_exact((size_t)ncells, stream_);
  if (mask_qstamp_.cap < (size_t)(2 * ncells) || mask_serial_ > 0xFFFF0000u) {  // (tags never repeat within the stamps' lifetime)
    mask_qstamp_.ensure_exact((size_t)(2 * ncells), stream_);
    FIESTA_HIP_CHECK(hipMemsetAsync(mask_qstamp_.p, 0, (size_t)(2 * ncells) * sizeof(uint32_t), stream_));
    mask_serial_ = 0;
  }
  MaskArgs ma;
  memset(&ma, 0, sizeof(ma));
  ma.g = g, ma.ncx = ncx, ma.ncy = ncy, ma.ncz = ncz;
  ma.occbits = occbits_, ma.obsbits = obsbits_, ma.effocc = effocc_.p, ma.cellobs = cellobs_.p, ma.celldist = celldist_.p, ma.cellnb = cellnb_.p;
  ma.cellst = cellst_.p;
  ma.old = coc_, ma.out = mask_out_.p;
  ma.ubits = mask_ubits_.p, ma.walks = reinterpret_cast<uint2 *>(mask_walks_.p), ma.seg_cap = (uint32_t)seg_cap;
  ma.uq = mask_uq_.p, ma.qstamp[0] = mask_qstamp_.p, ma.qstamp[1] = mask_qstamp_.p + ncells;
  ma.ctr = mask_ctr_;
  FIESTA_HIP_CHECK(hipMemsetAsync(mask_ctr_, 0, MC_COUNT * sizeof(unsigned long long), stream_));
  FIESTA_HIP_CHECK(hipMemsetAsync(mask_ubits_.p, 0, (size_t)nbitwords_ * sizeof(uint32_t), stream_));
  // the cells' summaries follow the observed set alone, and that only grows: as many observed voxels as when they were last
  // built (and no restore or load since) = the same set
  if (mask_obs_count_ != (long long)h_counters_[C_OBSERVED]) {
    hipLaunchKernelGGL(k_obs_cells, dim3(grid_for(nquads, 4, 16384)), dim3(256), 0, stream_, g_, ncx, ncy, ncz,
                       (const uint32_t *)obsbits_, cellobs_.p);
    hipLaunchKernelGGL(k_cell_dist, dim3(grid_for(ncells, 256, 4096)), dim3(256), 0, stream_, ncx, ncy, ncz, (const uint8_t *)cellobs_.p, celldist_.p,
                       cellnb_.p, cellst_.p);
    mask_obs_count_ = (long long)h_counters_[C_OBSERVED];
  }
  hipLaunchKernelGGL(k_eff_occ, dim3(grid_for(nbitwords_, 256, 8192)), dim3(256), 0, stream_, g_, (const uint32_t *)occbits_,
                     (const uint32_t *)obsbits_, effocc_.p, nbitwords_);
  {  // the hidden sites' portals: a table of at least four slots per obstacle
    size_t slots = 4096;
    while (slots < 4 * (size_t)std::max<long long>((long long)h_counters_[C_NOCC], 1)) slots *= 2;
    mask_ptab_.ensure_exact(slots, stream_);
    ma.ptab = reinterpret_cast<uint2 *>(mask_ptab_.p), ma.ptab_mask = (uint32_t)(slots - 1);
    FIESTA_HIP_CHECK(hipMemsetAsync(mask_ptab_.p, 0xFF, slots * sizeof(unsigned long long), stream_));
    hipLaunchKernelGGL(k_portal_sites, dim3(grid_for(nbitwords_, 256, 8192)), dim3(256), 0, stream_, ma, nbitwords_);
  }
  FIESTA_HIP_CHECK(hipGetLastError());
  struct Restore {  // (the transforms read and write the masked transform's buffers only while it runs)
    DenseMap *m;
    ~Restore() { m->tr_occ_ = nullptr, m->tr_out_ = nullptr, m->tr_cellobs_ = nullptr; }
  } restore{this};
  tr_occ_ = effocc_.p, tr_out_ = mask_out_.p, tr_cellobs_ = cellobs_.p;
#ifndef FIESTA_CLASSIFY_BLOCKS
#define FIESTA_CLASSIFY_BLOCKS 4096  /* (persistent waves: 16384 -> 581 us, 8192 -> 436, 4096 -> 393, 2048 -> 392) */
#endif
  const int classify_blocks = (int)std::min<int64_t>(std::max<int64_t>((nquads + 3) / 4, 1), FIESTA_CLASSIFY_BLOCKS);
#ifndef FIESTA_REPAIR_BLOCKS
#define FIESTA_REPAIR_BLOCKS 4096  /* (k_repair_cell on config 2-partial: 1024 -> 121 us, 2048 -> 124, 4096 -> 116, 8192 -> 124; the commit: 27 / 33 / 43 / 51) */
#endif
  const int repair_blocks = (int)std::min<int64_t>(std::max<int64_t>((ncells + 3) / 4, 1), FIESTA_REPAIR_BLOCKS);
  int gi = 0;  // global repair iterations of this update launched so far
  auto launch_chain = [&](const int n, const unsigned long long *failed) {
    MaskArgs a = ma;
    a.failed = failed;
    for (int k = 0; k < n; ++k, ++gi) {
      const uint32_t tag = ++mask_serial_;
      const int rd = (gi + 1) & 1;
      hipLaunchKernelGGL(k_repair_cell, dim3(repair_blocks), dim3(256), 0, stream_, a, k, rd, tag - 1u, gi == 0 ? 1 : 0);
      hipLaunchKernelGGL(k_repair_commit, dim3(std::min(repair_blocks, 1024)), dim3(256), 0, stream_, a, k, rd, tag - 1u, tag, gi == 0 ? 1 : 0);
    }
    FIESTA_HIP_CHECK(hipGetLastError());
  };
  bool cells = false;
  hipEvent_t ev_cert = pool_event(0), ev_rep = pool_event(1), ev_end = pool_event(2);
  int chain = 0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    cells = attempt == 0 && cells_wanted() && run_cells(st, 0, /*publish=*/false);
    if (!cells) {
      bool exact = true;
      if (!run_bulk(st, 0, &exact)) return false;
    }
    const unsigned long long *failed = cells ? &counters_[C_NN_FAILED] : nullptr;
    MaskArgs a = ma;
    a.failed = failed;
    FIESTA_HIP_CHECK(hipEventRecord(ev_cert, stream_));
    hipLaunchKernelGGL(k_mask_classify, dim3(classify_blocks), dim3(256), 0, stream_, a);
    #ifndef FIESTA_WALK_PARTS
#define FIESTA_WALK_PARTS 32  /* (measured on config 2-partial: 8 -> 1.52 ms, 16 -> 1.20, 32 -> 0.98, 64 -> 0.98: short work-groups even out the tail) */
#endif
    hipLaunchKernelGGL(k_mask_walk, dim3(kMaskSegs * FIESTA_WALK_PARTS), dim3(256), 0, stream_, a);
    hipLaunchKernelGGL(k_mask_cells, dim3(std::min(classify_blocks, 1024)), dim3(256), 0, stream_, a);
    FIESTA_HIP_CHECK(hipGetLastError());
    FIESTA_HIP_CHECK(hipEventRecord(ev_rep, stream_));
    gi = 0;
    chain = std::min(std::max(mask_chain_hint_, 2), kMaskIters);
    launch_chain(chain, failed);
    FIESTA_HIP_CHECK(hipEventRecord(ev_end, stream_));
    FIESTA_HIP_CHECK(hipMemcpyAsync(h_mask_ctr_, mask_ctr_, MC_SEG0 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream_));
    if (cells) FIESTA_HIP_CHECK(hipMemcpyAsync(&h_counters_[C_FT_OVF0], &counters_[C_FT_OVF0], 7 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream_));
    FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
#if defined(FIESTA_PROBE)
    fprintf(stderr, "PROBE walks %llu: straight n-sum %llu, steps %llu | paths %llu, positions loaded %llu, refused by cell %llu, n-sum %llu | entries %llu, candidates %llu, none %llu, certified %llu\n",
            h_mask_ctr_[MC_WALKS], h_mask_ctr_[MC_CHANGED0 + 12], h_mask_ctr_[MC_CHANGED0 + 13], h_mask_ctr_[MC_CHANGED0 + 14], h_mask_ctr_[MC_CHANGED0 + 15],
            h_mask_ctr_[MC_CHANGED0 + 16], h_mask_ctr_[MC_CHANGED0 + 17], h_mask_ctr_[MC_CHANGED0 + 18], h_mask_ctr_[MC_CHANGED0 + 19], h_mask_ctr_[MC_CHANGED0 + 20],
            h_mask_ctr_[MC_CHANGED0 + 21]);
#endif
    if (!cells || h_counters_[C_NN_FAILED] == 0) {
      if (cells) nn_fail_streak_ = 0;
      break;
    }
    // a cell without a list: nothing was written, certified or repaired -- the envelope passes provide T
    nn_fail_streak_ = std::min(nn_fail_streak_ + 1, 6);
    nn_skip_ = 4 << nn_fail_streak_;
    const int64_t nfailed = (int64_t)h_counters_[C_NN_FAILED];
    reset_stats_counters(/*lists=*/true);
    if (st) {
      memset(&st->cells, 0, sizeof(st->cells));
      st->nn_failed = nfailed;
    }
  }
  float m1 = 0, m2 = 0, m3 = 0, mc = 0, mr = 0;
  (void)hipEventElapsedTime(&m1, ft_ev_[0], ft_ev_[1]);
  (void)hipEventElapsedTime(&m2, ft_ev_[1], ft_ev_[2]);
  (void)hipEventElapsedTime(&m3, ft_ev_[2], ft_ev_[3]);
  (void)hipEventElapsedTime(&mc, ev_cert, ev_rep);
  (void)hipEventElapsedTime(&mr, ev_rep, ev_end);
  if (st)  // (diagnostics: voxels changed in the first eight repair iterations)
    for (int k = 0; k < 8; ++k) st->prof[k] = (int64_t)h_mask_ctr_[MC_CHANGED0 + k];
  // a segment of the walk list ran out of room: nothing is committed (k_mask_walk and everything behind it returned at once,
  // the field is as it was) and the frontier rounds serve the update
  if (h_mask_ctr_[MC_OVERFLOW] != 0) {
    if (mask_seg_cap_ * kMaskSegs >= (size_t)g.n) return false;  // (cannot be: a voxel is walked once)
    mask_seg_cap_ *= 2;  // a scene with more walks than the list was sized for: once more, with room
    tr_occ_ = nullptr, tr_out_ = nullptr, tr_cellobs_ = nullptr;
    reset_stats_counters(/*lists=*/true);
    return run_masked(st, h0);
  }
  // the repair: further chains until an iteration changes nothing
  int total_iters = 0;
  for (int n = chain;;) {
    int k = 0;
    while (k < n && h_mask_ctr_[MC_CHANGED0 + k] != 0) ++k;
    total_iters += k < n ? k + 1 : n;
    if (k < n || h_mask_ctr_[MC_QUADS] == 0) break;
    // (the chain's last iteration still changed something)
    FIESTA_HIP_CHECK(hipMemsetAsync(&mask_ctr_[MC_CHANGED0], 0, kMaskIters * sizeof(unsigned long long), stream_));
    FIESTA_HIP_CHECK(hipEventRecord(ev_rep, stream_));
    n = 4;
    launch_chain(n, nullptr);
    FIESTA_HIP_CHECK(hipEventRecord(ev_end, stream_));
    FIESTA_HIP_CHECK(hipMemcpyAsync(&h_mask_ctr_[MC_CHANGED0], &mask_ctr_[MC_CHANGED0], kMaskIters * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream_));
    FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
    float more = 0;
    (void)hipEventElapsedTime(&more, ev_rep, ev_end);
    mr += more;
  }
  mask_chain_hint_ = std::min(total_iters + 1, kMaskIters);
  // commit: the side buffer BECOMES the field (every user of the field takes the pointer at call time, in stream order)
  std::swap(coc_, mask_out_.p);
  zero_counters(C_INSERT, 2);  // both queues are drained
  host_counts_[0] = host_counts_[1] = 0;
  queues_zeroed_ = false;
  if (track_) {  // the distance bound of the delete scan: repaired voxels may lie farther than anything the transform wrote
    zero_counter(C_MAXD2);
    hipLaunchKernelGGL(k_maxd2_scan, dim3(grid_for(g_.n, 256, 4096)), dim3(256), 0, stream_, g_, (const vox_t *)coc_, counters_);
    FIESTA_HIP_CHECK(hipGetLastError());
  }
  FIESTA_HIP_CHECK(hipEventRecord(ev1_, stream_));
  FIESTA_HIP_CHECK(hipEventSynchronize(ev1_));
  h_counters_fresh_ = false;
  if (cells) nn_last_ms_ = (double)m1 + m2 + m3, nn_last_nocc_ = (long long)h_counters_[C_NOCC];
  else ft_last_ms_ = (double)m1 + m2 + m3;
  if (st) {
    float ms = 0;
    FIESTA_HIP_CHECK(hipEventElapsedTime(&ms, ev0_, ev1_));
    st->device_ms = ms;
    st->bulk = 1, st->cells = cells ? 1 : 0, st->masked = 1;
    if (cells) {
      st->nn_cells_ms = m1, st->nn_lists_ms = m2, st->nn_fill_ms = m3;
      st->nn_entries = (int64_t)h_counters_[C_NN_ENTRIES];
    } else {
      st->ft_rows_ms = m1, st->ft_plane_ms = m2, st->ft_x_ms = m3;
    }
    st->mask_certify_ms = mc, st->mask_repair_ms = mr;
    st->mask_uncertified = (int64_t)h_mask_ctr_[MC_MARKED], st->mask_iterations = total_iters, st->mask_walks = (int64_t)h_mask_ctr_[MC_WALKS];
    st->mask_quads = (int64_t)h_mask_ctr_[MC_QUADS];
    st->relax_ms = (double)m1 + m2 + m3 + mc + mr;
    st->relax_launches = 3 + 3 + 2 * total_iters;
    st->host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
  }
  return true;
}

// The cost half of the engine choice, from the measured crossover (profiles/r03a_delta_sweep.json: C2's map, both scenes,
// deltas 100 ... 50 k on either engine).  The transform is one fixed sweep, ~6 ps per voxel of the grid; the rounds cost a
// fixed ~0.8 ms of launches and scans on a 512^3 map plus ~0.3 ns per voxel whose obstacle changes, about
// delta x (voxels per obstacle) of them.  In voxel units: bulk pays iff  n <= 1.3e8 + 50 x (estimated updated voxels) --
// on every grid up to 512^3 that is always (measured: 0.80 vs 0.93 ms at a delta of 100), on a 1024^3 shard from a delta of
// ~2 % of the obstacles.  FIESTA_HIP_BULK_RATIO (a fraction of the occupied voxels) overrides the model.
bool DenseMap::bulk_pays(double delta, double nocc, double n) const { return bulk_pays_model(delta, nocc, n, ft_last_ms_, bulk_ratio_); }
// (static: the sharded driver decides from numbers every rank has seen -- the gathered maxima -- never from one rank's own)
bool DenseMap::bulk_pays_model(double delta, double nocc, double n, double ft_last_ms, double bulk_ratio) {
  if (bulk_ratio >= 0) return delta >= bulk_ratio * std::max(nocc, 1.0);
  const double updated = std::min(n, delta * n / std::max(nocc, 1.0));
  // a scene of deep deques (surfaces: the transform takes twice the sweep's nominal time) and a handful of voxels: the
  // rounds win by a few percent (measured: 1.38 vs 1.53 ms at a delta of 100, 2.1 vs 1.5 ms at 500)
  if (delta <= 128 && ft_last_ms > 1.25 * n * 6e-9) return false;
  return n <= 1.3e8 + 50.0 * updated;
}

// After a successful bulk transform: the queues are consumed, timings and counters reported.
void DenseMap::bulk_finish(fiesta_hip_stats *st, std::chrono::steady_clock::time_point h0, bool cells, bool published) {
  static_assert(C_DELETE == C_INSERT + 1, "counter layout");
  nn_clean_ = false;
  if (published) {
    // a cell transform that reports for itself (k_nn_close): ONE synchronisation -- results in h_counters_, the transform's
    // counters clean for the next update, and unless a cell failed the queue lengths too
    FIESTA_HIP_CHECK(hipEventRecord(ev1_, stream_));
    FIESTA_HIP_CHECK(hipEventSynchronize(ev1_));
    published = h_counters_[C_NN_CURSOR] == nn_tag_;  // (always, unless a launch failed)
    if (published) {
      h_counters_fresh_ = false;
      if (h_counters_[C_NN_FAILED] == 0) {  // ... and the queue lengths are cleared too
        queues_zeroed_ = false;
        host_counts_[0] = host_counts_[1] = 0;
        nn_clean_ = true;
      }
    }
  }
  if (!published) {
    if (!queues_zeroed_) zero_counters(C_INSERT, 2);  // both queues are drained
    queues_zeroed_ = false;
    host_counts_[0] = host_counts_[1] = 0;
    if (g_.sharded) zero_counter(C_REMOTE_DEL);
    FIESTA_HIP_CHECK(hipEventRecord(ev1_, stream_));
    collect_stats(nullptr);
    FIESTA_HIP_CHECK(hipEventSynchronize(ev1_));
  }
  float m1 = 0, m2 = 0, m3 = 0;
  const bool timed = hipEventElapsedTime(&m1, ft_ev_[0], ft_ev_[1]) == hipSuccess && hipEventElapsedTime(&m2, ft_ev_[1], ft_ev_[2]) == hipSuccess &&
                     hipEventElapsedTime(&m3, ft_ev_[2], ft_ev_[3]) == hipSuccess;
  if (st) {
    float ms = 0;
    FIESTA_HIP_CHECK(hipEventElapsedTime(&ms, ev0_, ev1_));
    st->device_ms = ms;
    if (cells) {
      st->nn_cells_ms = m1, st->nn_lists_ms = m2, st->nn_fill_ms = m3;
      st->nn_entries = (int64_t)h_counters_[C_NN_ENTRIES];
      st->nn_failed = (int64_t)h_counters_[C_NN_FAILED];
      if (st->nn_incremental) st->nn_dirty_cells = (int64_t)h_counters_[C_NN_DIRTY];
      else st->nn_brute_cells = (int64_t)h_counters_[C_NN_BRUTE];
    } else {
      st->ft_rows_ms = m1, st->ft_plane_ms = m2, st->ft_x_ms = m3;
      st->ft_overflow[0] = (int64_t)h_counters_[C_FT_OVF0], st->ft_overflow[3] = (int64_t)h_counters_[C_FT_OVF0 + 3];
    }
    st->relax_ms = (double)m1 + m2 + m3;
    st->host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
  }
  if (cells && published && !(st && st->nn_incremental)) nn_last_brute_ = (long long)h_counters_[C_NN_BRUTE];
  if (timed) {
    if (cells) {
      if (h_counters_[C_NN_FAILED] == 0) nn_last_ms_ = (double)m1 + m2 + m3, nn_last_nocc_ = (long long)h_counters_[C_NOCC];
    } else {
      ft_last_ms_ = (double)m1 + m2 + m3;
    }
  }
}

// The sharded driver's bulk step (shard_group.hip): transform with `margin`, report exactness; commit consumes the queues.
bool DenseMap::bulk_try(fiesta_hip_stats *st, int margin, bool *exact) {
  ++field_epoch_;  // (the host-side brick cache of the scalar queries is stale from here on)
  use_device();
  nn_clean_ = false;
  nn_valid_ = false;
  ++epoch_;
  FIESTA_HIP_CHECK(hipEventRecord(ev0_, stream_));
  reset_stats_counters();
  if (st) memset(st, 0, sizeof(*st));
  // a sparse obstacle set: the cell transform first, as on an unsharded map (update_esdf).  Its cells decide exactness
  // themselves -- a search window that touches an open face of the region fails its cell -- so a transform without a failed
  // cell is exact whatever the margin; one with failed cells wrote nothing and the envelope passes serve this try.
  tried_cells_ = false;
  if (cells_wanted() && run_cells(st, margin, /*publish=*/false)) {
    FIESTA_HIP_CHECK(hipMemcpyAsync(&h_counters_[C_FT_OVF0], &counters_[C_FT_OVF0], 7 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream_));
    FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));  // (C_NN_CURSOR ... C_NN_ENTRIES and C_FT_MAXD2 lie in this block)
    if (h_counters_[C_NN_FAILED] == 0) {
      nn_fail_streak_ = 0;
      tried_cells_ = true;
      if (exact) *exact = true;
      if (st) st->ft_max_d2 = (int64_t)h_counters_[C_FT_MAXD2];
      return true;
    }
    nn_fail_streak_ = std::min(nn_fail_streak_ + 1, 6);
    nn_skip_ = 4 << nn_fail_streak_;
    const int64_t failed = (int64_t)h_counters_[C_NN_FAILED];
    reset_stats_counters();
    if (st) {
      memset(st, 0, sizeof(*st));
      st->nn_failed = failed;
    }
  }
  return run_bulk(st, margin, exact);
}
void DenseMap::bulk_commit(fiesta_hip_stats *st) {
  ++field_epoch_;  // (the host-side brick cache of the scalar queries is stale from here on)
  use_device();
  // the side buffer BECOMES the field (both hold exactly n words; every user of the field takes the pointer at call time,
  // in stream order behind the transform): a copy would move 4 B per voxel once more
  if (!ft_in_place_) std::swap(coc_, ft_out_.p);
  bulk_finish(st, std::chrono::steady_clock::now(), tried_cells_);
}
void DenseMap::bulk_probe(unsigned long long *ni, unsigned long long *nd, long long *nocc, bool *eligible) {
  use_device();
  FIESTA_HIP_CHECK(hipMemcpyAsync(&h_counters_[C_INSERT], &counters_[C_INSERT], 4 * sizeof(unsigned long long),
                                  hipMemcpyDeviceToHost, stream_));
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
  for (int k = 0; k < 4; ++k) host_counts_[k] = h_counters_[C_INSERT + k];
  host_counts_valid_ = true;
  *ni = h_counters_[C_INSERT], *nd = h_counters_[C_DELETE];
  *nocc = (long long)h_counters_[C_NOCC];
  *eligible = bulk_eligible(*ni, *nd);
}

void DenseMap::level_tuning(int grid_groups, long long spin_limit) {
  if (!lv_) lv_ = new LevelEngine;
  if (grid_groups >= 0) lv_->grid_groups = grid_groups;
  if (spin_limit >= 0) lv_->spin_limit = (uint32_t)spin_limit;
}

int DenseMap::level_trace(uint32_t *out48) const {
  memset(out48, 0, 48 * sizeof(uint32_t));
  if (!lv_ || !lv_->h_ctl) return 0;
  memcpy(out48, lv_->h_ctl->trace, 48 * sizeof(uint32_t));
  return (int)lv_->h_ctl->level;
}

// UpdateESDF by the level engine (level_kernels.hpp).  Seeds: the insert queue, and the orphans the delete scan finds.
// Returns false if the update did not fit the engine's lists: the field then carries frontier tags and the list of active
// tiles is set up for the frontier rounds, which the caller runs.
bool DenseMap::run_levels(fiesta_hip_stats *st, unsigned long long ni, unsigned long long nd) {
  if (!lv_) {
    lv_ = new LevelEngine;
#ifdef FIESTA_HIP_TUNING
    if (const char *e = getenv("FIESTA_HIP_GRID_ENTER")) lv_->grid_enter = (uint32_t)atoi(e);
    if (const char *e = getenv("FIESTA_HIP_GRID_MIN")) lv_->grid_min = (uint32_t)atoi(e);
#endif
  }
  if (!lv_done_) FIESTA_HIP_CHECK(hipEventCreate(&lv_done_));
  // list capacity: what a frame needs is a few thousand entries; a forced level engine gets room for a large update
  const uint32_t want = update_engine_ == 3 ? (uint32_t)std::min<int64_t>(std::max<int64_t>(g_.n / 4, 1 << 20), 1 << 26) : (1u << 20);
  lv_->ensure(want, stream_);
  lv_->begin();
  LevelArgs a = lv_->args(coc_, counters_, track_);
  DenseSpace sp{g_, occbits_, lv_box(g_)};
  TileGrid tg{tx_, ty_, ntx_, nty_, ntz_};
  const bool win_all = g_.wx0 <= 0 && g_.wy0 <= 0 && g_.wz0 <= 0 && g_.wx1 >= g_.nx - 1 && g_.wy1 >= g_.ny - 1 && g_.wz1 >= g_.nz - 1;
  if (ni) {
    hipLaunchKernelGGL((k_level_seed_insert<DenseSpace>), dim3(grid_for((int64_t)ni)), dim3(256), 0, stream_, sp, a,
                       (const uint32_t *)ins_.p, (int64_t)ni);
    FIESTA_HIP_CHECK(hipGetLastError());
  }
  const int bounded = (track_ && nd) ? 1 : 0;
  if (nd) {
    if (bounded) {
      hipLaunchKernelGGL(k_del_bbox, dim3(1), dim3(1024), 0, stream_, g_, (const uint32_t *)del_.p, (int64_t)nd, counters_);
      FIESTA_HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_invalidate<true>, dim3((grid_for(g_.n / 16 + 1, 256, 4096) + 7) / 8 * 8), dim3(256), 0, stream_, g_, tg, coc_,
                       (const uint32_t *)occbits_, (const uint32_t *)gocc_, tile_flag_[0], tile_list_[0], &counters_[C_LIST0],
                       counters_, bounded, a);
    FIESTA_HIP_CHECK(hipGetLastError());
    if (!win_all) {
      hipLaunchKernelGGL((k_level_outside<DenseSpace>), dim3(64), dim3(256), 0, stream_, sp, a);
      FIESTA_HIP_CHECK(hipGetLastError());
    }
    // the delete drain's list walk: the dead cells fill from their rims inwards before level 0 (level_kernels.hpp).  Whole-map
    // updates only: under a partial window the orphans outside it keep asking their in-window neighbours (k_level_outside), and
    // that proxy was fitted to neighbours that wait for their first pull (test_local_sliding_window_mode).
    if (win_all) {
      hipLaunchKernelGGL((k_level_fill<DenseSpace, 1024>), dim3(1), dim3(1024), 0, stream_, sp, a);
      FIESTA_HIP_CHECK(hipGetLastError());
    }
  }
  host_counts_[0] = host_counts_[1] = 0;  // (k_level_run clears the device's queue counters)
  int64_t launches = 0;
  const LevelEngine::Outcome how = lv_->run(sp, a, stream_, lv_done_, update_engine_ == 3, ni + nd <= (unsigned long long)LevelEngine::kTiny, &launches);
  const LevelCtl &c = *lv_->h_ctl;
  if (how == LevelEngine::kDone) {
    h_counters_fresh_ = false;
    if (st) {
      float ms = 0;
      FIESTA_HIP_CHECK(hipEventElapsedTime(&ms, ev0_, lv_done_));
      st->device_ms = ms;
      st->relax_ms = ms;
      st->rounds = (int64_t)c.work;
      st->relax_launches = launches;
      st->voxel_writes = (int64_t)c.writes;
      st->invalidated = (int64_t)c.invalidated;
      st->levels = 1;
      st->grid_levels = (int64_t)c.grid_levels;
      st->prof[0] = (int64_t)c.ticks * 10, st->prof[1] = (int64_t)c.level;  // ns inside k_level_run, levels
      for (int k = 0; k < 4; ++k) st->prof[2 + k] = (int64_t)c.phase[k] * 10;
      st->prof[6] = (int64_t)c.items, st->prof[7] = (int64_t)c.peak;
    }
    return true;
  }
  if (st) st->prof[1] = (int64_t)c.level, st->prof[6] = (int64_t)c.items, st->prof[7] = (int64_t)c.peak;  // (how far the levels got)
  if (how == LevelEngine::kAbort)  // k_level_grid gave up between two phases: both repairs below, the list's first
    FIESTA_HIP_CHECK(hipMemsetAsync(&lv_->ctl->overflow, 0, sizeof(uint32_t), stream_));  // (k_level_pull skips a given-up update)
  if (how == LevelEngine::kHandOver || how == LevelEngine::kAbort) {  // the frontier grew beyond the level engine's reach: its entries become active tiles
    a.level = c.level;  // (phase A of the level the engine stopped in front of: see k_level_list_to_tiles)
    // (work-groups in proportion to the frontier: a delete on a surface orphans 10^5 voxels and level 0 is handed over whole)
    const uint32_t n_over = std::min(c.n[c.level % 3u], lv_->cap);
    hipLaunchKernelGGL((k_level_pull<DenseSpace>), dim3(std::min(std::max(n_over / 64u, 64u), 8192u)), dim3(256), 0, stream_, sp, a);
    hipLaunchKernelGGL((k_level_list_to_tiles<DenseSpace>), dim3(std::min(std::max(n_over / 256u, 16u), 4096u)), dim3(256), 0, stream_, sp, a, tg,
                       tile_flag_[0], tile_list_[0], &counters_[C_LIST0]);
    FIESTA_HIP_CHECK(hipGetLastError());
    if (how == LevelEngine::kHandOver) return false;
  }
  // Overflow.  What the level engine left behind: frontier tags on queued voxels, reset orphans that still carry their
  // dead id, and -- if the scan's own lists overflowed -- orphans outside the window that nobody touched.  The ordinary scan
  // finds the latter (their links are still dead), k_level_to_tiles turns the tags into active tiles.
  if (nd) {
    hipLaunchKernelGGL(k_invalidate<false>, dim3((grid_for(g_.n / 16 + 1, 256, 4096) + 7) / 8 * 8), dim3(256), 0, stream_, g_, tg, coc_,
                       (const uint32_t *)occbits_, (const uint32_t *)gocc_, tile_flag_[0], tile_list_[0], &counters_[C_LIST0],
                       counters_, bounded, LevelArgs{});
    FIESTA_HIP_CHECK(hipGetLastError());
  }
  hipLaunchKernelGGL((k_level_to_tiles<DenseSpace>), dim3(grid_for(g_.n, 256, 8192)), dim3(256), 0, stream_, sp, a, g_.n, tg,
                     tile_flag_[0], tile_list_[0], &counters_[C_LIST0]);
  FIESTA_HIP_CHECK(hipGetLastError());
  return false;
}

void DenseMap::update_esdf(fiesta_hip_stats *st, bool seed_only) {  // UpdateESDF (src/ESDFMap.cpp:273-398)
  use_device();
  notes_ = 0;
  const bool nn_was_clean = nn_clean_;  // (only a cell transform that reports for itself sets it again: bulk_finish)
  nn_clean_ = false;
  const bool nn_was_valid = nn_valid_;  // (set again by a cell transform that succeeds on the map's own field)
  const auto h0 = std::chrono::steady_clock::now();
  if (host_counts_valid_) {  // (what UpdateOccupancy read last: nothing else changes these four)
    for (int k = 0; k < 4; ++k) h_counters_[C_INSERT + k] = host_counts_[k];
  } else {
    static_assert(C_LATE > C_NOCC, "counter layout");
    FIESTA_HIP_CHECK(hipMemcpyAsync(&h_counters_[C_INSERT], &counters_[C_INSERT], (C_LATE - C_INSERT + 1) * sizeof(unsigned long long),
                                    hipMemcpyDeviceToHost, stream_));  // C_INSERT, C_DELETE, C_OBSERVED, C_NOCC ... C_LATE
    FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
  }
  const unsigned long long ni = h_counters_[C_INSERT], nd = h_counters_[C_DELETE];
  const bool remote_del = g_.sharded && read_counter(C_REMOTE_DEL) != 0;
  if (st) {
    memset(st, 0, sizeof(*st));
    st->inserted = (int64_t)ni;
    st->deleted = (int64_t)nd;
    st->observed_voxels = (int64_t)h_counters_[C_OBSERVED];
    st->occupied_voxels = (int64_t)h_counters_[C_NOCC];
  }
  if (ni == 0 && nd == 0 && !seed_only) {
    if (st) st->host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
    return;
  }
  nn_valid_ = false;
  ++field_epoch_;  // (the host-side brick cache of the scalar queries is stale from here on; an update without work leaves it valid)
  ++epoch_;
  TileGrid tg{tx_, ty_, ntx_, nty_, ntz_};
  FIESTA_HIP_CHECK(hipEventRecord(ev0_, stream_));
  {  // why this update will take the path it takes (fiesta_hip_stats.path_notes): the standing conditions here, the rest as they bite
    const long long owned = (long long)(g_.ox1 - g_.ox0 + 1) * (g_.oy1 - g_.oy0 + 1) * (g_.oz1 - g_.oz0 + 1);
    if ((long long)h_counters_[C_OBSERVED] != owned) notes_ |= FIESTA_HIP_NOTE_PARTLY_OBSERVED;
    if (g_.sharded) notes_ |= FIESTA_HIP_NOTE_SHARDED;
    if (g_.wrap) notes_ |= FIESTA_HIP_NOTE_ID_WRAP;
    if (update_engine_ != 0) notes_ |= FIESTA_HIP_NOTE_ENGINE_PINNED;
  }
  const bool full_window = g_.wx0 <= 0 && g_.wy0 <= 0 && g_.wz0 <= 0 && g_.wx1 >= g_.nx - 1 && g_.wy1 >= g_.ny - 1 && g_.wz1 >= g_.nz - 1;
  // (a map that held no obstacle before this update has no history: cleared BEFORE this update's own window is recorded --
  //  the first obstacles inserted under a partial window on an empty map must leave the flag set)
  if (win_dirty_ && !g_.sharded && (long long)h_counters_[C_NOCC] - (long long)ni + (long long)nd <= 0) win_dirty_ = false;
  if (!full_window) win_dirty_ = true, notes_ |= FIESTA_HIP_NOTE_PARTIAL_WINDOW;  // (see bulk_eligible)
  else if (win_dirty_) notes_ |= FIESTA_HIP_NOTE_WINDOW_HISTORY;
  // Engine choice.  The bulk transform costs one fixed sweep over the grid; the frontier rounds cost in proportion to
  // the voxels whose closest obstacle changes, roughly (inserts + deletes) x (grid / occupied voxels).  The level engine
  // (level_kernels.hpp) takes every update of a few thousand voxels -- a sensor frame -- that the transform does not, and,
  // if asked for (update_engine 3), every update its lists can hold; one that outgrows them is finished by the rounds.
  bool levels_fell_back = false;
  // On a map the transform's gate is open for -- every voxel observed, no history of partial windows -- the reference's
  // field is the exact transform whatever the order (section 3c): the level engine has nothing to add there, and a delete on
  // such a map orphans a whole Voronoi cell (10^5 voxels behind a surface), which is the rounds' or the transform's work.
  const bool gate_open = !seed_only && !g_.sharded && bulk_eligible(ni, nd);
  // (the inserts ARE level 0: more of them than one work-group carries and the level engine would only hand the update on)
  const bool try_levels = !seed_only && !g_.sharded && !g_.wrap && update_engine_ != 1 &&
                          (update_engine_ == 3 || (!gate_open && ni + nd <= (unsigned long long)small_update_ && ni <= (unsigned long long)LevelEngine::kInsertCap));
  const bool try_bulk = gate_open && (bulk_pinned() || bulk_pays((double)(ni + nd), (double)(long long)h_counters_[C_NOCC], (double)g_.n));
  if (gate_open && !try_bulk) notes_ |= FIESTA_HIP_NOTE_SMALL_DELTA;
  if (!gate_open && !seed_only && (long long)h_counters_[C_OBSERVED] < g_.n && !bulk_pinned() &&
      !bulk_pays((double)(ni + nd), (double)(long long)h_counters_[C_NOCC], (double)g_.n))
    notes_ |= FIESTA_HIP_NOTE_SMALL_DELTA;
  bool counters_reset = false;
  // A partially observed map (every map a sensor builds): a delta too large for the level engine goes to the masked transform
  // (mask_kernels.hpp) where the map's history allows it -- a fixed sweep like the other transforms -- instead of the rounds.
  if (!seed_only && !gate_open && !try_bulk && (long long)h_counters_[C_OBSERVED] < g_.n &&
      (update_engine_ == 6 || (!try_levels && (update_engine_ == 0 || bulk_pinned()) &&
                               bulk_pays((double)(ni + nd), (double)(long long)h_counters_[C_NOCC], (double)g_.n))) &&
      masked_eligible(ni, nd)) {
    reset_stats_counters(/*lists=*/true);
    counters_reset = true;
    if (run_masked(st, h0)) return;
    notes_ |= FIESTA_HIP_NOTE_MASKED_GAVE_UP;
    if (st) {
      const fiesta_hip_stats keep = *st;
      memset(st, 0, sizeof(*st));
      st->inserted = keep.inserted, st->deleted = keep.deleted, st->observed_voxels = keep.observed_voxels, st->occupied_voxels = keep.occupied_voxels;
    }
    FIESTA_HIP_CHECK(hipEventRecord(ev0_, stream_));
    reset_stats_counters(/*lists=*/true);
  }
  if (try_bulk) {
    // (an unsharded map's transform cannot fail to serve the update -- the envelope passes stand behind the cell transform --
    //  so the two queue lengths go in the same launch as the statistics)
    // a sparse obstacle set: the cell transform first (nn_kernels.hpp).  A cell without a list fails it -- k_nn_fill then
    // wrote nothing -- and the envelope passes below serve the update; the obstacle count is remembered and not retried.
    const bool want_cells = cells_wanted();
    // (the last update was a cell transform that cleaned up behind itself and nothing has touched the counters since: no
    //  reset launch ahead of this one -- its fill will clear the queue lengths at its end)
    const bool lean = want_cells && nn_was_clean;
    if (lean) {
      queues_zeroed_ = false;
      ft_counters_clean_ = true;
    } else {
      queues_zeroed_ = !g_.sharded && g_.nx <= 2048 && g_.ny <= 2048 && g_.nz <= 2048;
      reset_stats_counters(/*lists=*/true, queues_zeroed_);
    }
    counters_reset = true;
    // the lists of the last update are still valid and little has changed: only the cells a changed voxel can reach are redone
    // (a voxel dirties the ~(2 x reach + 1)^3 cells around it: worth it while that is a fraction of the map)
    const int64_t ncells8 = (int64_t)((g_.nx + 7) / 8) * ((g_.ny + 7) / 8) * ((g_.nz + 7) / 8);
    if (want_cells && nn_was_valid && !g_.wrap && (int64_t)(ni + nd) * 180 <= ncells8 / 2 && run_cells(st, 0, /*publish=*/true, /*incremental=*/true, ni, nd)) {
      bulk_finish(st, h0, /*cells=*/true, /*published=*/true);
      if (h_counters_[C_NN_FAILED] == 0) {
        nn_fail_streak_ = 0;
        nn_valid_ = true;
        return;
      }
      // more dirty cells than the list holds, or a cell that lost its last obstacle in reach: the full transform, from scratch
      notes_ |= FIESTA_HIP_NOTE_INCREMENTAL_REDONE;
      FIESTA_HIP_CHECK(hipMemsetAsync(nn_dirty_flag_.p, 0, nn_dirty_flag_.cap * sizeof(uint32_t), stream_));
      if (st) memset(&st->cells, 0, sizeof(st->cells)), st->nn_incremental = 0;
      FIESTA_HIP_CHECK(hipEventRecord(ev0_, stream_));
      reset_stats_counters(/*lists=*/true);
    }
    if (want_cells && run_cells(st, 0, /*publish=*/true)) {
      bulk_finish(st, h0, /*cells=*/true, /*published=*/true);
      if (h_counters_[C_NN_FAILED] == 0) {
        nn_fail_streak_ = 0;
        nn_valid_ = !g_.sharded;
        return;
      }
      nn_fail_streak_ = std::min(nn_fail_streak_ + 1, 6);
      nn_skip_ = 4 << nn_fail_streak_;
      notes_ |= FIESTA_HIP_NOTE_CELLS_FAILED;
      const int64_t failed = (int64_t)h_counters_[C_NN_FAILED];
      if (st) {
        memset(&st->cells, 0, sizeof(st->cells));
        st->nn_failed = failed;
      }
      FIESTA_HIP_CHECK(hipEventRecord(ev0_, stream_));
      reset_stats_counters(/*lists=*/true);
    }
    bool exact = true;
    if (run_bulk(st, 0, &exact)) {
      bulk_finish(st, h0);
      return;
    }
    if (queues_zeroed_) {  // (cannot happen: see above -- but the engines below read the queue lengths on the device)
      h_counters_[C_INSERT] = ni, h_counters_[C_DELETE] = nd;
      FIESTA_HIP_CHECK(hipMemcpyAsync(&counters_[C_INSERT], &h_counters_[C_INSERT], 2 * sizeof(unsigned long long), hipMemcpyHostToDevice, stream_));
      queues_zeroed_ = false;
    }
  }
  if (try_levels) {  // (the level engine keeps its statistics in its own control block: no counter reset on its path)
    if (run_levels(st, ni, nd)) {
      if (st) st->host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
      return;
    }
    levels_fell_back = true;
    notes_ |= FIESTA_HIP_NOTE_LEVELS_GAVE_UP;
    zero_counters(C_INVALIDATED, C_COUNT - C_INVALIDATED);  // (the tile list is set up: its counters stay)
  } else if (!counters_reset) {
    reset_stats_counters(/*lists=*/true);
  }
  if (ni && !levels_fell_back) {
    hipLaunchKernelGGL(k_seed_insert, dim3(grid_for((int64_t)ni)), dim3(256), 0, stream_, g_, tg,
                       (const uint32_t *)ins_.p, (int64_t)ni, coc_, (const uint32_t *)occbits_, tile_flag_[0],
                       tile_list_[0], &counters_[C_LIST0]);
    FIESTA_HIP_CHECK(hipGetLastError());
  }
  if ((nd || remote_del) && !levels_fell_back) {  // (a fall-back: the level engine's scan reset the orphans, run_levels set up their tiles)
    // the scan is bounded by (delete queue's box) + (largest stored distance) where that bound is tracked
    // (enable_distance_tracking)
    const int bounded = (track_ && nd) ? 1 : 0;
    if (bounded) {
      hipLaunchKernelGGL(k_del_bbox, dim3(1), dim3(1024), 0, stream_, g_, (const uint32_t *)del_.p, (int64_t)nd, counters_);
      FIESTA_HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_invalidate<false>, dim3((grid_for(g_.n / 16 + 1, 256, 4096) + 7) / 8 * 8), dim3(256), 0, stream_, g_, tg, coc_,
                       (const uint32_t *)occbits_, (const uint32_t *)gocc_, tile_flag_[0], tile_list_[0],
                       &counters_[C_LIST0], counters_, bounded, LevelArgs{});
    FIESTA_HIP_CHECK(hipGetLastError());
  }
  static_assert(C_DELETE == C_INSERT + 1, "counter layout");
  zero_counters(C_INSERT, 2);  // both queues are drained
  host_counts_[0] = host_counts_[1] = 0;
  if (g_.sharded) zero_counter(C_REMOTE_DEL);
  if (seed_only) {  // sharded driver: ghost exchange comes next, then relax_pending()
    FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
    if (st) st->host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
    return;
  }
  // a small delta (a depth frame): do not even read how many tiles were seeded, the chain of rounds finds out on the device
  const uint32_t n0 = (ni + nd <= (unsigned long long)small_update_ && !remote_del && !levels_fell_back) ? kCountOnDevice : (uint32_t)read_counter(C_LIST0);
  run_rounds(st, n0, 0);
  bool reseeded = false;
  {  // orphans outside the update window (local / sliding-window maps): their pull, after the rounds
    const Geom &g = g_;
    const bool win_all = g.wx0 <= 0 && g.wy0 <= 0 && g.wz0 <= 0 && g.wx1 >= g.nx - 1 && g.wy1 >= g.ny - 1 && g.wz1 >= g.nz - 1;
    if (!win_all && (nd || remote_del)) {
      reseeded = true;
      hipLaunchKernelGGL(k_reseed_outside, dim3(grid_for(g_.n, 256, 8192)), dim3(256), 0, stream_, g_, coc_, (const uint32_t *)occbits_,
                         (const uint32_t *)gocc_, (const unsigned long long *)counters_, (track_ && nd) ? 1 : 0);
      FIESTA_HIP_CHECK(hipGetLastError());
    }
  }
  // (an update that ended with a chain of rounds is already synchronised and has its counters: its last event is the end)
  hipEvent_t end = ev1_;
  if (h_counters_fresh_ && !reseeded && last_chain_event_) {
    end = last_chain_event_;
    collect_stats(st);
  } else {
    FIESTA_HIP_CHECK(hipEventRecord(ev1_, stream_));
    collect_stats(st);
    FIESTA_HIP_CHECK(hipEventSynchronize(ev1_));
  }
  if (st) {
    float ms = 0;
    FIESTA_HIP_CHECK(hipEventElapsedTime(&ms, ev0_, end));
    st->device_ms = ms;
    st->host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
  }
}

void DenseMap::relax_pending(fiesta_hip_stats *st, int64_t *pending) {
  ++field_epoch_;  // (the host-side brick cache of the scalar queries is stale from here on)
  use_device();
  if (st) memset(st, 0, sizeof(*st));
  reset_stats_counters();
  const uint32_t n0 = (uint32_t)read_counter(C_LIST0);
  if (pending) *pending = n0;
  FIESTA_HIP_CHECK(hipEventRecord(ev0_, stream_));
  run_rounds(st, n0, 0);
  zero_counter(C_LIST0);
  FIESTA_HIP_CHECK(hipEventRecord(ev1_, stream_));
  collect_stats(st);
  FIESTA_HIP_CHECK(hipEventSynchronize(ev1_));
  if (st) {
    float ms = 0;
    FIESTA_HIP_CHECK(hipEventElapsedTime(&ms, ev0_, ev1_));
    st->device_ms = ms;
  }
}

// ---- queries ----
void DenseMap::get_distance_vox(const int32_t *vox, int64_t n, double *out) {
  if (n <= 0) return;
  if (n <= kHostQueries) {  // a scalar call of the drop-in class: the host-side brick cache
    HostWords wd{this};
    for (int64_t i = 0; i < n; ++i) out[i] = vox_distance(g_, wd, vox[3 * i] - g_.gx0, vox[3 * i + 1] - g_.gy0, vox[3 * i + 2] - g_.gz0);
    return;
  }
  use_device();
  stage_a_.ensure(n * 3 * sizeof(int32_t), stream_);
  stage_c_.ensure(n * sizeof(double), stream_);
  FIESTA_HIP_CHECK(hipMemcpyAsync(stage_a_.p, vox, n * 3 * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
  hipLaunchKernelGGL(k_query_dist_vox, dim3(grid_for(n)), dim3(256), 0, stream_, g_, (const vox_t *)coc_,
                     (const int32_t *)stage_a_.p, n, (double *)stage_c_.p);
  FIESTA_HIP_CHECK(hipMemcpyAsync(out, stage_c_.p, n * sizeof(double), hipMemcpyDeviceToHost, stream_));
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
}
void DenseMap::get_distance_pos(const double *pos, int64_t n, double *out) {
  if (n <= 0) return;
  if (n <= kHostQueries) {
    HostWords wd{this};
    for (int64_t i = 0; i < n; ++i) out[i] = query_dist_pos(g_, wd, pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
    return;
  }
  use_device();
  stage_a_.ensure(n * 3 * sizeof(double), stream_);
  stage_c_.ensure(n * sizeof(double), stream_);
  FIESTA_HIP_CHECK(hipMemcpyAsync(stage_a_.p, pos, n * 3 * sizeof(double), hipMemcpyHostToDevice, stream_));
  hipLaunchKernelGGL(k_query_dist_pos, dim3(grid_for(n)), dim3(256), 0, stream_, g_, (const vox_t *)coc_,
                     (const double *)stage_a_.p, n, (double *)stage_c_.p);
  FIESTA_HIP_CHECK(hipMemcpyAsync(out, stage_c_.p, n * sizeof(double), hipMemcpyDeviceToHost, stream_));
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
}
void DenseMap::get_dist_grad(const double *pos, int64_t n, double *dist, double *grad, bool dev) {
  if (n <= 0) return;
  if (!dev && n <= kHostQueries) {
    HostWords wd{this};
    for (int64_t i = 0; i < n; ++i) dist[i] = query_trilinear(g_, wd, pos + 3 * i, grad ? grad + 3 * i : nullptr);
    return;
  }
  use_device();
  if (dev) {
    hipLaunchKernelGGL(k_query_trilinear, dim3(grid_for(n)), dim3(256), 0, stream_, g_, (const vox_t *)coc_, pos, n,
                       dist, grad);
    FIESTA_HIP_CHECK(hipGetLastError());
    return;
  }
  stage_a_.ensure(n * 3 * sizeof(double), stream_);
  stage_b_.ensure(n * 3 * sizeof(double), stream_);
  stage_c_.ensure(n * sizeof(double), stream_);
  FIESTA_HIP_CHECK(hipMemcpyAsync(stage_a_.p, pos, n * 3 * sizeof(double), hipMemcpyHostToDevice, stream_));
  hipLaunchKernelGGL(k_query_trilinear, dim3(grid_for(n)), dim3(256), 0, stream_, g_, (const vox_t *)coc_,
                     (const double *)stage_a_.p, n, (double *)stage_c_.p, (double *)stage_b_.p);
  FIESTA_HIP_CHECK(hipMemcpyAsync(dist, stage_c_.p, n * sizeof(double), hipMemcpyDeviceToHost, stream_));
  if (grad) FIESTA_HIP_CHECK(hipMemcpyAsync(grad, stage_b_.p, n * 3 * sizeof(double), hipMemcpyDeviceToHost, stream_));
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
}
void DenseMap::get_occupancy_vox(const int32_t *vox, int64_t n, int32_t *out) {
  if (n <= 0) return;
  if (n <= kHostQueries) {
    for (int64_t i = 0; i < n; ++i) {
      const int x = vox[3 * i] - g_.gx0, y = vox[3 * i + 1] - g_.gy0, z = vox[3 * i + 2] - g_.gz0;
      out[i] = g_.in_grid(x, y, z) ? host_occ(x, y, z) : 0;
    }
    return;
  }
  use_device();
  stage_a_.ensure(n * 3 * sizeof(int32_t), stream_);
  stage_c_.ensure(n * sizeof(int32_t), stream_);
  FIESTA_HIP_CHECK(hipMemcpyAsync(stage_a_.p, vox, n * 3 * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
  hipLaunchKernelGGL(k_query_occ_vox, dim3(grid_for(n)), dim3(256), 0, stream_, g_, (const uint32_t *)occbits_,
                     (const int32_t *)stage_a_.p, n, (int32_t *)stage_c_.p);
  FIESTA_HIP_CHECK(hipMemcpyAsync(out, stage_c_.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
}
void DenseMap::get_occupancy_pos(const double *pos, int64_t n, int32_t *out) {
  if (n <= 0) return;
  if (n <= kHostQueries) {
    for (int64_t i = 0; i < n; ++i) {
      const double px = pos[3 * i], py = pos[3 * i + 1], pz = pos[3 * i + 2];
      if (!pos_in_map(g_, px, py, pz)) {
        out[i] = FIESTA_HIP_UNDEFINED;
        continue;
      }
      const int x = (int)floor((px - g_.org[0]) / g_.res) - g_.gx0, y = (int)floor((py - g_.org[1]) / g_.res) - g_.gy0,
                z = (int)floor((pz - g_.org[2]) / g_.res) - g_.gz0;
      out[i] = g_.in_grid(x, y, z) ? host_occ(x, y, z) : 0;
    }
    return;
  }
  use_device();
  stage_a_.ensure(n * 3 * sizeof(double), stream_);
  stage_c_.ensure(n * sizeof(int32_t), stream_);
  FIESTA_HIP_CHECK(hipMemcpyAsync(stage_a_.p, pos, n * 3 * sizeof(double), hipMemcpyHostToDevice, stream_));
  hipLaunchKernelGGL(k_query_occ_pos, dim3(grid_for(n)), dim3(256), 0, stream_, g_, (const uint32_t *)occbits_,
                     (const double *)stage_a_.p, n, (int32_t *)stage_c_.p);
  FIESTA_HIP_CHECK(hipMemcpyAsync(out, stage_c_.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
}

void DenseMap::download_field(int32_t *d2, int32_t *coc, uint8_t *occ, double *logodds) {
  use_device();
  const int64_t n = g_.n;
  int32_t *dd2 = nullptr, *dc = nullptr;
  uint8_t *docc = nullptr;
  if (d2) {
    stage_a_.ensure(n * sizeof(int32_t), stream_);
    dd2 = (int32_t *)stage_a_.p;
  }
  if (coc) {
    stage_b_.ensure(n * 3 * sizeof(int32_t), stream_);
    dc = (int32_t *)stage_b_.p;
  }
  if (occ) {
    stage_c_.ensure(n, stream_);
    docc = (uint8_t *)stage_c_.p;
  }
  if (d2 || coc || occ) {
    hipLaunchKernelGGL(k_export, dim3(grid_for(n, 256, 8192)), dim3(256), 0, stream_, g_, (const vox_t *)coc_,
                       (const uint32_t *)occbits_, dd2, dc, docc);
    FIESTA_HIP_CHECK(hipGetLastError());
  }
  if (d2) FIESTA_HIP_CHECK(hipMemcpyAsync(d2, dd2, n * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
  if (coc) FIESTA_HIP_CHECK(hipMemcpyAsync(coc, dc, n * 3 * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
  if (occ) FIESTA_HIP_CHECK(hipMemcpyAsync(occ, docc, n, hipMemcpyDeviceToHost, stream_));
  if (logodds) FIESTA_HIP_CHECK(hipMemcpyAsync(logodds, logodds_, n * sizeof(double), hipMemcpyDeviceToHost, stream_));
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
}

// Observed voxels of the owned box that hold no obstacle (distance +10000): freshly observed ones that wait for a wave,
// everything while the map is empty -- and, on grids beyond 1024 voxels per axis, whatever lies farther than the 512
// voxels an id reaches from every obstacle (common.hpp: kD2Cap).
int64_t DenseMap::count_no_obstacle() {
  use_device();
  zero_counter(C_SCRATCH);
  hipLaunchKernelGGL(k_count_stale, dim3(grid_for(g_.n, 256, 4096)), dim3(256), 0, stream_, g_, (const vox_t *)coc_, &counters_[C_SCRATCH]);
  FIESTA_HIP_CHECK(hipGetLastError());
  return (int64_t)read_counter(C_SCRATCH);
}

int64_t DenseMap::occupied_voxels(int32_t *vox, int64_t cap) {
  use_device();
  zero_counter(C_SCRATCH);
  const int64_t nwords = nbitwords_;
  int32_t *dout = nullptr;
  if (vox && cap > 0) {
    stage_a_.ensure((size_t)cap * 3 * sizeof(int32_t), stream_);
    dout = (int32_t *)stage_a_.p;
  }
  hipLaunchKernelGGL(k_occupied_list, dim3(grid_for(nwords, 256, 8192)), dim3(256), 0, stream_, g_, (const uint32_t *)occbits_,
                     dout, (unsigned long long)(dout ? cap : 0), &counters_[C_SCRATCH]);
  FIESTA_HIP_CHECK(hipGetLastError());
  const int64_t n = (int64_t)read_counter(C_SCRATCH);
  if (dout && n) FIESTA_HIP_CHECK(hipMemcpyAsync(vox, dout, (size_t)std::min(n, cap) * 3 * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
  return n;
}

int64_t DenseMap::point_cloud(int vis_lower_bound, int vis_upper_bound, float *xyz, int64_t cap) {
  use_device();
  zero_counter(C_SCRATCH);
  float *dout = nullptr;
  if (xyz && cap > 0) {
    stage_a_.ensure((size_t)cap * 3 * sizeof(float), stream_);
    dout = (float *)stage_a_.p;
  }
  hipLaunchKernelGGL(k_point_cloud, dim3(grid_for(nbitwords_, 256, 8192)), dim3(256), 0, stream_, g_, (const uint32_t *)occbits_,
                     vis_lower_bound, vis_upper_bound, dout, (unsigned long long)(dout ? cap : 0), &counters_[C_SCRATCH]);
  FIESTA_HIP_CHECK(hipGetLastError());
  const int64_t n = (int64_t)read_counter(C_SCRATCH);
  if (dout && n) FIESTA_HIP_CHECK(hipMemcpyAsync(xyz, dout, (size_t)std::min(n, cap) * 3 * sizeof(float), hipMemcpyDeviceToHost, stream_));
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
  return n;
}

int64_t DenseMap::slice_marker(int slice, double max_dist, double *xyz, float *rgba, int64_t cap) {
  use_device();
  const int z = slice - g_.gz0;
  if (z < 0 || z >= g_.nz) throw Error(FIESTA_HIP_ERR_INVALID, "slice outside the grid");
  zero_counter(C_SCRATCH);
  double *dx = nullptr;
  float *dc = nullptr;
  if (xyz && rgba && cap > 0) {
    stage_a_.ensure((size_t)cap * 3 * sizeof(double), stream_);
    stage_b_.ensure((size_t)cap * 4 * sizeof(float), stream_);
    dx = (double *)stage_a_.p, dc = (float *)stage_b_.p;
  }
  const int64_t cells = (int64_t)g_.nx * g_.ny;
  hipLaunchKernelGGL(k_slice_marker, dim3(grid_for(cells, 256, 4096)), dim3(256), 0, stream_, g_, (const vox_t *)coc_, z, max_dist, dx,
                     dc, (unsigned long long)(dx ? cap : 0), &counters_[C_SCRATCH]);
  FIESTA_HIP_CHECK(hipGetLastError());
  const int64_t n = (int64_t)read_counter(C_SCRATCH);
  if (dx && n) {
    const size_t k = (size_t)std::min(n, cap);
    FIESTA_HIP_CHECK(hipMemcpyAsync(xyz, dx, k * 3 * sizeof(double), hipMemcpyDeviceToHost, stream_));
    FIESTA_HIP_CHECK(hipMemcpyAsync(rgba, dc, k * 4 * sizeof(float), hipMemcpyDeviceToHost, stream_));
  }
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
  return n;
}

void DenseMap::slice_distances(int z_vox, double *out) {
  use_device();
  const int z = z_vox - g_.gz0;
  if (z < 0 || z >= g_.nz) throw Error(FIESTA_HIP_ERR_INVALID, "slice outside the grid");
  const int64_t n = (int64_t)g_.nx * g_.ny;
  stage_c_.ensure((size_t)n * sizeof(double), stream_);
  hipLaunchKernelGGL(k_slice, dim3(grid_for(n, 256, 4096)), dim3(256), 0, stream_, g_, (const vox_t *)coc_, z, (double *)stage_c_.p);
  FIESTA_HIP_CHECK(hipGetLastError());
  FIESTA_HIP_CHECK(hipMemcpyAsync(out, stage_c_.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, stream_));
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
}

void DenseMap::download_counts(int32_t *num_hit, int32_t *num_miss) {
  use_device();
  std::vector<unsigned long long> h(g_.n);
  FIESTA_HIP_CHECK(hipMemcpyAsync(h.data(), cnt_, g_.n * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream_));
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
  for (int64_t i = 0; i < g_.n; ++i) {
    if (num_hit) num_hit[i] = (int32_t)(h[i] >> 32);
    if (num_miss) num_miss[i] = (int32_t)(uint32_t)h[i];
  }
}

// ---- snapshots ----
void DenseMap::snapshot_save(int slot) {
  use_device();
  if (slot < 0 || slot >= 4) throw Error(FIESTA_HIP_ERR_INVALID, "snapshot slot out of range");
  Snapshot &s = snaps_[slot];
  const int64_t n = g_.n;
  FIESTA_HIP_CHECK(hipMemcpyAsync(h_counters_, counters_, C_COUNT * sizeof(unsigned long long),
                                  hipMemcpyDeviceToHost, stream_));
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
  memcpy(s.counters, h_counters_, sizeof(s.counters));
  s.coc.ensure(n, stream_);
  s.logodds.ensure(n, stream_);
  s.cnt.ensure(n, stream_);
  s.occbits.ensure(nbitwords_, stream_);
  FIESTA_HIP_CHECK(hipMemcpyAsync(s.coc.p, coc_, n * sizeof(vox_t), hipMemcpyDeviceToDevice, stream_));
  FIESTA_HIP_CHECK(hipMemcpyAsync(s.logodds.p, logodds_, n * sizeof(double), hipMemcpyDeviceToDevice, stream_));
  FIESTA_HIP_CHECK(hipMemcpyAsync(s.cnt.p, cnt_, n * sizeof(unsigned long long), hipMemcpyDeviceToDevice, stream_));
  FIESTA_HIP_CHECK(hipMemcpyAsync(s.occbits.p, occbits_, nbitwords_ * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream_));
  s.obsbits.ensure(nbitwords_, stream_);
  s.latebits.ensure(nbitwords_, stream_);
  FIESTA_HIP_CHECK(hipMemcpyAsync(s.obsbits.p, obsbits_, nbitwords_ * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream_));
  FIESTA_HIP_CHECK(hipMemcpyAsync(s.latebits.p, latebits_, nbitwords_ * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream_));
  if (gocc_) {
    s.gocc.ensure(ngoccwords_, stream_);
    FIESTA_HIP_CHECK(hipMemcpyAsync(s.gocc.p, gocc_, ngoccwords_ * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream_));
  }
  const size_t nt = s.counters[C_TOUCHED], ni = s.counters[C_INSERT], nd = s.counters[C_DELETE];
  if (nt) {
    s.touched.ensure(nt, stream_);
    FIESTA_HIP_CHECK(hipMemcpyAsync(s.touched.p, touched_.p, nt * 4, hipMemcpyDeviceToDevice, stream_));
  }
  if (ni) {
    s.ins.ensure(ni, stream_);
    FIESTA_HIP_CHECK(hipMemcpyAsync(s.ins.p, ins_.p, ni * 4, hipMemcpyDeviceToDevice, stream_));
  }
  if (nd) {
    s.del.ensure(nd, stream_);
    FIESTA_HIP_CHECK(hipMemcpyAsync(s.del.p, del_.p, nd * 4, hipMemcpyDeviceToDevice, stream_));
  }
  s.g = g_;
  s.stale_inf = stale_inf_;
  s.win_dirty = win_dirty_;
  s.valid = true;
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
}

void DenseMap::snapshot_restore(int slot) {
  ++field_epoch_;  // (the host-side brick cache of the scalar queries is stale from here on)
  nn_clean_ = false;  // (the counters come back as they were saved)
  nn_valid_ = false;
  mask_obs_count_ = -1;
  use_device();
  if (slot < 0 || slot >= 4 || !snaps_[slot].valid) throw Error(FIESTA_HIP_ERR_STATE, "no such snapshot");
  Snapshot &s = snaps_[slot];
  const int64_t n = g_.n;
  FIESTA_HIP_CHECK(hipMemcpyAsync(coc_, s.coc.p, n * sizeof(vox_t), hipMemcpyDeviceToDevice, stream_));
  FIESTA_HIP_CHECK(hipMemcpyAsync(logodds_, s.logodds.p, n * sizeof(double), hipMemcpyDeviceToDevice, stream_));
  FIESTA_HIP_CHECK(hipMemcpyAsync(cnt_, s.cnt.p, n * sizeof(unsigned long long), hipMemcpyDeviceToDevice, stream_));
  FIESTA_HIP_CHECK(hipMemcpyAsync(occbits_, s.occbits.p, nbitwords_ * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream_));
  FIESTA_HIP_CHECK(hipMemcpyAsync(obsbits_, s.obsbits.p, nbitwords_ * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream_));
  FIESTA_HIP_CHECK(hipMemcpyAsync(latebits_, s.latebits.p, nbitwords_ * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream_));
  if (gocc_) FIESTA_HIP_CHECK(hipMemcpyAsync(gocc_, s.gocc.p, ngoccwords_ * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream_));
  const size_t nt = s.counters[C_TOUCHED], ni = s.counters[C_INSERT], nd = s.counters[C_DELETE];
  if (nt) {
    touched_.ensure(nt, stream_);
    FIESTA_HIP_CHECK(hipMemcpyAsync(touched_.p, s.touched.p, nt * 4, hipMemcpyDeviceToDevice, stream_));
  }
  if (ni) {
    ins_.ensure(ni, stream_);
    FIESTA_HIP_CHECK(hipMemcpyAsync(ins_.p, s.ins.p, ni * 4, hipMemcpyDeviceToDevice, stream_));
  }
  if (nd) {
    del_.ensure(nd, stream_);
    FIESTA_HIP_CHECK(hipMemcpyAsync(del_.p, s.del.p, nd * 4, hipMemcpyDeviceToDevice, stream_));
  }
  unsigned long long c[C_COUNT];
  memcpy(c, s.counters, sizeof(c));
  c[C_LIST0] = c[C_LIST1] = c[C_LIST2] = 0;
  memcpy(h_counters_, c, sizeof(c));
  FIESTA_HIP_CHECK(hipMemcpyAsync(counters_, h_counters_, sizeof(c), hipMemcpyHostToDevice, stream_));
  touched_upper_ = (int64_t)nt;
  g_ = s.g;
  stale_inf_ = s.stale_inf;
  win_dirty_ = s.win_dirty;
  host_counts_valid_ = false;
  if (track_) {  // the snapshot may predate the tracking: recompute the distance bound for the restored field
    zero_counter(C_MAXD2);
    hipLaunchKernelGGL(k_maxd2_scan, dim3(grid_for(g_.n, 256, 4096)), dim3(256), 0, stream_, g_, (const vox_t *)coc_, counters_);
    FIESTA_HIP_CHECK(hipGetLastError());
  }
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
}

// Raw dump (write) / load of the whole map state: one routine for both directions (checkpoint.hpp).
void DenseMap::checkpoint(const char *path, bool write) {
  nn_clean_ = false;  // (the counters come back as they were saved)
  if (!write) nn_valid_ = false, mask_obs_count_ = -1, ++field_epoch_;  // (a loaded field: the host-side brick cache of the scalar queries is stale)
  use_device();
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
  DevFile f(path, write, stream_);
  checkpoint_header(f, FIESTA_HIP_MODE_ARRAY, g_);
  if (write) {
    FIESTA_HIP_CHECK(hipMemcpyAsync(h_counters_, counters_, C_COUNT * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream_));
    FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
  }
  unsigned long long c[C_COUNT];
  memcpy(c, h_counters_, sizeof(c));
  f.host(c, sizeof(c));
  ProbParams pp = pp_;
  f.host(&pp, sizeof(pp));
  Geom g = g_;
  f.host(&g, sizeof(g));  // (only the update ranges are taken from the file; the rest must equal this map's)
  uint32_t flags[4] = {stale_inf_ ? 1u : 0u, gocc_ ? 1u : 0u, win_dirty_ ? 1u : 0u, 0u};
  f.host(flags, sizeof(flags));
  if ((flags[1] != 0) != (gocc_ != nullptr)) throw Error(FIESTA_HIP_ERR_INVALID, "checkpoint: sharded / unsharded mismatch");
  const size_t nt = c[C_TOUCHED], ni = c[C_INSERT], nd = c[C_DELETE];
  if (!write) {
    // Everything the file claims is checked BEFORE any device state is replaced: the layout half of its Geom against
    // this map's, the queue lengths against the grid, and the file's size against the sum of its sections -- a
    // truncated or foreign file is refused with the map untouched.  (An I/O error after this point leaves the map
    // undefined: the caller must discard it.)
    Geom a = g, b = g_;
    a.wx0 = a.wy0 = a.wz0 = a.wx1 = a.wy1 = a.wz1 = a.px0 = a.py0 = a.pz0 = a.px1 = a.py1 = a.pz1 = 0;
    b.wx0 = b.wy0 = b.wz0 = b.wx1 = b.wy1 = b.wz1 = b.px0 = b.py0 = b.pz0 = b.px1 = b.py1 = b.pz1 = 0;
    if (memcmp(&a, &b, sizeof(Geom)) != 0) throw Error(FIESTA_HIP_ERR_INVALID, "checkpoint: geometry of the file does not match this map");
    const int wlo[3] = {g.wx0, g.wy0, g.wz0}, whi[3] = {g.wx1, g.wy1, g.wz1}, plo[3] = {g.px0, g.py0, g.pz0}, phi[3] = {g.px1, g.py1, g.pz1};
    for (int k = 0; k < 3; ++k)  // (windows are voxel coordinates of Pos2Vox of clamped positions: a few voxels around the array at most)
      if (wlo[k] < -4 || whi[k] > 4096 + 4 || plo[k] < -4 || phi[k] > 4096 + 4) throw Error(FIESTA_HIP_ERR_INVALID, "checkpoint: update range of the file is out of bounds");
    if (nt > (size_t)g_.n || ni > (size_t)g_.n || nd > (size_t)g_.n) throw Error(FIESTA_HIP_ERR_INVALID, "checkpoint: queue lengths of the file exceed the grid");
    const unsigned long long sections = 8 + (gocc_ ? 1 : 0);
    const unsigned long long expect = f.position() + sections * sizeof(unsigned long long) + (unsigned long long)g_.n * (sizeof(vox_t) + sizeof(double) + sizeof(unsigned long long)) +
                                      2ull * (unsigned long long)nbitwords_ * sizeof(uint32_t) + (gocc_ ? (unsigned long long)ngoccwords_ * sizeof(uint32_t) : 0ull) +
                                      (unsigned long long)(nt + ni + nd) * sizeof(uint32_t);
    if (f.file_size() != expect) throw Error(FIESTA_HIP_ERR_INVALID, "checkpoint: file size does not match its header (truncated or corrupt)");
    touched_.ensure(nt, stream_);
    ins_.ensure(ni, stream_);
    del_.ensure(nd, stream_);
  }
  f.device(coc_, (size_t)g_.n * sizeof(vox_t));
  f.device(logodds_, (size_t)g_.n * sizeof(double));
  f.device(cnt_, (size_t)g_.n * sizeof(unsigned long long));
  f.device(occbits_, (size_t)nbitwords_ * sizeof(uint32_t));
  f.device(latebits_, (size_t)nbitwords_ * sizeof(uint32_t));
  if (gocc_) f.device(gocc_, (size_t)ngoccwords_ * sizeof(uint32_t));
  f.device(touched_.p, nt * sizeof(uint32_t));
  f.device(ins_.p, ni * sizeof(uint32_t));
  f.device(del_.p, nd * sizeof(uint32_t));
  f.finish();
  if (write) return;
  c[C_LIST0] = c[C_LIST1] = c[C_LIST2] = 0;
  memcpy(h_counters_, c, sizeof(c));
  FIESTA_HIP_CHECK(hipMemcpyAsync(counters_, h_counters_, sizeof(c), hipMemcpyHostToDevice, stream_));
  touched_upper_ = (int64_t)nt;
  pp_ = pp;
  g_.wx0 = g.wx0, g_.wy0 = g.wy0, g_.wz0 = g.wz0, g_.wx1 = g.wx1, g_.wy1 = g.wy1, g_.wz1 = g.wz1;  // (the ranges only)
  g_.px0 = g.px0, g_.py0 = g.py0, g_.pz0 = g.pz0, g_.px1 = g.px1, g_.py1 = g.py1, g_.pz1 = g.pz1;
  stale_inf_ = flags[0] != 0;
  win_dirty_ = flags[2] != 0;
  host_counts_valid_ = false;
  // (the observed bitmap is the field's own: a word other than "never observed")
  hipLaunchKernelGGL(k_obs_rebuild, dim3(grid_for(nbitwords_, 256, 8192)), dim3(256), 0, stream_, g_, (const vox_t *)coc_, obsbits_, nbitwords_);
  FIESTA_HIP_CHECK(hipGetLastError());
  if (track_) {
    zero_counter(C_MAXD2);
    hipLaunchKernelGGL(k_maxd2_scan, dim3(grid_for(g_.n, 256, 4096)), dim3(256), 0, stream_, g_, (const vox_t *)coc_, counters_);
    FIESTA_HIP_CHECK(hipGetLastError());
  }
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
}

int64_t DenseMap::snapshot_count_updated(int slot) {
  use_device();
  if (slot < 0 || slot >= 4 || !snaps_[slot].valid) throw Error(FIESTA_HIP_ERR_STATE, "no such snapshot");
  zero_counter(C_SCRATCH);
  hipLaunchKernelGGL(k_count_updated, dim3(grid_for(g_.n, 256, 8192)), dim3(256), 0, stream_, g_,
                     (const vox_t *)snaps_[slot].coc.p, (const vox_t *)coc_, (const uint32_t *)occbits_,
                     (const uint32_t *)gocc_, &counters_[C_SCRATCH]);
  FIESTA_HIP_CHECK(hipGetLastError());
  return (int64_t)read_counter(C_SCRATCH);
}

// ---- multi-GPU: ghost-layer exchange and the replicated occupancy bitmap ----
// Boxes are in LOCAL array coordinates, inclusive. pack copies the words of a box into a dense buffer
// (x-major, z fastest); apply compares a received buffer with the local words of a box of the same shape
// (ghost cells): a word that differs is replaced, tagged as a frontier source if it carries an obstacle, and
// its tile is activated (list 0).
__global__ void k_halo_pack(Geom g, int x0, int y0, int z0, int ex, int ey, int ez, const vox_t *coc, uint32_t *out) {
  const int64_t n = (int64_t)ex * ey * ez;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int z = z0 + (int)(i % ez), y = y0 + (int)((i / ez) % ey), x = x0 + (int)(i / ((int64_t)ez * ey));
    out[i] = coc[g.idx(x, y, z)];
  }
}
__device__ inline vox_t strip_tag(vox_t w) { return w == kUnobserved ? w : (w & ~kAct); }
__global__ void k_halo_apply(Geom g, TileGrid tg, int x0, int y0, int z0, int ex, int ey, int ez, vox_t *coc,
                             const uint32_t *in, uint32_t *flag, uint32_t *list, unsigned long long *count,
                             unsigned long long *changed) {
  const int64_t n = (int64_t)ex * ey * ez;
  unsigned long long local = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int z = z0 + (int)(i % ez), y = y0 + (int)((i / ez) % ey), x = x0 + (int)(i / ((int64_t)ez * ey));
    const int64_t idx = g.idx(x, y, z);
    const vox_t mine = strip_tag(coc[idx]), theirs = strip_tag(in[i]);
    if (mine == theirs) continue;
    ++local;
    if (theirs & kNoCoc) {
      coc[idx] = theirs;  // unobserved / no obstacle: not a source
    } else {
      coc[idx] = theirs | kAct;
      const uint32_t t = tg.tile_of(x, y, z);
      if (flag[t] == 0u) activate_tile(t, flag, list, count);
    }
  }
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off);
  if ((threadIdx.x & 63) == 0 && local) atomicAdd(changed, local);
}
// Occupancy transitions of this shard since the queues were last drained: TWO words per entry, x | y << 16 and
// z | "occupied now" << 31 (global coordinates, up to 65535 / 2^31 per axis) -- idempotent, order-free updates for the
// other shards' replicas of the global bitmap.
__global__ void k_export_transitions(Geom g, const uint32_t *ins, int64_t ni, const uint32_t *del, int64_t nd,
                                     const uint32_t *occbits, uint32_t *out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= ni + nd) return;
  const uint32_t idx = i < ni ? ins[i] : del[i - ni];
  const int z = idx % g.nz, y = (idx / g.nz) % g.ny, x = idx / (g.nz * g.ny);
  out[2 * i] = (uint32_t)(x + g.gx0) | ((uint32_t)(y + g.gy0) << 16);
  out[2 * i + 1] = (uint32_t)(z + g.gz0) | (occ_test(occbits, g, x, y, z) ? 0x80000000u : 0u);
}
__global__ void k_apply_transitions(Geom g, const uint32_t *ent, int64_t n, uint32_t *gocc,
                                    unsigned long long *remote_del) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t e0 = ent[2 * i], e = ent[2 * i + 1];
  const int x = (int)(e0 & 0xFFFFu), y = (int)(e0 >> 16), z = (int)(e & 0x7FFFFFFFu);
  if (x >= g.GX || y >= g.GY || z >= g.GZ) return;
  if (e & 0x80000000u)
    atomicOr(&gocc[g.gbitword(x, y, z)], 1u << (z & 31));
  else {
    atomicAnd(&gocc[g.gbitword(x, y, z)], ~(1u << (z & 31)));
    *remote_del = 1ull;  // voxels of THIS shard may point at it: the next UpdateESDF must run the invalidation scan
  }
}

// ---- sparse ghost exchange (shard_group.hip) --------------------------------------------------------------------------
// diff: the owned cells of the inclusive LOCAL box that a neighbour shard keeps as ghost cells; a cell whose word (tag
// stripped) differs from what was last sent (`shadow`, same shape as the box) becomes one entry
// {linear index of the ghost cell in the RECEIVER's array, word} and the shadow is updated.  apply: the receiver
// replaces each named ghost cell that differs, tags it as a frontier source if it carries an obstacle and wakes its tile.
__global__ void k_halo_diff(Geom g, int x0, int y0, int z0, int ex, int ey, int ez, const vox_t *coc, uint32_t *shadow,
                            int rx0, int ry0, int rz0, int rny, int rnz, uint32_t *entries, unsigned long long *count) {
  const int64_t n = (int64_t)ex * ey * ez;
  for (int64_t i0 = blockIdx.x * (int64_t)blockDim.x; i0 < n; i0 += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = i0 + threadIdx.x;
    bool send = false;
    vox_t w = 0;
    int dx = 0, dy = 0, dz = 0;
    if (i < n) {
      dz = (int)(i % ez), dy = (int)((i / ez) % ey), dx = (int)(i / ((int64_t)ez * ey));
      w = strip_tag(coc[g.idx(x0 + dx, y0 + dy, z0 + dz)]);
      send = w != shadow[i];
      if (send) shadow[i] = w;
    }
    const unsigned long long m = __ballot(send);
    if (!m) continue;
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(count, (unsigned long long)__popcll(m));
    base = __shfl(base, leader);
    if (send) {
      const unsigned long long k = base + __popcll(m & ((1ull << lane) - 1ull));
      entries[2 * k] = (uint32_t)(((int64_t)(rx0 + dx) * rny + (ry0 + dy)) * rnz + (rz0 + dz));
      entries[2 * k + 1] = w;
    }
  }
}
__global__ void k_halo_apply_sparse(Geom g, TileGrid tg, const uint32_t *entries, int64_t n, vox_t *coc, uint32_t *flag,
                                    uint32_t *list, unsigned long long *count, unsigned long long *changed) {
  unsigned long long local = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t idx = entries[2 * i];
    const vox_t theirs = entries[2 * i + 1], mine = strip_tag(coc[idx]);
    if (mine == theirs) continue;
    ++local;
    if (theirs & kNoCoc) {
      coc[idx] = theirs;
    } else {
      coc[idx] = theirs | kAct;
      const int z = idx % g.nz, y = (idx / g.nz) % g.ny, x = idx / (g.nz * g.ny);
      const uint32_t t = tg.tile_of(x, y, z);
      if (flag[t] == 0u) activate_tile(t, flag, list, count);
    }
  }
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off);
  if ((threadIdx.x & 63) == 0 && local) atomicAdd(changed, local);
}

static void check_box(const Geom &g, const int32_t *lo, const int32_t *hi) {
  if (lo[0] < 0 || lo[1] < 0 || lo[2] < 0 || hi[0] >= g.nx || hi[1] >= g.ny || hi[2] >= g.nz || hi[0] < lo[0] ||
      hi[1] < lo[1] || hi[2] < lo[2])
    throw Error(FIESTA_HIP_ERR_INVALID, "halo box outside the local array");
}

void DenseMap::halo_pack(const int32_t *lo, const int32_t *hi, uint32_t *out_dev) {
  use_device();
  check_box(g_, lo, hi);
  const int ex = hi[0] - lo[0] + 1, ey = hi[1] - lo[1] + 1, ez = hi[2] - lo[2] + 1;
  hipLaunchKernelGGL(k_halo_pack, dim3(grid_for((int64_t)ex * ey * ez, 256, 8192)), dim3(256), 0, stream_, g_, lo[0],
                     lo[1], lo[2], ex, ey, ez, (const vox_t *)coc_, out_dev);
  FIESTA_HIP_CHECK(hipGetLastError());
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));  // the buffer is handed to another stream / RCCL next
}

int64_t DenseMap::halo_apply(const int32_t *lo, const int32_t *hi, const uint32_t *in_dev) {
  ++field_epoch_;  // (the host-side brick cache of the scalar queries is stale from here on)
  use_device();
  check_box(g_, lo, hi);
  const int ex = hi[0] - lo[0] + 1, ey = hi[1] - lo[1] + 1, ez = hi[2] - lo[2] + 1;
  TileGrid tg{tx_, ty_, ntx_, nty_, ntz_};
  zero_counter(C_SCRATCH);
  hipLaunchKernelGGL(k_halo_apply, dim3(grid_for((int64_t)ex * ey * ez, 256, 8192)), dim3(256), 0, stream_, g_, tg, lo[0],
                     lo[1], lo[2], ex, ey, ez, coc_, in_dev, tile_flag_[0], tile_list_[0], &counters_[C_LIST0],
                     &counters_[C_SCRATCH]);
  FIESTA_HIP_CHECK(hipGetLastError());
  return (int64_t)read_counter(C_SCRATCH);
}

int64_t DenseMap::export_transitions(uint32_t *out_dev, int64_t cap) {
  use_device();
  FIESTA_HIP_CHECK(hipMemcpyAsync(&h_counters_[C_INSERT], &counters_[C_INSERT], 2 * sizeof(unsigned long long),
                                  hipMemcpyDeviceToHost, stream_));
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
  const int64_t ni = (int64_t)h_counters_[C_INSERT], nd = (int64_t)h_counters_[C_DELETE];
  if (out_dev == nullptr) return ni + nd;
  if (ni + nd > cap) throw Error(FIESTA_HIP_ERR_INVALID, "transition buffer too small");
  if (ni + nd) {
    hipLaunchKernelGGL(k_export_transitions, dim3(grid_for(ni + nd)), dim3(256), 0, stream_, g_, (const uint32_t *)ins_.p,
                       ni, (const uint32_t *)del_.p, nd, (const uint32_t *)occbits_, out_dev);
    FIESTA_HIP_CHECK(hipGetLastError());
    FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
  }
  return ni + nd;
}

void DenseMap::apply_transitions(const uint32_t *ent_dev, int64_t n) {
  ++field_epoch_;  // (the host-side brick cache of the scalar queries is stale from here on)
  use_device();
  if (!g_.sharded) throw Error(FIESTA_HIP_ERR_STATE, "apply_transitions: not a sharded map");
  if (n <= 0) return;
  hipLaunchKernelGGL(k_apply_transitions, dim3(grid_for(n)), dim3(256), 0, stream_, g_, ent_dev, n, gocc_,
                     &counters_[C_REMOTE_DEL]);
  FIESTA_HIP_CHECK(hipGetLastError());
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
}

void DenseMap::halo_diff(const int32_t *lo, const int32_t *hi, uint32_t *shadow_dev, const int32_t *recv_lo,
                         const int32_t *recv_dims, uint32_t *entries_dev, unsigned long long *count_dev) {
  use_device();
  check_box(g_, lo, hi);
  const int ex = hi[0] - lo[0] + 1, ey = hi[1] - lo[1] + 1, ez = hi[2] - lo[2] + 1;
  hipLaunchKernelGGL(k_halo_diff, dim3(grid_for((int64_t)ex * ey * ez, 256, 4096)), dim3(256), 0, stream_, g_, lo[0], lo[1],
                     lo[2], ex, ey, ez, (const vox_t *)coc_, shadow_dev, recv_lo[0], recv_lo[1], recv_lo[2], recv_dims[1],
                     recv_dims[2], entries_dev, count_dev);
  FIESTA_HIP_CHECK(hipGetLastError());
}

void DenseMap::halo_apply_sparse(const uint32_t *entries_dev, int64_t n, unsigned long long *changed_dev) {
  ++field_epoch_;  // (the host-side brick cache of the scalar queries is stale from here on)
  use_device();
  if (n <= 0) return;
  TileGrid tg{tx_, ty_, ntx_, nty_, ntz_};
  hipLaunchKernelGGL(k_halo_apply_sparse, dim3(grid_for(n, 256, 4096)), dim3(256), 0, stream_, g_, tg, entries_dev, n, coc_,
                     tile_flag_[0], tile_list_[0], &counters_[C_LIST0], changed_dev);
  FIESTA_HIP_CHECK(hipGetLastError());
}

// The scratch buffers of a sharded bulk transform with this margin, allocated now (group creation) instead of inside the
// first update (measured: 250 ms of hipMalloc on the critical path of the first UpdateESDF at 1024^3).
void DenseMap::bulk_reserve(int margin) {
  use_device();
  const Geom &g = g_;
  if (!g.sharded) return;
  const int l0[3] = {g.gx0, g.gy0, g.gz0}, ln[3] = {g.nx, g.ny, g.nz}, G[3] = {g.GX, g.GY, g.GZ};
  int64_t ext[3];
  bool open_side = false;
  for (int k = 0; k < 3; ++k) {
    const int lo = std::max(0, l0[k] - margin), hi = std::min(G[k] - 1, l0[k] + ln[k] - 1 + margin + (k == 2 ? 31 : 0));
    ext[k] = hi - lo + 1;
    open_side = open_side || lo > 0 || hi < G[k] - 1;
    if (ext[k] > 2048) return;
  }
  const int64_t nzc = (ext[2] + 63) / 64;
  ft_inter_.ensure((size_t)(ext[0] * ext[1] * ext[2]), stream_);
  ft_rowlist_.ensure((size_t)(ext[0] * ext[1]), stream_);
  ft_rowcnt_.ensure((size_t)ext[0] + 64, stream_);
  ft_spill_.ensure_exact((size_t)std::min<int64_t>((std::max(ext[0], ext[1]) * nzc + 3) / 4, kFtBlocks) * 4 * ((size_t)std::max(ext[0], ext[1]) + 2) * (512u / sizeof(unsigned long long)), stream_);
  if (open_side) ft_out_.ensure_exact((size_t)g.n, stream_);
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
}

// device address of a counter (the shard group assembles its per-sweep row on the device)
const unsigned long long *DenseMap::counter_dev(int which) const { return &counters_[which]; }

int64_t DenseMap::pending_tiles() {
  use_device();
  return (int64_t)read_counter(C_LIST0);
}

void DenseMap::synchronize() {
  use_device();
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
}

}  // namespace fiesta
