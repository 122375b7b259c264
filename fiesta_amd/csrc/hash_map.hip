// fiesta_amd/csrc/hash_map.hip -- gfx950 kernels + host driver of the sparse ("hash-block") ESDF map.
// Design: hash_map.hpp. What replaces what (reference = HKUST-Aerial-Robotics/FIESTA, -DHASH_TABLE build):
//   hash_table_ / FindAndInsert / IncreaseCapacity   src/ESDFMap.cpp:705-765 -> page directory + page pool
//   SetOccupancy x2 (PosInMap == true, :46-48)       :401-437                 -> k_h_mark + k_h_observe_*
//   UpdateOccupancy                                  :235-271                 -> k_h_fuse
//   UpdateESDF                                       :273-398                 -> k_h_seed_insert, k_h_invalidate,
//                                                                                k_relax_q<16,16,1024,PAGED>
//   GetDistance / GetDistWithGradTrilinear / GetOccupancy :452-540           -> k_h_query_*
#include "hash_map.hpp"
#include "checkpoint.hpp"

#include <vector>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "level_kernels.hpp"
#include "relax_kernels.hpp"

namespace fiesta {

namespace {
constexpr int kWin = HashMap::kWin, kHalf = HashMap::kHalf;
constexpr int kNTY = HashMap::kNTY, kNTZ = HashMap::kNTZ, kNTiles = HashMap::kNTiles;
constexpr int kPageVox = HashMap::kPageVox;
constexpr int C_OUTSIDE = C_REMOTE_DEL;  // (dense shards only use that slot) a batch held a voxel outside the window

__device__ inline int tile_id(int x, int y, int z) { return ((x >> 4) * kNTY + (y >> 4)) * kNTZ + (z >> 5); }
__device__ inline bool in_win(int x, int y, int z) {
  return (unsigned)x < (unsigned)kWin && (unsigned)y < (unsigned)kWin && (unsigned)z < (unsigned)kWin;
}
// pool address of window voxel (x,y,z); -1 if its page is not allocated
__device__ inline int64_t vaddr(const int32_t *dir, int x, int y, int z) {
  const int32_t p = dir[tile_id(x, y, z)];
  return p < 0 ? -1 : (int64_t)p * kPageVox + (((x & 15) * 16 + (y & 15)) * 32 + (z & 31));
}
__device__ inline void vcoords(const int32_t *page_tile, uint32_t addr, int &x, int &y, int &z) {
  const int t = page_tile[addr / kPageVox], off = addr % kPageVox;
  x = (t / (kNTY * kNTZ)) * 16 + (off >> 9);
  y = ((t / kNTZ) % kNTY) * 16 + ((off >> 5) & 15);
  z = (t % kNTZ) * 32 + (off & 31);
}
__device__ inline bool hbit(const uint32_t *bits, int64_t addr) { return (bits[addr >> 5] >> (addr & 31)) & 1u; }
// The window MOVES (HashMap::ensure_window): window voxel (0,0,0) is map voxel (g.gx0, g.gy0, g.gz0), a multiple of the
// tile size.  Closest-obstacle ids are therefore stored as MAP coordinates modulo 1024 and decoded relative to the voxel
// that holds them (common.hpp: coc_offset with wrap; reach 512 voxels) -- they stay valid when the window moves.
__device__ inline vox_t h_pack(const Geom &g, int x, int y, int z) { return pack_coc(x + g.gx0, y + g.gy0, z + g.gz0); }
__device__ inline int32_t h_dist2(const Geom &g, int x, int y, int z, vox_t w) {
  return dist2(1, x + g.gx0, y + g.gy0, z + g.gz0, w);
}
// window coordinates of the obstacle named by the id held at window voxel (x, y, z) (may lie outside the window)
__device__ inline void h_obstacle(const Geom &g, int x, int y, int z, vox_t w, int &cx, int &cy, int &cz) {
  int dx, dy, dz;
  coc_offset(1, x + g.gx0, y + g.gy0, z + g.gz0, w, dx, dy, dz);
  cx = x - dx, cy = y - dy, cz = z - dz;
}
__device__ inline bool h_alive(const Geom &g, const int32_t *dir, const uint32_t *occbits, int x, int y, int z, vox_t w) {
  int cx, cy, cz;
  h_obstacle(g, x, y, z, w, cx, cy, cz);
  if (!in_win(cx, cy, cz)) return false;  // (its page left the window with it)
  const int64_t ca = vaddr(dir, cx, cy, cz);
  return ca >= 0 && hbit(occbits, ca);
}

// one thread per element unless the caller names a cap (= the kernel strides over the grid): the default must cover the
// largest arrays (a 1024^3 shard touches 10^9 voxels at once -- a cap of 2^20 blocks silently dropped three quarters
// of them, found by tools/c5_smoke.py)
static inline int grid_for(int64_t n, int block = 256, int cap = 0x7FFFFFFF) {
  int64_t b = (n + block - 1) / block;
  return (int)std::min<int64_t>(std::max<int64_t>(b, 1), cap);
}

template <typename T>
__global__ void k_h_fill(T *p, T v, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

// ---- page allocation: mark the tiles a batch touches, collect the unallocated ones, assign fresh pages ----
__device__ inline bool obs_vox(const Geom &g, const int32_t *vox, int64_t i, int &x, int &y, int &z) {
  x = vox[3 * i] - g.gx0, y = vox[3 * i + 1] - g.gy0, z = vox[3 * i + 2] - g.gz0;
  return in_win(x, y, z) && g.in_window(x, y, z);  // VoxInRange (src/ESDFMap.cpp:420)
}
__device__ inline bool obs_pos(const Geom &g, const double *pos, const int32_t *occ, int64_t i, int &x, int &y, int &z) {
  const int o = occ[i];
  if (o != 0 && o != 1) return false;  // "occ value error!" (:402-405); PosInMap is always true here (:46-48)
  x = (int)floor((pos[3 * i] - g.org[0]) / g.res) - g.gx0;  // Pos2Vox (:74-77)
  y = (int)floor((pos[3 * i + 1] - g.org[1]) / g.res) - g.gy0;
  z = (int)floor((pos[3 * i + 2] - g.org[2]) / g.res) - g.gz0;
  return in_win(x, y, z) && g.in_window(x, y, z);
}
// (outside: set when a voxel of the batch lies outside the window -- the host then moves the window and marks again)
__global__ void k_h_mark_vox(Geom g, const int32_t *vox, int64_t n, uint32_t *need, unsigned long long *outside) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  int x, y, z;
  if (obs_vox(g, vox, i, x, y, z))
    need[tile_id(x, y, z)] = 1u;
  else if (outside && !in_win(x, y, z))
    *outside = 1ull;
}
__global__ void k_h_mark_pos(Geom g, const double *pos, const int32_t *occ, int64_t n, uint32_t *need) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int x, y, z;
  if (i < n && obs_pos(g, pos, occ, i, x, y, z)) need[tile_id(x, y, z)] = 1u;
}
__global__ void k_h_collect(uint32_t *need, const int32_t *dir, uint32_t *fresh, unsigned long long *count) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= kNTiles || !need[t]) return;
  need[t] = 0u;
  if (dir[t] < 0) fresh[atomicAdd(count, 1ull)] = (uint32_t)t;
}
__global__ void k_h_assign(Geom g, const uint32_t *fresh, int64_t k, int32_t first_page, int32_t *dir, int32_t *page_tile,
                           int32_t *page_gtile) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= k) return;
  const int t = (int)fresh[i], page = first_page + (int32_t)i;
  dir[t] = page;
  page_tile[page] = t;
  page_gtile[3 * page] = (g.gx0 >> 4) + t / (kNTY * kNTZ);  // (the window origin is a whole number of tiles)
  page_gtile[3 * page + 1] = (g.gy0 >> 4) + (t / kNTZ) % kNTY;
  page_gtile[3 * page + 2] = (g.gz0 >> 5) + t % kNTZ;
}
// The window moved to g's origin: rebuild the directory (cleared by the caller) from the pages' map tiles.  A page
// outside the new window is parked (page_tile -1); one that comes back is marked fresh for k_h_invalidate.
__global__ void k_h_rebuild(Geom g, int64_t npages, const int32_t *page_gtile, int32_t *dir, int32_t *page_tile,
                            uint32_t *page_fresh) {
  const int64_t page = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (page >= npages) return;
  const int tx = page_gtile[3 * page] - (g.gx0 >> 4), ty = page_gtile[3 * page + 1] - (g.gy0 >> 4),
            tz = page_gtile[3 * page + 2] - (g.gz0 >> 5);
  const bool in = (unsigned)tx < (unsigned)(kWin / 16) && (unsigned)ty < (unsigned)kNTY && (unsigned)tz < (unsigned)kNTZ;
  if (in) {
    const int t = (tx * kNTY + ty) * kNTZ + tz;
    dir[t] = (int32_t)page;
    if (page_tile[page] < 0) page_fresh[page] = 1u;
    page_tile[page] = t;
  } else {
    page_tile[page] = -1;
  }
}
// bounding box of a device-resident batch of voxels (min xyz, max xyz), for ensure_window
__global__ void k_h_bbox(const int32_t *vox, int64_t n, int32_t *box) {
  int lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    for (int k = 0; k < 3; ++k) lo[k] = min(lo[k], vox[3 * i + k]), hi[k] = max(hi[k], vox[3 * i + k]);
  __shared__ int blo[3], bhi[3];
  if (threadIdx.x < 3) blo[threadIdx.x] = INT32_MAX, bhi[threadIdx.x] = INT32_MIN;
  __syncthreads();
  for (int k = 0; k < 3; ++k) {
    for (int off = 32; off; off >>= 1) lo[k] = min(lo[k], __shfl_xor(lo[k], off)), hi[k] = max(hi[k], __shfl_xor(hi[k], off));
    if ((threadIdx.x & 63) == 0 && lo[k] <= hi[k]) atomicMin(&blo[k], lo[k]), atomicMax(&bhi[k], hi[k]);
  }
  __syncthreads();
  if (threadIdx.x < 3 && blo[threadIdx.x] <= bhi[threadIdx.x])
    atomicMin(&box[threadIdx.x], blo[threadIdx.x]), atomicMax(&box[3 + threadIdx.x], bhi[threadIdx.x]);
}

// ---- SetOccupancy (src/ESDFMap.cpp:401-437) ----
__device__ inline void h_count(int64_t addr, int occ, unsigned long long *cnt, uint32_t *touched, unsigned long long *counters) {
  const unsigned long long old = atomicAdd(&cnt[addr], ((unsigned long long)(uint32_t)occ << 32) | 1ull);
  wave_append((uint32_t)old == 0, (uint32_t)addr, touched, &counters[C_TOUCHED]);
}
__global__ void k_h_observe_vox(Geom g, const int32_t *dir, const int32_t *vox, const int32_t *occ, int64_t n,
                                unsigned long long *cnt, uint32_t *touched, unsigned long long *counters) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int x = 0, y = 0, z = 0;
  const bool ok = i < n && obs_vox(g, vox, i, x, y, z);
  if (ok) h_count(vaddr(dir, x, y, z), occ[i], cnt, touched, counters);
  // the reference's hash build accepts any voxel; an observation outside the addressable window is lost here: count it
  const unsigned long long lost = __ballot(i < n && !in_win(x, y, z));
  if (lost && (threadIdx.x & 63) == 0) atomicAdd(&counters[C_DROPPED], (unsigned long long)__popcll(lost));
}
__global__ void k_h_observe_pos(Geom g, const int32_t *dir, const double *pos, const int32_t *occ, int64_t n,
                                unsigned long long *cnt, uint32_t *touched, unsigned long long *counters) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int x = 0, y = 0, z = 0;
  const bool valid = i < n && (occ[i] == 0 || occ[i] == 1);
  if (valid && obs_pos(g, pos, occ, i, x, y, z)) h_count(vaddr(dir, x, y, z), occ[i], cnt, touched, counters);
  const unsigned long long lost = __ballot(valid && !in_win(x, y, z));
  if (lost && (threadIdx.x & 63) == 0) atomicAdd(&counters[C_DROPPED], (unsigned long long)__popcll(lost));
}

// SetOccupancy(Vector3i, occ) for every voxel of a box given in WINDOW coordinates (inclusive): tiles first ...
__global__ void k_h_mark_box(Geom g, int x0, int y0, int z0, int x1, int y1, int z1, uint32_t *need) {
  const int tx0 = x0 >> 4, ty0 = y0 >> 4, tz0 = z0 >> 5;
  const int ntx = (x1 >> 4) - tx0 + 1, nty = (y1 >> 4) - ty0 + 1, ntz = (z1 >> 5) - tz0 + 1;
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= (int64_t)ntx * nty * ntz) return;
  const int tz = (int)(i % ntz), ty = (int)((i / ntz) % nty), tx = (int)(i / ((int64_t)ntz * nty));
  need[((tx0 + tx) * kNTY + (ty0 + ty)) * kNTZ + (tz0 + tz)] = 1u;
}
// ... then the voxels: 16 waves per work-group, one z-run of 64 voxels of a box row per wave and step; every voxel is
// visited once (no atomic on the counter word), first-touch appends aggregated per work-group (dense_map.hip,
// k_observe_box).
__global__ __launch_bounds__(1024) void k_h_observe_box(Geom g, const int32_t *dir, int x0, int y0, int z0, int ex, int ey,
                                                        int ez, int occ, unsigned long long *cnt, uint32_t *touched,
                                                        unsigned long long *counters) {
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
    int64_t addr = 0;
    if (item < nitems) {
      const int zc = (int)(item % zchunks);
      const int64_t row = item / zchunks;
      const int y = y0 + (int)(row % ey), x = x0 + (int)(row / ey);
      const int zi = zc * 64 + lane, z = z0 + zi;
      if (zi < ez && in_win(x, y, z) && g.in_window(x, y, z)) {
        addr = vaddr(dir, x, y, z);
        const unsigned long long old = cnt[addr];
        cnt[addr] = old + (((unsigned long long)(uint32_t)occ << 32) | 1ull);
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
    if (first) touched[blk_base + woff + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)addr;
  }
}

// "updated voxel" as SURVEY.md 8d defines it (d^2 differs, or the old closest obstacle vanished), between a saved copy
// of the pool's state words and now; pages allocated after the copy read as unobserved before.
__global__ void k_h_count_updated(Geom g, const int32_t *dir, const int32_t *page_tile, const vox_t *before, int64_t n_before,
                                  const vox_t *now, int64_t n_now, const uint32_t *occbits, unsigned long long *out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long local = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_now; i += stride) {
    vox_t a = i < n_before ? before[i] : kUnobserved, b = now[i];
    if (a != kUnobserved) a &= ~kAct;
    if (b != kUnobserved) b &= ~kAct;
    if (a == b || page_tile[i / kPageVox] < 0) continue;  // (a page that left the window is no longer part of the map)
    int x, y, z;
    vcoords(page_tile, (uint32_t)i, x, y, z);
    const int32_t da = (a == kUnobserved) ? -1 : ((a & kNoCoc) ? kD2Inf : h_dist2(g, x, y, z, a));
    const int32_t db = (b == kUnobserved) ? -1 : ((b & kNoCoc) ? kD2Inf : h_dist2(g, x, y, z, b));
    bool upd = da != db;
    if (!upd && !(a & kNoCoc)) upd = !h_alive(g, dir, occbits, x, y, z, a);
    local += upd;
  }
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off);
  if ((threadIdx.x & 63) == 0 && local) atomicAdd(out, local);
}

// ---- UpdateOccupancy (src/ESDFMap.cpp:235-271) ----
__global__ void k_h_fuse(Geom g, ProbParams pp, int global_map, const int32_t *page_tile, const uint32_t *touched,
                         int64_t n, unsigned long long *cnt, double *logodds, vox_t *coc, uint32_t *occbits, uint32_t *ins,
                         uint32_t *del, unsigned long long *counters) {
  if (n < 0) n = (int64_t)counters[C_TOUCHED];  // the host only knows an upper bound (it sized the grid with it)
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t a = touched[i];
    const unsigned long long c = cnt[a];
    cnt[a] = 0;
    const int64_t hits = (int64_t)(int32_t)(c >> 32), seen = (int64_t)(uint32_t)c;
    const double step = (hits >= seen - hits) ? pp.l_hit : pp.l_miss;
    double L = logodds[a];
    const bool was = L > pp.l_occ;
    if (coc[a] == kUnobserved) coc[a] = kInf;
    if ((step >= 0 && L >= pp.l_max) || (step <= 0 && L <= pp.l_min)) continue;
    if (!global_map && page_tile[a / kPageVox] >= 0) {  // (a page parked since the observation fuses like a global map's)
      int x, y, z;
      vcoords(page_tile, a, x, y, z);
      if (!g.in_prev_window(x, y, z)) {  // (distance = infinity, the link stays: common.hpp, stale_link)
        L = 0;
        coc[a] = stale_link(coc[a]);
      }
    }
    L = fmin(fmax(L + step, pp.l_min), pp.l_max);
    logodds[a] = L;
    const bool now = L > pp.l_occ;
    const uint32_t bit = 1u << (a & 31);
    if (now && !was) {
      atomicOr(&occbits[a >> 5], bit);
      wave_append(true, a, ins, &counters[C_INSERT]);  // one atomic per wave on the hot counter
    } else if (!now && was) {
      atomicAnd(&occbits[a >> 5], ~bit);
      wave_append(true, a, del, &counters[C_DELETE]);
    }
  }
}

// ---- UpdateESDF seeding (src/ESDFMap.cpp:278-337) ----
__global__ void k_h_seed_insert(Geom g, const int32_t *page_tile, const uint32_t *ins, int64_t n, vox_t *coc,
                                const uint32_t *occbits, uint32_t *flag, uint32_t *list, unsigned long long *count) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t a = ins[i];
  if (!hbit(occbits, a) || page_tile[a / kPageVox] < 0) return;
  int x, y, z;
  vcoords(page_tile, a, x, y, z);
  coc[a] = h_pack(g, x, y, z) | kAct;
  activate_tile((uint32_t)page_tile[a / kPageVox], flag, list, count);
}
// LEVELS: the level engine seeds itself from this scan (level_kernels.hpp; as dense_map.hip: k_invalidate<true>).
template <bool LEVELS>
__global__ __launch_bounds__(256) void k_h_invalidate(Geom g, const int32_t *dir, const int32_t *page_tile,
                                                      const uint32_t *page_fresh, int64_t nvox, vox_t *coc,
                                                      const uint32_t *occbits, uint32_t *flag, uint32_t *list,
                                                      unsigned long long *count, unsigned long long *counters, LevelArgs lv) {
  const int lane = threadIdx.x & 63;
  unsigned long long local = 0;
  const bool win_all = g.wx0 <= 0 && g.wy0 <= 0 && g.wz0 <= 0 && g.wx1 >= kWin - 1 && g.wy1 >= kWin - 1 && g.wz1 >= kWin - 1;
  for (int64_t base = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64; base < nvox; base += (int64_t)gridDim.x * 256) {
    const int64_t a = base + lane;  // 64 consecutive pool words = two z-rows of one page
    bool reset = false, outside = false;
    uint32_t entry = 0;  // (LEVELS: the level engine names a voxel by its packed window coordinates)
    const vox_t w = coc[a];
    const int32_t tile = page_tile[a / kPageVox];  // (wave-uniform; < 0: the page left the window)
    if (tile >= 0 && page_fresh[a / kPageVox]) {
      // a page that was parked missed every update meanwhile: rebuild it -- its obstacles are seeds again, every other
      // observed voxel asks its neighbours
      if (w != kUnobserved) {
        vox_t nw = kReset;
        if (LEVELS || hbit(occbits, a)) {
          int x, y, z;
          vcoords(page_tile, (uint32_t)a, x, y, z);
          if (hbit(occbits, a)) nw = h_pack(g, x, y, z) | kAct;
          entry = ((uint32_t)x << 20) | ((uint32_t)y << 10) | (uint32_t)z;
        }
        coc[a] = nw;
        reset = true;
      }
    } else if (tile >= 0 && has_link(w)) {
      int x, y, z;
      vcoords(page_tile, (uint32_t)a, x, y, z);
      entry = ((uint32_t)x << 20) | ((uint32_t)y << 10) | (uint32_t)z;
      if (!h_alive(g, dir, occbits, x, y, z, w & kIdMask)) {
        if (LEVELS && !win_all && !g.in_window(x, y, z)) {
          outside = true;  // (keeps its word until k_level_outside has judged it)
        } else {
          coc[a] = LEVELS ? (kReset | (w & kIdMask)) : kReset;
          reset = true;
        }
      }
    }
    if (LEVELS) {
      const unsigned long long mm = __ballot(reset || outside);
      if (mm) {
        bool fits = lv_append(reset, entry, lv.list[0], &lv.ctl->n[0], lv.cap);
        if (!win_all) fits &= lv_append(outside, entry, lv.outside, &lv.ctl->nout, lv.cap);
        if (!fits) lv.ctl->overflow = 1;
        if (lane == 0) local += __popcll(mm);
      }
      continue;
    }
    const unsigned long long m = __ballot(reset);
    if (m && lane == 0) {
      local += __popcll(m);
      const uint32_t t = (uint32_t)tile;
      if (flag[t] == 0u) activate_tile(t, flag, list, count);
    }
  }
  __shared__ unsigned long long blk_local;  // one atomic per work-group on the hot counter
  if (threadIdx.x == 0) blk_local = 0;
  __syncthreads();
  if (lane == 0 && local) atomicAdd(&blk_local, local);
  __syncthreads();
  if (threadIdx.x == 0 && blk_local) {
    if (LEVELS)
      atomicAdd(&lv.ctl->invalidated, (uint32_t)blk_local);
    else
      atomicAdd(&counters[C_INVALIDATED], blk_local);
  }
}

// ---- queries (src/ESDFMap.cpp:452-540); an unallocated voxel reads like a freshly allocated one ----
// The reference's hash map answers for ANY voxel it ever allocated (:732-765).  Inside the window a lookup is one load of
// the dense directory; outside it -- pages that are parked -- a query goes through the MAP-WIDE page table: every page's
// tile in map coordinates as a sorted 64-bit key (a few thousand entries: a binary search), so that a planner may ask for
// a goal far from the sensor.  A parked page answers with the field it held when the window left it.
__host__ __device__ inline unsigned long long tile_key(int tx, int ty, int tz) {
  return ((unsigned long long)(uint32_t)(tx + (1 << 20)) << 42) | ((unsigned long long)(uint32_t)(ty + (1 << 20)) << 21) |
         (unsigned long long)(uint32_t)(tz + (1 << 20));
}
// pool address of WINDOW voxel (x, y, z), which may lie outside the window; -1: no page anywhere holds it
__device__ inline int64_t h_lookup(const Geom &g, const int32_t *dir, const PageTable &tab, int x, int y, int z) {
  if (in_win(x, y, z)) return vaddr(dir, x, y, z);
  const int vx = x + g.gx0, vy = y + g.gy0, vz = z + g.gz0;  // map voxel; the window origin is a whole number of tiles
  const unsigned long long key = tile_key(vx >> 4, vy >> 4, vz >> 5);
  int lo = 0, hi = tab.n - 1;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    const unsigned long long k = tab.keys[mid];
    if (k == key) return (int64_t)tab.pages[mid] * kPageVox + (((vx & 15) * 16 + (vy & 15)) * 32 + (vz & 31));
    if (k < key)
      lo = mid + 1;
    else
      hi = mid - 1;
  }
  return -1;
}
__device__ inline double h_distance(const Geom &g, const int32_t *dir, const PageTable &tab, const vox_t *coc, int x, int y, int z) {
  const int64_t a = h_lookup(g, dir, tab, x, y, z);
  if (a < 0) return (double)FIESTA_HIP_INFINITY;
  const vox_t w = coc[a] & ~kAct;
  if (w & kNoCoc) return (double)FIESTA_HIP_INFINITY;
  return sqrt((double)h_dist2(g, x, y, z, w)) * g.res;
}
__global__ void k_h_query_dist(Geom g, const int32_t *dir, PageTable tab, const vox_t *coc, const int32_t *vox, const double *pos,
                               int64_t n, double *out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  int x, y, z;
  if (vox) {
    x = vox[3 * i] - g.gx0, y = vox[3 * i + 1] - g.gy0, z = vox[3 * i + 2] - g.gz0;
  } else {
    x = (int)floor((pos[3 * i] - g.org[0]) / g.res) - g.gx0;
    y = (int)floor((pos[3 * i + 1] - g.org[1]) / g.res) - g.gy0;
    z = (int)floor((pos[3 * i + 2] - g.org[2]) / g.res) - g.gz0;
  }
  out[i] = h_distance(g, dir, tab, coc, x, y, z);
}
__global__ void k_h_query_occ(Geom g, const int32_t *dir, PageTable tab, const uint32_t *occbits, const int32_t *vox, const double *pos,
                              int64_t n, int32_t *out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  int x, y, z;
  if (vox) {
    x = vox[3 * i] - g.gx0, y = vox[3 * i + 1] - g.gy0, z = vox[3 * i + 2] - g.gz0;
  } else {
    x = (int)floor((pos[3 * i] - g.org[0]) / g.res) - g.gx0;
    y = (int)floor((pos[3 * i + 1] - g.org[1]) / g.res) - g.gy0;
    z = (int)floor((pos[3 * i + 2] - g.org[2]) / g.res) - g.gz0;
  }
  const int64_t a = h_lookup(g, dir, tab, x, y, z);
  out[i] = a < 0 ? 0 : (int)hbit(occbits, a);
}
// GetDistWithGradTrilinear (src/ESDFMap.cpp:481-540), f64 in the reference's operation order -- written once over a source of
// corner distances D(map voxel x, y, z), compiled for the device (the batch kernel) and for the host (scalar calls from the brick
// cache): the two are bit-equal
template <class D>
__host__ __device__ inline double h_trilinear(const Geom &g, D &&corner, const double *p, double *grad) {
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
      for (int iz = 0; iz < 2; ++iz) v[ix][iy][iz] = corner(b[0] + ix, b[1] + iy, b[2] + iz);
  const double v00 = (1 - f[0]) * v[0][0][0] + f[0] * v[1][0][0];
  const double v01 = (1 - f[0]) * v[0][0][1] + f[0] * v[1][0][1];
  const double v10 = (1 - f[0]) * v[0][1][0] + f[0] * v[1][1][0];
  const double v11 = (1 - f[0]) * v[0][1][1] + f[0] * v[1][1][1];
  const double v0 = (1 - f[1]) * v00 + f[1] * v10;
  const double v1 = (1 - f[1]) * v01 + f[1] * v11;
  const double d = (1 - f[2]) * v0 + f[2] * v1;
  if (grad) {
    grad[2] = (v1 - v0) * g.res_inv;
    grad[1] = ((1 - f[2]) * (v10 - v00) + f[2] * (v11 - v01)) * g.res_inv;
    double gx = (1 - f[2]) * (1 - f[1]) * (v[1][0][0] - v[0][0][0]);
    gx += (1 - f[2]) * f[1] * (v[1][1][0] - v[0][1][0]);
    gx += f[2] * (1 - f[1]) * (v[1][0][1] - v[0][0][1]);
    gx += f[2] * f[1] * (v[1][1][1] - v[0][1][1]);
    grad[0] = gx * g.res_inv;
  }
  return d;
}
__global__ void k_h_query_trilinear(Geom g, const int32_t *dir, PageTable tab, const vox_t *coc, const double *pos, int64_t n, double *dist,
                                    double *grad) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double p[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
  auto corner = [&](int vx, int vy, int vz) { return h_distance(g, dir, tab, coc, vx - g.gx0, vy - g.gy0, vz - g.gz0); };
  dist[i] = h_trilinear(g, corner, p, grad ? grad + 3 * i : nullptr);
}
// a brick of the host-side cache: the 16^3 distances and occupancy bits of map brick (bx, by, bz), straight into pinned memory
__global__ __launch_bounds__(256) void k_h_fetch_brick(Geom g, const int32_t *dir, PageTable tab, const vox_t *coc, const uint32_t *occbits, int bx,
                                                       int by, int bz, double *dst) {
  const int t = threadIdx.x, x = 16 * bx + (t >> 4) - g.gx0, y = 16 * by + (t & 15) - g.gy0, z0 = 16 * bz - g.gz0;
  uint32_t bits = 0;
  for (int k = 0; k < 16; ++k) {
    dst[t * 16 + k] = h_distance(g, dir, tab, coc, x, y, z0 + k);
    const int64_t a = h_lookup(g, dir, tab, x, y, z0 + k);
    if (a >= 0 && hbit(occbits, a)) bits |= 1u << k;
  }
  reinterpret_cast<uint16_t *>(dst + 4096)[t] = (uint16_t)bits;
}

__global__ void k_h_export(const int32_t *page_gtile, int64_t nvox, const vox_t *coc, const uint32_t *occbits, int32_t *vox,
                           int32_t *d2, int32_t *cxyz, uint8_t *occ) {
  // every page, resident or parked, in MAP coordinates (ids decode relative to their voxel: no window involved)
  for (int64_t a = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; a < nvox; a += (int64_t)gridDim.x * blockDim.x) {
    const int64_t page = a / kPageVox;
    const int off = (int)(a % kPageVox);
    const int x = page_gtile[3 * page] * 16 + (off >> 9), y = page_gtile[3 * page + 1] * 16 + ((off >> 5) & 15),
              z = page_gtile[3 * page + 2] * 32 + (off & 31);
    const vox_t w = coc[a] & ((coc[a] == kUnobserved) ? 0xFFFFFFFFu : ~kAct);
    if (vox) vox[3 * a] = x, vox[3 * a + 1] = y, vox[3 * a + 2] = z;
    if (d2) d2[a] = (w == kUnobserved) ? -1 : ((w & kNoCoc) ? kD2Inf : dist2(1, x, y, z, w));
    if (cxyz) {
      int cx = FIESTA_HIP_UNDEFINED, cy = FIESTA_HIP_UNDEFINED, cz = FIESTA_HIP_UNDEFINED;
      if (!(w & kNoCoc)) unpack_coc(1, x, y, z, w, cx, cy, cz);
      cxyz[3 * a] = cx, cxyz[3 * a + 1] = cy, cxyz[3 * a + 2] = cz;
    }
    if (occ) occ[a] = hbit(occbits, a);
  }
}
// GetPointCloud / GetSliceMarker over the block store (src/ESDFMap.cpp:547-566, 657-677): every allocated voxel -- resident
// or parked -- inside the x/y update range (map voxels), the z test as in the dense build.  Order unspecified.
__global__ void k_h_point_cloud(Geom g, const int32_t *page_gtile, int64_t nrows, const uint32_t *occbits, int lox, int hix,
                                int loy, int hiy, int zlo, int zhi, float *out, unsigned long long cap, unsigned long long *count) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * blockDim.x) {
    uint32_t bits = occbits[r];  // one z-row of a page: 32 voxels
    if (!bits) continue;
    const int64_t page = r / HashMap::kPageRows;
    const int row = (int)(r % HashMap::kPageRows);
    const int x = page_gtile[3 * page] * 16 + (row >> 4), y = page_gtile[3 * page + 1] * 16 + (row & 15), z0 = page_gtile[3 * page + 2] * 32;
    if (x < lox || x > hix || y < loy || y > hiy) continue;
    uint32_t keep = 0;
    for (uint32_t b = bits; b; b &= b - 1) {
      const int k = __ffs(b) - 1;
      if (z0 + k >= zlo && z0 + k <= zhi) keep |= 1u << k;
    }
    if (!keep) continue;
    unsigned long long k = atomicAdd(count, (unsigned long long)__popc(keep));
    for (; keep; keep &= keep - 1, ++k) {
      if (k >= cap) continue;
      const int z = z0 + __ffs(keep) - 1;
      out[3 * k] = (float)((x + 0.5) * g.res + g.org[0]);
      out[3 * k + 1] = (float)((y + 0.5) * g.res + g.org[1]);
      out[3 * k + 2] = (float)((z + 0.5) * g.res + g.org[2]);
    }
  }
}
__global__ void k_h_slice_marker(Geom g, const int32_t *page_gtile, int64_t npages, const vox_t *coc, int lox, int hix, int loy,
                                 int hiy, int slice, double max_dist, double *xyz, float *rgba, unsigned long long cap,
                                 unsigned long long *count) {
  // one thread per (page, row): the page's z-range holds the slice or it does not (uniform per page)
  const int64_t n = npages * HashMap::kPageRows;
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
    const int64_t page = r / HashMap::kPageRows;
    const int row = (int)(r % HashMap::kPageRows), z0 = page_gtile[3 * page + 2] * 32;
    if (slice < z0 || slice >= z0 + 32) continue;
    const int x = page_gtile[3 * page] * 16 + (row >> 4), y = page_gtile[3 * page + 1] * 16 + (row & 15);
    if (x < lox || x > hix || y < loy || y > hiy) continue;
    const vox_t w = coc[page * kPageVox + row * 32 + (slice - z0)];
    if (w == kUnobserved || (w & kNoCoc)) continue;
    const double d = sqrt((double)dist2(1, x, y, slice, w & ~kAct)) * g.res;
    const unsigned long long k = atomicAdd(count, 1ull);
    if (k >= cap) continue;
    xyz[3 * k] = (x + 0.5) * g.res + g.org[0];
    xyz[3 * k + 1] = (y + 0.5) * g.res + g.org[1];
    xyz[3 * k + 2] = (slice + 0.5) * g.res + g.org[2];
    rainbow_rgba(d <= max_dist ? d / max_dist : 1, rgba + 4 * k);
  }
}
}  // namespace

// =====================================================================================================
void HashMap::use_device() const { FIESTA_HIP_CHECK(hipSetDevice(device_)); }

// (the host-side brick cache of the scalar queries: see host_brick below)
struct HashMap::HostBricks {
  static constexpr int kSlots = 1024, kDoubles = 4096 + 64;  // (4096 distances + 256 x 16 occupancy bits: 33 KB a slot, 34 MB)
  double *pool = nullptr;
  std::vector<int64_t> tag;
  std::vector<uint64_t> stamp, used;  // the field epoch a slot was fetched in; the tick it was last read at (two-way: the older one goes)
  uint64_t tick = 0;
  int64_t fetches = 0;
  ~HostBricks() {
    if (pool) (void)hipHostFree(pool);
  }
};
HashMap::HashMap(const fiesta_hip_config &cfg) {
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
  g.res_inv = 1 / cfg.resolution;
  for (int i = 0; i < 3; ++i) g.org[i] = cfg.origin[i];
  g.nx = g.ny = g.nz = kWin;
  g.nzw = kWin / 32;
  g.n = (int64_t)kWin * kWin * kWin;
  g.ox1 = g.oy1 = g.oz1 = kWin - 1;
  g.GX = g.GY = g.GZ = kWin;
  g.GZW = kWin / 32;
  g.gx0 = g.gy0 = g.gz0 = -kHalf;  // the window starts centred on map voxel 0 and follows the observations (ensure_window)
  g.wrap = 1;                      // ids are map coordinates modulo 1024, decoded relative to their voxel
  set_original_range();
#ifdef FIESTA_HIP_TUNING
  if (const char *e = getenv("FIESTA_HIP_PROF")) prof_ = atoi(e);
#endif
  if (cfg.update_engine < 0 || cfg.update_engine > 6) throw Error(FIESTA_HIP_ERR_INVALID, "unknown update_engine");
  update_engine_ = cfg.update_engine > 3 ? 0 : cfg.update_engine;  // (the transforms need a dense array: 2, 4, 5 and 6 mean 0 here)

  FIESTA_HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  FIESTA_HIP_CHECK(hipEventCreate(&ev0_));
  FIESTA_HIP_CHECK(hipEventCreate(&ev1_));
  FIESTA_HIP_CHECK(hipMalloc((void **)&dir_, kNTiles * sizeof(int32_t)));
  FIESTA_HIP_CHECK(hipMemsetAsync(dir_, 0xFF, kNTiles * sizeof(int32_t), stream_));
  uint32_t **zeroed[] = {&need_, &tile_epoch_, &cstamp_[0], &cstamp_[1], &tile_flag_[0], &tile_flag_[1], &tile_list_[0], &tile_list_[1]};
  for (uint32_t **p : zeroed) {
    FIESTA_HIP_CHECK(hipMalloc((void **)p, kNTiles * sizeof(uint32_t)));
    FIESTA_HIP_CHECK(hipMemsetAsync(*p, 0, kNTiles * sizeof(uint32_t), stream_));
  }
  FIESTA_HIP_CHECK(hipMalloc((void **)&counters_, C_COUNT * sizeof(unsigned long long)));
  FIESTA_HIP_CHECK(hipHostMalloc((void **)&h_counters_, C_COUNT * sizeof(unsigned long long)));
  FIESTA_HIP_CHECK(hipMemsetAsync(counters_, 0, C_COUNT * sizeof(unsigned long long), stream_));
  // reserve_size voxels up front (src/ESDFMap.cpp:141-145), at least a few pages
  ensure_pages(std::max<int64_t>(8, ((int64_t)cfg.reserve_size + kPageVox - 1) / kPageVox));
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
}

HashMap::~HashMap() {
  (void)hipSetDevice(device_);
  if (stream_) (void)hipStreamSynchronize(stream_);
  free_raycast_state();
  void *ptrs[] = {dir_, need_, tile_epoch_, cstamp_[0], cstamp_[1], tile_flag_[0], tile_flag_[1], tile_list_[0], tile_list_[1], counters_};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  if (h_counters_) (void)hipHostFree(h_counters_);
  delete lv_;
  delete bricks_;
  if (lv_done_) (void)hipEventDestroy(lv_done_);
  if (ev0_) (void)hipEventDestroy(ev0_);
  if (ev1_) (void)hipEventDestroy(ev1_);
  if (stream_) (void)hipStreamDestroy(stream_);
}

// Capacity for `need_total` pages; the pool doubles (IncreaseCapacity, src/ESDFMap.cpp:705-730). New pages start
// unobserved: distance -10000, no obstacle, log-odds 0, no pending observations.
void HashMap::ensure_pages(int64_t need_total) {
  if (need_total <= cap_pages_) return;
  int64_t cap = std::max<int64_t>(cap_pages_, 8);
  while (cap < need_total) cap *= 2;
  if (cap * kPageVox >= (1ll << 32)) throw Error(FIESTA_HIP_ERR_NOMEM, "page pool exceeds 2^32 voxels");
  const size_t keep_v = (size_t)cap_pages_ * kPageVox, keep_r = (size_t)cap_pages_ * kPageRows, keep_p = (size_t)cap_pages_;
  auto grow_exact = [&](auto &buf, size_t n, size_t keep) {
    if (n <= buf.cap) return;
    using T = std::remove_reference_t<decltype(*buf.p)>;
    T *q = nullptr;
    FIESTA_HIP_CHECK(hipMalloc((void **)&q, n * sizeof(T)));
    if (buf.p && keep) FIESTA_HIP_CHECK(hipMemcpyAsync(q, buf.p, keep * sizeof(T), hipMemcpyDeviceToDevice, stream_));
    FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
    if (buf.p) (void)hipFree(buf.p);
    buf.p = q;
    buf.cap = n;
  };
  grow_exact(coc_, (size_t)cap * kPageVox, keep_v);
  grow_exact(logodds_, (size_t)cap * kPageVox, keep_v);
  grow_exact(cnt_, (size_t)cap * kPageVox, keep_v);
  grow_exact(occbits_, (size_t)cap * kPageRows, keep_r);
  grow_exact(rbits_, (size_t)cap * kPageRows, keep_r);
  grow_exact(cbits_[0], (size_t)cap * kPageRows, keep_r);
  grow_exact(cbits_[1], (size_t)cap * kPageRows, keep_r);
  grow_exact(page_tile_, (size_t)cap, keep_p);
  grow_exact(page_gtile_, (size_t)cap * 3, keep_p * 3);
  grow_exact(page_fresh_, (size_t)cap, keep_p);
  FIESTA_HIP_CHECK(hipMemsetAsync(page_fresh_.p + keep_p, 0, (size_t)(cap - cap_pages_) * sizeof(uint32_t), stream_));
  const int64_t first = cap_pages_;
  cap_pages_ = cap;
  pristine_pages(first, cap - first);
}
// pages [first, first + count) as never-touched pool memory
void HashMap::pristine_pages(int64_t first, int64_t count) {
  if (count <= 0) return;
  const int64_t v0 = first * kPageVox, r0 = first * kPageRows, nv = count * kPageVox, nr = count * kPageRows;
  hipLaunchKernelGGL(k_h_fill<vox_t>, dim3(grid_for(nv, 256, 4096)), dim3(256), 0, stream_, coc_.p + v0, kUnobserved, nv);
  FIESTA_HIP_CHECK(hipGetLastError());
  FIESTA_HIP_CHECK(hipMemsetAsync(logodds_.p + v0, 0, nv * sizeof(double), stream_));
  FIESTA_HIP_CHECK(hipMemsetAsync(cnt_.p + v0, 0, nv * sizeof(unsigned long long), stream_));
  for (uint32_t *b : {occbits_.p, rbits_.p, cbits_[0].p, cbits_[1].p})
    FIESTA_HIP_CHECK(hipMemsetAsync(b + r0, 0, nr * sizeof(uint32_t), stream_));
}

// Raw dump (write) / load of the whole map state -- pool, directory, window, queues (checkpoint.hpp).
void HashMap::checkpoint(const char *path, bool write) {
  if (!write) ++field_epoch_;  // (a loaded field: the host-side brick cache of the scalar queries is stale)
  use_device();
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
  DevFile f(path, write, stream_);
  Geom hdr = g_;
  hdr.gx0 = hdr.gy0 = hdr.gz0 = 0;  // (the window origin is state, not identity)
  checkpoint_header(f, FIESTA_HIP_MODE_HASH, hdr);
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
  f.host(&g, sizeof(g));
  int64_t ur[6], pr[6];
  memcpy(ur, ur_, sizeof(ur));
  memcpy(pr, pr_, sizeof(pr));
  f.host(ur, sizeof(ur));
  f.host(pr, sizeof(pr));
  int64_t meta[4] = {npages_, moves_, dropped_host_, force_scan_ ? 1 : 0};
  f.host(meta, sizeof(meta));
  const size_t nt = c[C_TOUCHED], ni = c[C_INSERT], nd = c[C_DELETE];
  if (!write) {
    // validate everything the file claims before any state of this map is replaced (see DenseMap::checkpoint)
    Geom a = g, b = g_;
    for (Geom *q : {&a, &b}) {  // the window origin and the update ranges are state; the rest is identity
      q->gx0 = q->gy0 = q->gz0 = 0;
      q->wx0 = q->wy0 = q->wz0 = q->wx1 = q->wy1 = q->wz1 = q->px0 = q->py0 = q->pz0 = q->px1 = q->py1 = q->pz1 = 0;
    }
    if (memcmp(&a, &b, sizeof(Geom)) != 0) throw Error(FIESTA_HIP_ERR_INVALID, "checkpoint: geometry of the file does not match this map");
    if (meta[0] < 0 || meta[0] > (int64_t)kNTiles * 64) throw Error(FIESTA_HIP_ERR_INVALID, "checkpoint: page count of the file is out of bounds");
    const unsigned long long pv = (unsigned long long)meta[0] * kPageVox;
    if (nt > pv || ni > pv || nd > pv) throw Error(FIESTA_HIP_ERR_INVALID, "checkpoint: queue lengths of the file exceed its pages");
    const unsigned long long expect = f.position() + 11ull * sizeof(unsigned long long) + pv * (sizeof(vox_t) + sizeof(double) + sizeof(unsigned long long)) +
                                      (unsigned long long)meta[0] * kPageRows * sizeof(uint32_t) + (unsigned long long)meta[0] * (sizeof(int32_t) * 4 + sizeof(uint32_t)) +
                                      (unsigned long long)kNTiles * sizeof(int32_t) + (unsigned long long)(nt + ni + nd) * sizeof(uint32_t);
    if (f.file_size() != expect) throw Error(FIESTA_HIP_ERR_INVALID, "checkpoint: file size does not match its header (truncated or corrupt)");
    const int64_t had = npages_;
    ensure_pages(std::max<int64_t>(meta[0], 8));
    if (had > meta[0]) pristine_pages(meta[0], had - meta[0]);  // what this map held beyond the file's pages
    npages_ = meta[0];
    pp_ = pp;
    memcpy(ur_, ur, sizeof(ur_));
    memcpy(pr_, pr, sizeof(pr_));
    touched_.ensure(nt, stream_);
    ins_.ensure(ni, stream_);
    del_.ensure(nd, stream_);
  }
  const size_t nv = (size_t)npages_ * kPageVox, nr = (size_t)npages_ * kPageRows, np = (size_t)npages_;
  f.device(coc_.p, nv * sizeof(vox_t));
  f.device(logodds_.p, nv * sizeof(double));
  f.device(cnt_.p, nv * sizeof(unsigned long long));
  f.device(occbits_.p, nr * sizeof(uint32_t));
  f.device(page_tile_.p, np * sizeof(int32_t));
  f.device(page_gtile_.p, np * 3 * sizeof(int32_t));
  f.device(page_fresh_.p, np * sizeof(uint32_t));
  f.device(dir_, (size_t)kNTiles * sizeof(int32_t));
  f.device(touched_.p, nt * sizeof(uint32_t));
  f.device(ins_.p, ni * sizeof(uint32_t));
  f.device(del_.p, nd * sizeof(uint32_t));
  f.finish();
  if (write) return;
  c[C_LIST0] = c[C_LIST1] = c[C_LIST2] = 0;
  memcpy(h_counters_, c, sizeof(c));
  FIESTA_HIP_CHECK(hipMemcpyAsync(counters_, h_counters_, sizeof(c), hipMemcpyHostToDevice, stream_));
  for (uint32_t *p : {need_, tile_epoch_, cstamp_[0], cstamp_[1], tile_flag_[0], tile_flag_[1]})
    FIESTA_HIP_CHECK(hipMemsetAsync(p, 0, kNTiles * sizeof(uint32_t), stream_));
  g_ = g;
  moves_ = meta[1];
  dropped_host_ = meta[2];
  force_scan_ = meta[3] != 0;
  ptab_pages_built_ = -1;  // (page_gtile_ was replaced: the map-wide page table of the parked pages is rebuilt on demand)
  touched_upper_ = (int64_t)nt;
  host_queues_valid_ = false;
  shadow_vox_ = -1;
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
}

unsigned long long HashMap::read_counter(int which) {
  FIESTA_HIP_CHECK(hipMemcpyAsync(&h_counters_[which], &counters_[which], sizeof(unsigned long long), hipMemcpyDeviceToHost, stream_));
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
  return h_counters_[which];
}
void HashMap::zero_counters(int first, int n) {
  hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, stream_, &counters_[first], n);
  FIESTA_HIP_CHECK(hipGetLastError());
}
void HashMap::zero_counter(int which) { zero_counters(which, 1); }

void HashMap::set_prob_params(double p_hit, double p_miss, double p_min, double p_max, double p_occ) {
  auto logit = [](double x) { return std::log(x / (1 - x)); };
  pp_ = ProbParams{logit(p_hit), logit(p_miss), logit(p_min), logit(p_max), logit(p_occ)};
}
void HashMap::set_original_range() {  // src/ESDFMap.cpp:813-817: +-10000 voxels, i.e. everything
  for (int k = 0; k < 3; ++k) ur_[k] = pr_[k] = -(1ll << 40), ur_[3 + k] = pr_[3 + k] = 1ll << 40;
  refresh_range();
}
void HashMap::set_update_range(const double *mn, const double *mx, bool new_vec) {  // :792-810 (no clamping to a map box)
  const Geom &g = g_;
  if (new_vec) memcpy(pr_, ur_, sizeof(pr_));
  auto p2v = [&](double p, int i) {
    const double v = std::floor((p - g.org[i]) / g.res);
    return (int64_t)std::min(std::max(v, -1e12), 1e12);
  };
  for (int k = 0; k < 3; ++k) ur_[k] = p2v(mn[k], k), ur_[3 + k] = p2v(mx[k] - g.res / 2, k);
  refresh_range();
}
// the update boxes in window coordinates (clamped one voxel outside the window: "everything on that side")
void HashMap::refresh_range() {
  Geom &g = g_;
  const int o[3] = {g.gx0, g.gy0, g.gz0};
  auto w = [&](int64_t v, int k) { return (int)std::min<int64_t>(std::max<int64_t>(v - o[k], -1), kWin); };
  g.wx0 = w(ur_[0], 0), g.wy0 = w(ur_[1], 1), g.wz0 = w(ur_[2], 2);
  g.wx1 = w(ur_[3], 0), g.wy1 = w(ur_[4], 1), g.wz1 = w(ur_[5], 2);
  g.px0 = w(pr_[0], 0), g.py0 = w(pr_[1], 1), g.pz0 = w(pr_[2], 2);
  g.px1 = w(pr_[3], 0), g.py1 = w(pr_[4], 1), g.pz1 = w(pr_[5], 2);
}

// ---- the moving window ----
// A batch whose bounding box (map voxels, inclusive) does not fit the window recentres the window on it, on the axes
// where it does not fit; a box wider than the window is centred and loses its far ends (counted as dropped).
void HashMap::ensure_window(const int64_t lo[3], const int64_t hi[3]) {
  const int32_t o[3] = {g_.gx0, g_.gy0, g_.gz0}, tile[3] = {kTX, kTY, kTZ};
  int32_t n[3];
  bool move = false;
  for (int k = 0; k < 3; ++k) {
    n[k] = o[k];
    if (lo[k] > hi[k] || (lo[k] >= o[k] && hi[k] < (int64_t)o[k] + kWin)) continue;
    int64_t c = lo[k] + (hi[k] - lo[k]) / 2 - kHalf;                  // origin that centres the box ...
    c = (c >= 0 ? (c + tile[k] / 2) / tile[k] : -((-c + tile[k] / 2) / tile[k])) * tile[k];  // ... in whole tiles
    c = std::min<int64_t>(std::max<int64_t>(c, -(1ll << 30)), (1ll << 30) - kWin);
    n[k] = (int32_t)c;
    move |= n[k] != o[k];
  }
  if (move) move_window(n);
}
void HashMap::recentre(const int32_t centre[3]) {
  use_device();
  const int32_t tile[3] = {kTX, kTY, kTZ};
  int32_t n[3];
  for (int k = 0; k < 3; ++k) {
    const int64_t c = (int64_t)centre[k] - kHalf;
    n[k] = (int32_t)((c >= 0 ? (c + tile[k] / 2) / tile[k] : -((-c + tile[k] / 2) / tile[k])) * tile[k]);
  }
  if (n[0] != g_.gx0 || n[1] != g_.gy0 || n[2] != g_.gz0) move_window(n);
}
void HashMap::move_window(const int32_t origin[3]) {
  g_.gx0 = origin[0], g_.gy0 = origin[1], g_.gz0 = origin[2];
  refresh_range();
  FIESTA_HIP_CHECK(hipMemsetAsync(dir_, 0xFF, kNTiles * sizeof(int32_t), stream_));
  // everything indexed by window tile is between updates here: no list is live, stamps restart (epoch_/serial_ only grow)
  for (uint32_t *p : {need_, tile_epoch_, cstamp_[0], cstamp_[1], tile_flag_[0], tile_flag_[1]})
    FIESTA_HIP_CHECK(hipMemsetAsync(p, 0, kNTiles * sizeof(uint32_t), stream_));
  if (npages_)
    hipLaunchKernelGGL(k_h_rebuild, dim3(grid_for(npages_)), dim3(256), 0, stream_, g_, npages_, (const int32_t *)page_gtile_.p,
                       dir_, page_tile_.p, page_fresh_.p);
  FIESTA_HIP_CHECK(hipGetLastError());
  force_scan_ = true;
  ++moves_;
}
void HashMap::ensure_window_vox(const int32_t *vox, int64_t n, bool dev) {
  int64_t lo[3] = {INT64_MAX, INT64_MAX, INT64_MAX}, hi[3] = {INT64_MIN, INT64_MIN, INT64_MIN};
  if (!dev) {
    for (int64_t i = 0; i < n; ++i)
      for (int k = 0; k < 3; ++k) lo[k] = std::min<int64_t>(lo[k], vox[3 * i + k]), hi[k] = std::max<int64_t>(hi[k], vox[3 * i + k]);
  } else {
    stage_c_.ensure(6 * sizeof(int32_t), stream_);
    const int32_t init[6] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MIN, INT32_MIN, INT32_MIN};
    int32_t box[6];
    FIESTA_HIP_CHECK(hipMemcpyAsync(stage_c_.p, init, sizeof(init), hipMemcpyHostToDevice, stream_));
    hipLaunchKernelGGL(k_h_bbox, dim3(grid_for(n, 1024, 256)), dim3(1024), 0, stream_, vox, n, (int32_t *)stage_c_.p);
    FIESTA_HIP_CHECK(hipGetLastError());
    FIESTA_HIP_CHECK(hipMemcpyAsync(box, stage_c_.p, sizeof(box), hipMemcpyDeviceToHost, stream_));
    FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
    for (int k = 0; k < 3; ++k) lo[k] = box[k], hi[k] = box[3 + k];
  }
  ensure_window(lo, hi);
}

// Assign pages to every marked tile that has none yet.  Returns the "a voxel was outside the window" flag of the mark
// kernel (k_h_mark_vox), read in the same round trip as the number of fresh tiles.
bool HashMap::allocate_marked() {
  stage_d_.ensure((size_t)kNTiles * sizeof(uint32_t), stream_);
  zero_counter(C_SCRATCH);
  hipLaunchKernelGGL(k_h_collect, dim3(kNTiles / 256), dim3(256), 0, stream_, need_, (const int32_t *)dir_, (uint32_t *)stage_d_.p,
                     &counters_[C_SCRATCH]);
  FIESTA_HIP_CHECK(hipGetLastError());
  static_assert(C_OUTSIDE == C_SCRATCH + 1, "read together");
  FIESTA_HIP_CHECK(hipMemcpyAsync(&h_counters_[C_SCRATCH], &counters_[C_SCRATCH], 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream_));
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
  const int64_t k = (int64_t)h_counters_[C_SCRATCH];
  const bool outside = h_counters_[C_OUTSIDE] != 0;
  if (k == 0) return outside;
  ensure_pages(npages_ + k);
  hipLaunchKernelGGL(k_h_assign, dim3(grid_for(k)), dim3(256), 0, stream_, g_, (const uint32_t *)stage_d_.p, k, (int32_t)npages_,
                     dir_, page_tile_.p, page_gtile_.p);
  FIESTA_HIP_CHECK(hipGetLastError());
  npages_ += k;
  return outside;
}

void HashMap::observe_vox(const int32_t *vox, const int32_t *occ, int64_t n, int32_t *ret, bool dev) {
  use_device();
  if (n <= 0) return;
  const int32_t *dvox = vox, *docc = occ;
  if (!dev) {
    stage_a_.ensure(n * 3 * sizeof(int32_t), stream_);
    stage_b_.ensure(n * sizeof(int32_t), stream_);
    FIESTA_HIP_CHECK(hipMemcpyAsync(stage_a_.p, vox, n * 3 * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
    FIESTA_HIP_CHECK(hipMemcpyAsync(stage_b_.p, occ, n * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
    dvox = (const int32_t *)stage_a_.p;
    docc = (const int32_t *)stage_b_.p;
  }
  // host batches: the bounding box decides up front whether the window has to move; device batches: the mark kernel
  // reports a voxel outside the window (no extra round trip when there is none), then the box is reduced on the device
  if (!dev) ensure_window_vox(vox, n, false);
  zero_counter(C_OUTSIDE);
  hipLaunchKernelGGL(k_h_mark_vox, dim3(grid_for(n)), dim3(256), 0, stream_, g_, dvox, n, need_, dev ? &counters_[C_OUTSIDE] : nullptr);
  if (allocate_marked()) {
    const int64_t before = moves_;
    ensure_window_vox(dvox, n, true);
    if (moves_ != before) {
      zero_counter(C_OUTSIDE);
      hipLaunchKernelGGL(k_h_mark_vox, dim3(grid_for(n)), dim3(256), 0, stream_, g_, dvox, n, need_, (unsigned long long *)nullptr);
      allocate_marked();
    }
  }
  touched_upper_ = std::min<int64_t>(npages_ * kPageVox, touched_upper_ + n);
  touched_.ensure((size_t)touched_upper_, stream_, touched_.cap);
  hipLaunchKernelGGL(k_h_observe_vox, dim3(grid_for(n)), dim3(256), 0, stream_, g_, (const int32_t *)dir_, dvox, docc, n, cnt_.p,
                     touched_.p, counters_);
  FIESTA_HIP_CHECK(hipGetLastError());
  if (ret && !dev) {  // the reference returns its internal index (allocation-order dependent); what callers rely on is
              // "-10000 = rejected, otherwise a key that identifies the voxel" (include/Fiesta.h:221,253)
    for (int64_t i = 0; i < n; ++i) ret[i] = voxel_key(vox[3 * i], vox[3 * i + 1], vox[3 * i + 2]);
  }
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
}

void HashMap::observe_box(const int32_t *lo, const int32_t *hi, int occ) {
  use_device();
  {
    const int64_t l[3] = {lo[0], lo[1], lo[2]}, h[3] = {hi[0], hi[1], hi[2]};
    ensure_window(l, h);
  }
  // clip to the window (window coordinates); a box wider than the window loses its ends
  const int32_t o[3] = {g_.gx0, g_.gy0, g_.gz0};
  int a[3], b[3];
  int64_t asked = 1, kept = 1;
  for (int k = 0; k < 3; ++k) {
    a[k] = (int)std::max<int64_t>((int64_t)lo[k] - o[k], 0);
    b[k] = (int)std::min<int64_t>((int64_t)hi[k] - o[k], kWin - 1);
    asked *= std::max<int64_t>(0, (int64_t)hi[k] - lo[k] + 1);
    kept *= std::max<int64_t>(0, (int64_t)b[k] - a[k] + 1);
  }
  if (asked > kept) {  // the part of the box outside the addressable window is lost: count it (see C_DROPPED)
    dropped_host_ += asked - kept;
  }
  if (kept == 0) return;
  const int64_t ex = b[0] - a[0] + 1, ey = b[1] - a[1] + 1, ez = b[2] - a[2] + 1;
  const int64_t ntiles = (int64_t)((b[0] >> 4) - (a[0] >> 4) + 1) * ((b[1] >> 4) - (a[1] >> 4) + 1) * ((b[2] >> 5) - (a[2] >> 5) + 1);
  hipLaunchKernelGGL(k_h_mark_box, dim3(grid_for(ntiles)), dim3(256), 0, stream_, g_, a[0], a[1], a[2], b[0], b[1], b[2], need_);
  allocate_marked();
  touched_upper_ = std::min<int64_t>(npages_ * kPageVox, touched_upper_ + ex * ey * ez);
  touched_.ensure((size_t)touched_upper_, stream_, touched_.cap);
  hipLaunchKernelGGL(k_h_observe_box, dim3(grid_for(ex * ey * ((ez + 63) / 64), 16, 4096)), dim3(1024), 0, stream_, g_,
                     (const int32_t *)dir_, a[0], a[1], a[2], (int)ex, (int)ey, (int)ez, occ, cnt_.p, touched_.p, counters_);
  FIESTA_HIP_CHECK(hipGetLastError());
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
}

void HashMap::snapshot_save() {
  use_device();
  shadow_.ensure((size_t)npages_ * kPageVox, stream_);
  FIESTA_HIP_CHECK(hipMemcpyAsync(shadow_.p, coc_.p, (size_t)npages_ * kPageVox * sizeof(vox_t), hipMemcpyDeviceToDevice, stream_));
  shadow_vox_ = npages_ * kPageVox;
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
}

int64_t HashMap::snapshot_count_updated() {
  use_device();
  if (shadow_vox_ < 0) throw Error(FIESTA_HIP_ERR_STATE, "snapshot_count_updated: no snapshot saved");
  zero_counter(C_SCRATCH);
  const int64_t n_now = npages_ * kPageVox;
  if (n_now)
    hipLaunchKernelGGL(k_h_count_updated, dim3(grid_for(n_now, 256, 8192)), dim3(256), 0, stream_, g_, (const int32_t *)dir_,
                       (const int32_t *)page_tile_.p, (const vox_t *)shadow_.p, shadow_vox_, (const vox_t *)coc_.p, n_now,
                       (const uint32_t *)occbits_.p, &counters_[C_SCRATCH]);
  FIESTA_HIP_CHECK(hipGetLastError());
  return (int64_t)read_counter(C_SCRATCH);
}

void HashMap::observe_pos(const double *pos, const int32_t *occ, int64_t n, int32_t *ret) {
  use_device();
  if (n <= 0) return;
  {
    int64_t lo[3] = {INT64_MAX, INT64_MAX, INT64_MAX}, hi[3] = {INT64_MIN, INT64_MIN, INT64_MIN};
    for (int64_t i = 0; i < n; ++i) {
      if (occ[i] != 0 && occ[i] != 1) continue;
      double v[3];
      bool sane = true;
      for (int k = 0; k < 3; ++k) v[k] = std::floor((pos[3 * i + k] - g_.org[k]) / g_.res), sane &= std::fabs(v[k]) < 1e9;
      if (!sane) continue;  // (NaN / far-away garbage never moves the window; the kernel drops it)
      for (int k = 0; k < 3; ++k) lo[k] = std::min(lo[k], (int64_t)v[k]), hi[k] = std::max(hi[k], (int64_t)v[k]);
    }
    ensure_window(lo, hi);
  }
  stage_a_.ensure(n * 3 * sizeof(double), stream_);
  stage_b_.ensure(n * sizeof(int32_t), stream_);
  FIESTA_HIP_CHECK(hipMemcpyAsync(stage_a_.p, pos, n * 3 * sizeof(double), hipMemcpyHostToDevice, stream_));
  FIESTA_HIP_CHECK(hipMemcpyAsync(stage_b_.p, occ, n * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
  hipLaunchKernelGGL(k_h_mark_pos, dim3(grid_for(n)), dim3(256), 0, stream_, g_, (const double *)stage_a_.p,
                     (const int32_t *)stage_b_.p, n, need_);
  allocate_marked();
  touched_upper_ = std::min<int64_t>(npages_ * kPageVox, touched_upper_ + n);
  touched_.ensure((size_t)touched_upper_, stream_, touched_.cap);
  hipLaunchKernelGGL(k_h_observe_pos, dim3(grid_for(n)), dim3(256), 0, stream_, g_, (const int32_t *)dir_, (const double *)stage_a_.p,
                     (const int32_t *)stage_b_.p, n, cnt_.p, touched_.p, counters_);
  FIESTA_HIP_CHECK(hipGetLastError());
  if (ret) {
    for (int64_t i = 0; i < n; ++i) {
      const double *p = pos + 3 * i;
      const bool ok = occ[i] == 0 || occ[i] == 1;
      ret[i] = ok ? voxel_key((int)std::floor((p[0] - g_.org[0]) / g_.res), (int)std::floor((p[1] - g_.org[1]) / g_.res),
                              (int)std::floor((p[2] - g_.org[2]) / g_.res))
                  : FIESTA_HIP_UNDEFINED;
    }
  }
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
}

bool HashMap::check_update() {
  use_device();
  if (touched_upper_ == 0) return false;
  return read_counter(C_TOUCHED) != 0;
}

bool HashMap::update_occupancy(bool global_map, int64_t *n_ins, int64_t *n_del) {
  use_device();
  // ONE host synchronisation per call (none when nothing was observed): the host keeps the queue lengths of its last
  // read and an upper bound of the touched list; the fusion kernel reads the exact length itself.
  if (!host_queues_valid_) {
    FIESTA_HIP_CHECK(hipMemcpyAsync(&h_counters_[C_INSERT], &counters_[C_INSERT], 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream_));
    FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
    host_ni_ = h_counters_[C_INSERT], host_nd_ = h_counters_[C_DELETE];
    host_queues_valid_ = true;
  }
  const unsigned long long nt = (unsigned long long)touched_upper_;
  if (nt) {
    ++field_epoch_;  // (occupancy and first observations may change: the host-side brick cache of the scalar queries is stale)
    ins_.ensure(host_ni_ + nt, stream_, host_ni_);
    del_.ensure(host_nd_ + nt, stream_, host_nd_);
    hipLaunchKernelGGL(k_h_fuse, dim3(grid_for((int64_t)nt, 256, 8192)), dim3(256), 0, stream_, g_, pp_, global_map ? 1 : 0,
                       (const int32_t *)page_tile_.p, (const uint32_t *)touched_.p, (int64_t)-1, cnt_.p, logodds_.p, coc_.p,
                       occbits_.p, ins_.p, del_.p, counters_);
    FIESTA_HIP_CHECK(hipGetLastError());
    zero_counter(C_TOUCHED);
    touched_upper_ = 0;
    // (C_INSERT .. C_DROPPED: the queue lengths, and the lost-observation count UpdateESDF reports without a read of its own)
    static_assert(C_DROPPED == C_INSERT + 4, "counter layout");
    FIESTA_HIP_CHECK(hipMemcpyAsync(&h_counters_[C_INSERT], &counters_[C_INSERT], 5 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream_));
    FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
    host_ni_ = h_counters_[C_INSERT], host_nd_ = h_counters_[C_DELETE];
  }
  if (n_ins) *n_ins = (int64_t)host_ni_;
  if (n_del) *n_del = (int64_t)host_nd_;
  return host_ni_ != 0 || host_nd_ != 0;
}

void HashMap::run_rounds(fiesta_hip_stats *st, uint32_t first_count) {
  int cur = 0;
  uint32_t ncur = first_count;
  int64_t rounds = 0;
  double relax_ms = 0;
  TileGrid tg{kTX, kTY, kNTX, kNTY, kNTZ};
  serial_ += 2;
  // one round: the active-tile list `cur` (length known to the host, or read on the device) -> list `cur ^ 1`
  int64_t launches = 0;
  auto launch = [&](const int cur_list) {
    const int nxt = cur_list ^ 1;
    // list counters rotate (dense_map.hpp, C_LIST0): no memset between rounds
    const int c_in = C_LIST0 + (int)(launches % 3), c_out = C_LIST0 + (int)((launches + 1) % 3), c_zero = C_LIST0 + (int)((launches + 2) % 3);
    const unsigned long long *n_dev = &counters_[c_in];
    ++launches;
    ++serial_;
    RelaxQArgs a;
    a.g = g_;
    a.tg = tg;
    a.coc = coc_.p;
    a.rbits = rbits_.p;
    a.tile_epoch = tile_epoch_;
    a.epoch = epoch_;
    a.cbits_prev = cbits_[(serial_ - 1) & 1].p;
    a.cbits_cur = cbits_[serial_ & 1].p;
    a.cstamp_prev = cstamp_[(serial_ - 1) & 1];
    a.cstamp_cur = cstamp_[serial_ & 1];
    a.serial = serial_;
    a.list_cur = tile_list_[cur_list];
    a.n_cur = 0;
    a.n_cur_dev = n_dev;
    a.flag_cur = tile_flag_[cur_list];
    a.flag_next = tile_flag_[nxt];
    a.list_next = tile_list_[nxt];
    a.count_next = &counters_[c_out];
    a.count_zero = &counters_[c_zero];
    a.counters = counters_;
    a.prof = prof_;
    a.dir = dir_;
    a.spatial = 0;
    // (a round of a chain: one work-group per CU striding over the list -- a tile's keys fill a CU's LDS, more work-groups
    //  only queue, and a round that finds its list empty should cost as little as a launch can)
    const uint32_t blocks = 256u;
    hipLaunchKernelGGL((k_relax_q<kTX, kTY, 1024, true>), dim3(blocks), dim3(1024), 0, stream_, a);
    FIESTA_HIP_CHECK(hipGetLastError());
  };
  // Rounds go out in chains: every round reads the length of its list on the device and does nothing once a
  // predecessor activated no tile -- one host round trip per chain (a streaming frame's update is a handful of short
  // kernels; the round trips were most of its time).  The first chain is as long as the previous update's rounds plus
  // one.  first_count == 0xFFFFFFFF: nobody read the first list's length.
  if (ncur == 0xFFFFFFFFu) ncur = 256;
  bool first = true;
  while (ncur) {
    const int chain = first ? std::min(std::max(chain_hint_, 2), 12) : 4;
    first = false;
    FIESTA_HIP_CHECK(hipEventRecord(ev0_, stream_));
    for (int k = 0; k < chain; ++k) {
      launch(cur);
      cur ^= 1;
    }
    FIESTA_HIP_CHECK(hipEventRecord(ev1_, stream_));
    // (the whole block: an update that ends with this chain has its statistics with it)
    FIESTA_HIP_CHECK(hipMemcpyAsync(h_counters_, counters_, C_COUNT * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream_));
    FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
    ncur = (uint32_t)h_counters_[C_LIST0 + (int)(launches % 3)];
    rounds = (int64_t)h_counters_[C_ROUNDS];
    float ms = 0;
    FIESTA_HIP_CHECK(hipEventElapsedTime(&ms, ev0_, ev1_));
    relax_ms += ms;
  }
  // (a chain that ends on a round with work leaves that round's input counter set; trailing idle rounds clear it)
  if (h_counters_[C_LIST0] || h_counters_[C_LIST1] || h_counters_[C_LIST2]) zero_counters(C_LIST0, 3);
  chain_hint_ = (int)rounds + 1;
  if (st) {
    st->rounds = rounds;
    st->relax_ms = relax_ms;
    st->relax_launches = rounds;
  }
}

void HashMap::level_tuning(int grid_groups, long long spin_limit) {
  if (!lv_) lv_ = new LevelEngine;
  if (grid_groups >= 0) lv_->grid_groups = grid_groups;
  if (spin_limit >= 0) lv_->spin_limit = (uint32_t)spin_limit;
}

int HashMap::level_trace(uint32_t *out48) const {
  memset(out48, 0, 48 * sizeof(uint32_t));
  if (!lv_ || !lv_->h_ctl) return 0;
  memcpy(out48, lv_->h_ctl->trace, 48 * sizeof(uint32_t));
  return (int)lv_->h_ctl->level;
}

// UpdateESDF by the level engine (level_kernels.hpp; as DenseMap::run_levels).  false: the update did not fit its lists,
// the field carries frontier tags and the tile list is set up for the frontier rounds.
bool HashMap::run_levels(fiesta_hip_stats *st, unsigned long long ni, unsigned long long nd, bool scan) {
  if (!lv_) lv_ = new LevelEngine;
  if (!lv_done_) FIESTA_HIP_CHECK(hipEventCreate(&lv_done_));
  lv_->ensure(update_engine_ == 3 ? (1u << 24) : (1u << 20), stream_);
  lv_->begin();
  LevelArgs a = lv_->args(coc_.p, counters_, false);
  PagedSpace sp{g_, occbits_.p, dir_, page_tile_.p, lv_box(g_)};
  TileGrid tg{kTX, kTY, kNTX, kNTY, kNTZ};
  const bool win_all = g_.wx0 <= 0 && g_.wy0 <= 0 && g_.wz0 <= 0 && g_.wx1 >= kWin - 1 && g_.wy1 >= kWin - 1 && g_.wz1 >= kWin - 1;
  const int64_t nvox = npages_ * kPageVox;
  if (nvox >= (1ll << 32)) throw Error(FIESTA_HIP_ERR_STATE, "page pool too large for the level engine's 32-bit addresses");
  FIESTA_HIP_CHECK(hipEventRecord(ev0_, stream_));
  if (ni) {
    hipLaunchKernelGGL((k_level_seed_insert<PagedSpace>), dim3(grid_for((int64_t)ni)), dim3(256), 0, stream_, sp, a,
                       (const uint32_t *)ins_.p, (int64_t)ni);
    FIESTA_HIP_CHECK(hipGetLastError());
  }
  if (scan) {
    hipLaunchKernelGGL(k_h_invalidate<true>, dim3(grid_for(nvox / 16 + 1, 256, 16384)), dim3(256), 0, stream_, g_, (const int32_t *)dir_,
                       (const int32_t *)page_tile_.p, (const uint32_t *)page_fresh_.p, nvox, coc_.p, (const uint32_t *)occbits_.p,
                       tile_flag_[0], tile_list_[0], &counters_[C_LIST0], counters_, a);
    FIESTA_HIP_CHECK(hipGetLastError());
    if (force_scan_) FIESTA_HIP_CHECK(hipMemsetAsync(page_fresh_.p, 0, (size_t)npages_ * sizeof(uint32_t), stream_));
    force_scan_ = false;
    if (!win_all) {
      hipLaunchKernelGGL((k_level_outside<PagedSpace>), dim3(64), dim3(256), 0, stream_, sp, a);
      FIESTA_HIP_CHECK(hipGetLastError());
    }
    if (win_all) {  // (see DenseMap::run_levels)
      hipLaunchKernelGGL((k_level_fill<PagedSpace, 1024>), dim3(1), dim3(1024), 0, stream_, sp, a);
      FIESTA_HIP_CHECK(hipGetLastError());
    }
  }
  host_ni_ = host_nd_ = 0;  // (k_level_run clears the device's queue counters)
  int64_t launches = 0;
  const LevelEngine::Outcome how = lv_->run(sp, a, stream_, lv_done_, update_engine_ == 3, ni + nd <= (unsigned long long)LevelEngine::kTiny, &launches);
  const LevelCtl &c = *lv_->h_ctl;
  if (how == LevelEngine::kDone) {
    if (st) {
      float ms = 0;
      FIESTA_HIP_CHECK(hipEventElapsedTime(&ms, ev0_, lv_done_));
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
  if (how == LevelEngine::kAbort)  // (see DenseMap::run_levels)
    FIESTA_HIP_CHECK(hipMemsetAsync(&lv_->ctl->overflow, 0, sizeof(uint32_t), stream_));
  if (how == LevelEngine::kHandOver || how == LevelEngine::kAbort) {
    a.level = c.level;  // (phase A of the level the engine stopped in front of: see k_level_list_to_tiles)
    // (work-groups in proportion to the frontier: a delete on a surface orphans 10^5 voxels and level 0 is handed over whole)
    const uint32_t n_over = std::min(c.n[c.level % 3u], lv_->cap);
    hipLaunchKernelGGL((k_level_pull<PagedSpace>), dim3(std::min(std::max(n_over / 64u, 64u), 8192u)), dim3(256), 0, stream_, sp, a);
    hipLaunchKernelGGL((k_level_list_to_tiles<PagedSpace>), dim3(std::min(std::max(n_over / 256u, 16u), 4096u)), dim3(256), 0, stream_, sp, a, tg,
                       tile_flag_[0], tile_list_[0], &counters_[C_LIST0]);
    FIESTA_HIP_CHECK(hipGetLastError());
    if (how == LevelEngine::kHandOver) return false;
  }
  if (scan) {  // (see DenseMap::run_levels; `scan`, not nd: the first scan also ran for a window move)
    hipLaunchKernelGGL(k_h_invalidate<false>, dim3(grid_for(nvox / 16 + 1, 256, 16384)), dim3(256), 0, stream_, g_, (const int32_t *)dir_,
                       (const int32_t *)page_tile_.p, (const uint32_t *)page_fresh_.p, nvox, coc_.p, (const uint32_t *)occbits_.p,
                       tile_flag_[0], tile_list_[0], &counters_[C_LIST0], counters_, LevelArgs{});
    FIESTA_HIP_CHECK(hipGetLastError());
  }
  hipLaunchKernelGGL((k_level_to_tiles<PagedSpace>), dim3(grid_for(nvox, 256, 8192)), dim3(256), 0, stream_, sp, a, nvox, tg,
                     tile_flag_[0], tile_list_[0], &counters_[C_LIST0]);
  FIESTA_HIP_CHECK(hipGetLastError());
  return false;
}

void HashMap::update_esdf(fiesta_hip_stats *st) {  // UpdateESDF (src/ESDFMap.cpp:273-398)
  use_device();
  const auto h0 = std::chrono::steady_clock::now();
  // the queue lengths as UpdateOccupancy read them (nothing else appends); C_DROPPED comes with the statistics
  if (!host_queues_valid_) {
    FIESTA_HIP_CHECK(hipMemcpyAsync(&h_counters_[C_INSERT], &counters_[C_INSERT], 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream_));
    FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
    host_ni_ = h_counters_[C_INSERT], host_nd_ = h_counters_[C_DELETE];
    host_queues_valid_ = true;
  }
  const unsigned long long ni = host_ni_, nd = host_nd_;
  if (st) {
    memset(st, 0, sizeof(*st));
    st->inserted = (int64_t)ni;
    st->deleted = (int64_t)nd;
  }
  if (ni || nd || force_scan_) {
    ++field_epoch_;  // (the host-side brick cache of the scalar queries is stale from here on)
    ++epoch_;
    static_assert(C_LIST2 == C_LIST0 + 2 && C_INVALIDATED == C_LIST2 + 1, "counter layout");
    const auto d00 = std::chrono::steady_clock::now();
    bool levels_fell_back = false;
    // (the inserts ARE level 0: more of them than one work-group carries and the level engine would only hand the update on)
    if (update_engine_ != 1 && (update_engine_ == 3 || (ni + nd <= (unsigned long long)small_update_ && ni <= (unsigned long long)LevelEngine::kInsertCap))) {
      // (the level engine keeps its statistics in its own control block: no counter reset, no read-back of counters --
      //  dropped observations are reported from the last value the host saw plus what it clipped itself)
      if (run_levels(st, ni, nd, nd || force_scan_)) {
        if (st) {
          st->device_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - d00).count();
          st->dropped_observations = (int64_t)h_counters_[C_DROPPED] + dropped_host_;
          st->host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
        }
        return;
      }
      levels_fell_back = true;  // (seeds and scans are done; the field carries tags, the tile list is set up)
      zero_counters(C_INVALIDATED, C_COUNT - C_INVALIDATED);
    } else {
      zero_counters(C_LIST0, C_COUNT - C_LIST0);
    }
    if (ni && !levels_fell_back) {
      hipLaunchKernelGGL(k_h_seed_insert, dim3(grid_for((int64_t)ni)), dim3(256), 0, stream_, g_, (const int32_t *)page_tile_.p,
                         (const uint32_t *)ins_.p, (int64_t)ni, coc_.p, (const uint32_t *)occbits_.p, tile_flag_[0], tile_list_[0],
                         &counters_[C_LIST0]);
      FIESTA_HIP_CHECK(hipGetLastError());
    }
    if ((nd || force_scan_) && !levels_fell_back) {  // (a window move: obstacles left the window, parked pages came back)
      const int64_t nvox = npages_ * kPageVox;
      hipLaunchKernelGGL(k_h_invalidate<false>, dim3(grid_for(nvox / 16 + 1, 256, 16384)), dim3(256), 0, stream_, g_, (const int32_t *)dir_,
                         (const int32_t *)page_tile_.p, (const uint32_t *)page_fresh_.p, nvox, coc_.p, (const uint32_t *)occbits_.p, tile_flag_[0], tile_list_[0],
                         &counters_[C_LIST0], counters_, LevelArgs{});
      FIESTA_HIP_CHECK(hipGetLastError());
      if (force_scan_) FIESTA_HIP_CHECK(hipMemsetAsync(page_fresh_.p, 0, (size_t)npages_ * sizeof(uint32_t), stream_));
      force_scan_ = false;
    }
    static_assert(C_DELETE == C_INSERT + 1, "counter layout");
    zero_counters(C_INSERT, 2);  // both queues are drained
    host_ni_ = host_nd_ = 0;
    const auto d0 = std::chrono::steady_clock::now();
    run_rounds(st, 0xFFFFFFFFu);  // (the chain of rounds finds the seeded tiles' count on the device and brings the counters back)
    if (st) {
      st->invalidated = (int64_t)h_counters_[C_INVALIDATED];
      st->sweeps = (int64_t)h_counters_[C_SWEEPS];
      st->voxel_writes = (int64_t)h_counters_[C_WRITES];
      st->tile_visits = (int64_t)h_counters_[C_VISITS];
      for (int k = 0; k < 8; ++k) st->prof[k] = (int64_t)h_counters_[C_PROF0 + k];
      st->device_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - d0).count();
      st->dropped_observations = (int64_t)h_counters_[C_DROPPED] + dropped_host_;
    }
  } else if (st) {
    st->dropped_observations = (int64_t)read_counter(C_DROPPED) + dropped_host_;
  }
  if (st) st->host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
}

// ---- queries ----
// The map-wide page table of the query kernels (PageTable): sorted on the host whenever pages were added since it was built
// (a few thousand keys at most; pages are never freed and a page's map tile never changes).
PageTable HashMap::page_table() {
  if (ptab_pages_built_ != npages_) {
    std::vector<int32_t> gt((size_t)npages_ * 3);
    if (npages_) {
      FIESTA_HIP_CHECK(hipMemcpyAsync(gt.data(), page_gtile_.p, gt.size() * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
      FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
    }
    std::vector<std::pair<unsigned long long, int32_t>> e((size_t)npages_);
    for (int64_t p = 0; p < npages_; ++p) e[p] = {tile_key(gt[3 * p], gt[3 * p + 1], gt[3 * p + 2]), (int32_t)p};
    std::sort(e.begin(), e.end());
    std::vector<unsigned long long> keys(e.size());
    std::vector<int32_t> pages(e.size());
    for (size_t i = 0; i < e.size(); ++i) keys[i] = e[i].first, pages[i] = e[i].second;
    ptab_keys_.ensure(std::max<size_t>(1, keys.size()), stream_);
    ptab_pages_.ensure(std::max<size_t>(1, pages.size()), stream_);
    if (!keys.empty()) {
      FIESTA_HIP_CHECK(hipMemcpyAsync(ptab_keys_.p, keys.data(), keys.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, stream_));
      FIESTA_HIP_CHECK(hipMemcpyAsync(ptab_pages_.p, pages.data(), pages.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
      FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));  // (the host vectors go out of scope)
    }
    ptab_pages_built_ = npages_;
  }
  return PageTable{ptab_keys_.p, ptab_pages_.p, (int)ptab_pages_built_};
}

// ---- host-side brick cache of the scalar queries (the dense map's, dense_map.hip: HostBricks, for the paged map) -----------
// The reference's GetDistance is a hash lookup + an array read (src/ESDFMap.cpp:467-479, 732-765); a planner calls it one position
// at a time.  Through copy-in / launch / copy-out / synchronise such a call cost tens of microseconds.  Now the
// distances (f64, exactly what the query kernels compute) and occupancy bits of a 16^3-voxel brick are fetched on first touch by
// one small kernel into pinned host memory, and every further scalar query into that brick is a host read.  Bricks are keyed by
// MAP coordinates (the window may move; parked pages answer too); whatever can change the field bumps an epoch.
int64_t HashMap::host_brick_fetches() const { return bricks_ ? bricks_->fetches : 0; }
const double *HashMap::host_brick(int vx, int vy, int vz) {
  if (!bricks_) {
    bricks_ = new HostBricks;
    FIESTA_HIP_CHECK(hipHostMalloc((void **)&bricks_->pool, (size_t)HostBricks::kSlots * HostBricks::kDoubles * sizeof(double)));
    bricks_->tag.assign(HostBricks::kSlots, INT64_MIN);
    bricks_->stamp.assign(HostBricks::kSlots, 0);
    bricks_->used.assign(HostBricks::kSlots, 0);
  }
  const int bx = vx >> 4, by = vy >> 4, bz = vz >> 4;  // (arithmetic shifts: map coordinates may be negative)
  const int64_t id = ((int64_t)(bx + (1 << 20)) << 42) | ((int64_t)(by + (1 << 20)) << 21) | (int64_t)(bz + (1 << 20));
  const uint32_t hsh = ((uint32_t)bx * 0x9E3779B1u) ^ ((uint32_t)by * 0x85EBCA77u) ^ ((uint32_t)bz * 0xC2B2AE3Du);
  // two-way set associative: a planner's working set of a few dozen bricks must not thrash on one unlucky pair (the eight corners
  // of a trilinear query straddle up to eight bricks, asked for in turn)
  const int s0 = (int)((hsh ^ (hsh >> 15)) % (uint32_t)(HostBricks::kSlots / 2)) * 2, s1 = s0 + 1;
  int slot = s0;
  if (bricks_->tag[s0] == id && bricks_->stamp[s0] == field_epoch_) slot = s0;
  else if (bricks_->tag[s1] == id && bricks_->stamp[s1] == field_epoch_) slot = s1;
  else if (bricks_->stamp[s0] != field_epoch_) slot = s0;   // (a stale slot goes first,
  else if (bricks_->stamp[s1] != field_epoch_) slot = s1;
  else slot = bricks_->used[s0] <= bricks_->used[s1] ? s0 : s1;   //  else the one read longer ago)
  bricks_->used[slot] = ++bricks_->tick;
  double *b = bricks_->pool + (size_t)slot * HostBricks::kDoubles;
  if (bricks_->tag[slot] != id || bricks_->stamp[slot] != field_epoch_) {
    use_device();
    hipLaunchKernelGGL(k_h_fetch_brick, dim3(1), dim3(256), 0, stream_, g_, (const int32_t *)dir_, page_table(), (const vox_t *)coc_.p,
                       (const uint32_t *)occbits_.p, bx, by, bz, b);
    FIESTA_HIP_CHECK(hipGetLastError());
    FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
    bricks_->tag[slot] = id, bricks_->stamp[slot] = field_epoch_;
    ++bricks_->fetches;
  }
  return b;
}
double HashMap::host_distance(int vx, int vy, int vz) { return host_brick(vx, vy, vz)[((vx & 15) << 8) | ((vy & 15) << 4) | (vz & 15)]; }
int HashMap::host_occ(int vx, int vy, int vz) {
  const double *b = host_brick(vx, vy, vz);
  const int row = ((vx & 15) << 4) | (vy & 15);
  return (int)((reinterpret_cast<const uint16_t *>(b + 4096)[row] >> (vz & 15)) & 1u);
}

void HashMap::get_distance_vox(const int32_t *vox, int64_t n, double *out) {
  if (n > 0 && n <= kHostQueries) {  // a scalar call of the drop-in class: the host-side brick cache
    for (int64_t i = 0; i < n; ++i) out[i] = host_distance(vox[3 * i], vox[3 * i + 1], vox[3 * i + 2]);
    return;
  }
  use_device();
  if (n <= 0) return;
  stage_a_.ensure(n * 3 * sizeof(int32_t), stream_);
  stage_c_.ensure(n * sizeof(double), stream_);
  FIESTA_HIP_CHECK(hipMemcpyAsync(stage_a_.p, vox, n * 3 * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
  hipLaunchKernelGGL(k_h_query_dist, dim3(grid_for(n)), dim3(256), 0, stream_, g_, (const int32_t *)dir_, page_table(), (const vox_t *)coc_.p,
                     (const int32_t *)stage_a_.p, (const double *)nullptr, n, (double *)stage_c_.p);
  FIESTA_HIP_CHECK(hipMemcpyAsync(out, stage_c_.p, n * sizeof(double), hipMemcpyDeviceToHost, stream_));
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
}
void HashMap::get_distance_pos(const double *pos, int64_t n, double *out) {
  if (n > 0 && n <= kHostQueries) {
    for (int64_t i = 0; i < n; ++i)   // (Pos2Vox as k_h_query_dist does it)
      out[i] = host_distance((int)floor((pos[3 * i] - g_.org[0]) / g_.res), (int)floor((pos[3 * i + 1] - g_.org[1]) / g_.res),
                             (int)floor((pos[3 * i + 2] - g_.org[2]) / g_.res));
    return;
  }
  use_device();
  if (n <= 0) return;
  stage_a_.ensure(n * 3 * sizeof(double), stream_);
  stage_c_.ensure(n * sizeof(double), stream_);
  FIESTA_HIP_CHECK(hipMemcpyAsync(stage_a_.p, pos, n * 3 * sizeof(double), hipMemcpyHostToDevice, stream_));
  hipLaunchKernelGGL(k_h_query_dist, dim3(grid_for(n)), dim3(256), 0, stream_, g_, (const int32_t *)dir_, page_table(), (const vox_t *)coc_.p,
                     (const int32_t *)nullptr, (const double *)stage_a_.p, n, (double *)stage_c_.p);
  FIESTA_HIP_CHECK(hipMemcpyAsync(out, stage_c_.p, n * sizeof(double), hipMemcpyDeviceToHost, stream_));
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
}
void HashMap::get_dist_grad(const double *pos, int64_t n, double *dist, double *grad) {
  if (n > 0 && n <= kHostQueries) {
    auto corner = [&](int vx, int vy, int vz) { return host_distance(vx, vy, vz); };
    for (int64_t i = 0; i < n; ++i) dist[i] = h_trilinear(g_, corner, pos + 3 * i, grad ? grad + 3 * i : nullptr);
    return;
  }
  use_device();
  if (n <= 0) return;
  stage_a_.ensure(n * 3 * sizeof(double), stream_);
  stage_b_.ensure(n * 3 * sizeof(double), stream_);
  stage_c_.ensure(n * sizeof(double), stream_);
  FIESTA_HIP_CHECK(hipMemcpyAsync(stage_a_.p, pos, n * 3 * sizeof(double), hipMemcpyHostToDevice, stream_));
  hipLaunchKernelGGL(k_h_query_trilinear, dim3(grid_for(n)), dim3(256), 0, stream_, g_, (const int32_t *)dir_, page_table(), (const vox_t *)coc_.p,
                     (const double *)stage_a_.p, n, (double *)stage_c_.p, (double *)stage_b_.p);
  FIESTA_HIP_CHECK(hipMemcpyAsync(dist, stage_c_.p, n * sizeof(double), hipMemcpyDeviceToHost, stream_));
  if (grad) FIESTA_HIP_CHECK(hipMemcpyAsync(grad, stage_b_.p, n * 3 * sizeof(double), hipMemcpyDeviceToHost, stream_));
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
}
void HashMap::get_occupancy_vox(const int32_t *vox, int64_t n, int32_t *out) {
  if (n > 0 && n <= kHostQueries) {
    for (int64_t i = 0; i < n; ++i) out[i] = host_occ(vox[3 * i], vox[3 * i + 1], vox[3 * i + 2]);
    return;
  }
  use_device();
  if (n <= 0) return;
  stage_a_.ensure(n * 3 * sizeof(int32_t), stream_);
  stage_c_.ensure(n * sizeof(int32_t), stream_);
  FIESTA_HIP_CHECK(hipMemcpyAsync(stage_a_.p, vox, n * 3 * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
  hipLaunchKernelGGL(k_h_query_occ, dim3(grid_for(n)), dim3(256), 0, stream_, g_, (const int32_t *)dir_, page_table(), (const uint32_t *)occbits_.p,
                     (const int32_t *)stage_a_.p, (const double *)nullptr, n, (int32_t *)stage_c_.p);
  FIESTA_HIP_CHECK(hipMemcpyAsync(out, stage_c_.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
}
void HashMap::get_occupancy_pos(const double *pos, int64_t n, int32_t *out) {
  if (n > 0 && n <= kHostQueries) {
    for (int64_t i = 0; i < n; ++i)
      out[i] = host_occ((int)floor((pos[3 * i] - g_.org[0]) / g_.res), (int)floor((pos[3 * i + 1] - g_.org[1]) / g_.res),
                        (int)floor((pos[3 * i + 2] - g_.org[2]) / g_.res));
    return;
  }
  use_device();
  if (n <= 0) return;
  stage_a_.ensure(n * 3 * sizeof(double), stream_);
  stage_c_.ensure(n * sizeof(int32_t), stream_);
  FIESTA_HIP_CHECK(hipMemcpyAsync(stage_a_.p, pos, n * 3 * sizeof(double), hipMemcpyHostToDevice, stream_));
  hipLaunchKernelGGL(k_h_query_occ, dim3(grid_for(n)), dim3(256), 0, stream_, g_, (const int32_t *)dir_, page_table(), (const uint32_t *)occbits_.p,
                     (const int32_t *)nullptr, (const double *)stage_a_.p, n, (int32_t *)stage_c_.p);
  FIESTA_HIP_CHECK(hipMemcpyAsync(out, stage_c_.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
}

int64_t HashMap::download(int32_t *vox, int32_t *d2, int32_t *coc, uint8_t *occ) {
  use_device();
  const int64_t n = npages_ * kPageVox;
  if (n == 0 || (!vox && !d2 && !coc && !occ)) return n;
  int32_t *dv = nullptr, *dd = nullptr, *dc = nullptr;
  uint8_t *doc = nullptr;
  if (vox) stage_a_.ensure(n * 3 * sizeof(int32_t), stream_), dv = (int32_t *)stage_a_.p;
  if (coc) stage_b_.ensure(n * 3 * sizeof(int32_t), stream_), dc = (int32_t *)stage_b_.p;
  if (d2) stage_c_.ensure(n * sizeof(int32_t), stream_), dd = (int32_t *)stage_c_.p;
  if (occ) stage_d_.ensure(std::max<size_t>((size_t)n, (size_t)kNTiles * 4), stream_), doc = (uint8_t *)stage_d_.p;
  hipLaunchKernelGGL(k_h_export, dim3(grid_for(n, 256, 8192)), dim3(256), 0, stream_, (const int32_t *)page_gtile_.p, n,
                     (const vox_t *)coc_.p, (const uint32_t *)occbits_.p, dv, dd, dc, doc);
  FIESTA_HIP_CHECK(hipGetLastError());
  if (vox) FIESTA_HIP_CHECK(hipMemcpyAsync(vox, dv, n * 3 * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
  if (coc) FIESTA_HIP_CHECK(hipMemcpyAsync(coc, dc, n * 3 * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
  if (d2) FIESTA_HIP_CHECK(hipMemcpyAsync(d2, dd, n * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
  if (occ) FIESTA_HIP_CHECK(hipMemcpyAsync(occ, doc, n, hipMemcpyDeviceToHost, stream_));
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
  return n;
}

int64_t HashMap::point_cloud(int vis_lower_bound, int vis_upper_bound, float *xyz, int64_t cap) {
  use_device();
  zero_counter(C_SCRATCH);
  float *dout = nullptr;
  if (xyz && cap > 0) {
    stage_a_.ensure((size_t)cap * 3 * sizeof(float), stream_);
    dout = (float *)stage_a_.p;
  }
  auto c = [](int64_t v) { return (int)std::min<int64_t>(std::max<int64_t>(v, -(1ll << 30)), 1ll << 30); };
  const int64_t nrows = npages_ * kPageRows;
  if (nrows)
    hipLaunchKernelGGL(k_h_point_cloud, dim3(grid_for(nrows, 256, 8192)), dim3(256), 0, stream_, g_, (const int32_t *)page_gtile_.p, nrows,
                       (const uint32_t *)occbits_.p, c(ur_[0]), c(ur_[3]), c(ur_[1]), c(ur_[4]), vis_lower_bound, vis_upper_bound, dout,
                       (unsigned long long)(dout ? cap : 0), &counters_[C_SCRATCH]);
  FIESTA_HIP_CHECK(hipGetLastError());
  const int64_t n = (int64_t)read_counter(C_SCRATCH);
  if (dout && n) FIESTA_HIP_CHECK(hipMemcpyAsync(xyz, dout, (size_t)std::min(n, cap) * 3 * sizeof(float), hipMemcpyDeviceToHost, stream_));
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
  return n;
}

int64_t HashMap::slice_marker(int slice, double max_dist, double *xyz, float *rgba, int64_t cap) {
  use_device();
  zero_counter(C_SCRATCH);
  double *dx = nullptr;
  float *dc = nullptr;
  if (xyz && rgba && cap > 0) {
    stage_a_.ensure((size_t)cap * 3 * sizeof(double), stream_);
    stage_b_.ensure((size_t)cap * 4 * sizeof(float), stream_);
    dx = (double *)stage_a_.p, dc = (float *)stage_b_.p;
  }
  auto c = [](int64_t v) { return (int)std::min<int64_t>(std::max<int64_t>(v, -(1ll << 30)), 1ll << 30); };
  if (npages_)
    hipLaunchKernelGGL(k_h_slice_marker, dim3(grid_for(npages_ * kPageRows, 256, 8192)), dim3(256), 0, stream_, g_,
                       (const int32_t *)page_gtile_.p, npages_, (const vox_t *)coc_.p, c(ur_[0]), c(ur_[3]), c(ur_[1]), c(ur_[4]), slice,
                       max_dist, dx, dc, (unsigned long long)(dx ? cap : 0), &counters_[C_SCRATCH]);
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

// pending (num_hit_, num_miss_) of every voxel of every allocated page, in the order of download()
void HashMap::download_counts(int32_t *num_hit, int32_t *num_miss) {
  use_device();
  const int64_t n = npages_ * kPageVox;
  if (n == 0) return;
  std::vector<unsigned long long> h((size_t)n);
  FIESTA_HIP_CHECK(hipMemcpyAsync(h.data(), cnt_.p, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream_));
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
  for (int64_t i = 0; i < n; ++i) {
    if (num_hit) num_hit[i] = (int32_t)(h[i] >> 32);
    if (num_miss) num_miss[i] = (int32_t)(uint32_t)h[i];
  }
}

void HashMap::synchronize() {
  use_device();
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
}

}  // namespace fiesta
