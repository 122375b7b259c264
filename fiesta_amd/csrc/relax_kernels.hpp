// fiesta_amd/csrc/relax_kernels.hpp -- device code shared by the dense-array map (dense_map.hip) and the paged
// hash-block map (hash_map.hip): tile bookkeeping and the work-queue relaxation kernel k_relax_q.
#pragma once
#include <type_traits>

#include "common.hpp"
#include "dense_map.hpp"

namespace fiesta {

__device__ inline bool occ_test(const uint32_t *occbits, const Geom &g, int x, int y, int z) {
  return (occbits[g.bitword(x, y, z)] >> (z & 31)) & 1u;
}

// a few 64-bit counters to zero: ONE small kernel (a memset of an unaligned range goes out as up to three)
namespace {  // (this header is included by two translation units)
__global__ void k_zero_words(unsigned long long *p, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0;
}
__global__ void k_zero_words2(unsigned long long *p, int n, unsigned long long *q, int m) {  // two ranges, one launch
  for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0;
  for (int i = threadIdx.x; i < m; i += blockDim.x) q[i] = 0;
}
}  // namespace

struct TileGrid {
  int tx, ty;  // tile extent in x and y (z extent is 32)
  int ntx, nty, ntz;
  __device__ inline int tile_of(int x, int y, int z) const { return ((x / tx) * nty + (y / ty)) * ntz + (z >> 5); }
};

__device__ inline void activate_tile(uint32_t t, uint32_t *flag, uint32_t *list, unsigned long long *count) {
  if (atomicExch(&flag[t], 1u) == 0u) list[atomicAdd(count, 1ull)] = t;
}

// =====================================================================================================
// k_relax_q -- work-efficient tile relaxation: an LDS work queue instead of Jacobi sweeps.
//
// The reference pops one voxel at a time and runs 24 pulls + 24 pushes for it (src/ESDFMap.cpp:339-392),
// ~1.04 expansions per updated voxel.  k_relax (v1, dense_map.hip) keeps lanes on fixed voxels and re-reads all 24
// neighbours of EVERY voxel in EVERY sweep, which costs ~20 sweeps x 24 LDS reads per voxel per visit.
// Here a tile + 2-voxel halo is staged in LDS as 64-bit keys  (d^2 << 32 | closest obstacle | flag)  and
// only voxels whose key CHANGED do work, exactly like the reference's queue:
//   * push   a changed voxel v offers its obstacle c to its 24 neighbours n.  |n-c|^2 is not recomputed:
//            |v+e-c|^2 = d(v) + 2 e.(v-c) + |e|^2  (two integer adds per direction); a plain 32-bit read
//            of d(n) filters, ds_min_u64 on the key decides, the winner's bit is set in the next level's
//            frontier bitmap (level-synchronous inside the tile: bitmap -> compact queue -> items, two
//            barriers per level).
//   * pull   a voxel that goes from "no obstacle" to a finite distance (first reached by a wave, or
//            orphaned by a delete: the re-seed of :308-321) asks the neighbours whose value predates
//            this UpdateESDF -- they are a fixed point among themselves and would never push.  Neighbours
//            that changed during this update push by themselves, so they are skipped (flag bit 30 in LDS
//            = "joined the frontier during this update" = the rbits bitmap in HBM).
// Halo voxels are sources only (their d^2 field is 0, so no push can ever win against them); a halo voxel
// offers its obstacle iff it changed in the PREVIOUS round (cbits bitmap, double-buffered by round
// parity and validated by a per-tile round stamp) or still carries a seed tag in HBM.
// =====================================================================================================
struct RelaxQArgs {
  Geom g;
  TileGrid tg;
  vox_t *coc;
  uint32_t *rbits;
  uint32_t *tile_epoch;
  uint32_t epoch;
  uint32_t *cbits_prev;   // written by the previous round (read here for halo voxels)
  uint32_t *cbits_cur;    // written by this round
  const uint32_t *cstamp_prev;
  uint32_t *cstamp_cur;
  uint32_t serial;        // serial number of this round; stamps equal to serial-1 validate cbits_prev
  const uint32_t *list_cur;
  uint32_t n_cur;
  const unsigned long long *n_cur_dev;  // list mode: if set, the length of list_cur is read from here (a round launched
                                        // before the host knows how many tiles its predecessor activated)
  uint32_t *flag_cur;
  uint32_t *flag_next;
  uint32_t *list_next;
  unsigned long long *count_next;
  unsigned long long *count_zero;  // the list counter nobody uses during this round: cleared here for the round after next
  unsigned long long *counters;
  int spatial;  // dense maps: 1 = walk all tiles in XCD-chunked spatial order (flag_cur is the list)
  int prof;  // 1: accumulate per-phase cycle counters into counters[C_PROF0..]
  // paged (hash-block) maps: voxel data lives in a pool of pages, one page = one tile (TX x TY x 32 voxels, same
  // z-fastest row layout), found through the dense page directory dir[tile] (-1: not allocated = all unobserved)
  const int32_t *dir;
};

// TRACK: also maintain counters[C_MAXD2], an upper bound of every finite d^2 stored (bounds the delete scan of maps that
// take many small updates, see k_invalidate; a separate instance so that the default code stays as it is).
template <int TX, int TY, int NT, bool PAGED = false, bool TRACK = false>
__global__ __launch_bounds__(NT, ((NT >= 1024 || NT * 4 >= TX * TY * 32) ? 4 : 2)) void k_relax_q(RelaxQArgs a) {
  // 4 waves per SIMD (128 VGPRs): one 1024-thread work-group per CU, or two 512-thread work-groups on 8x8x32 tiles
  constexpr bool LOWREG = NT >= 1024 || NT * 4 >= TX * TY * 32;
  constexpr int TZ = 32, H = 2;
  constexpr int RX = TX + 2 * H, RY = TY + 2 * H, RZ = TZ + 2 * H;
  constexpr int RSIZE = RX * RY * RZ;
  constexpr int NW = (RSIZE + 63) / 64 * 2;  // bitmap words (32 voxels each), padded to whole waves
  constexpr int SLOTS = NT / 32, RPT = TX * TY / SLOTS;
  constexpr int NROWW = RX * RY * 3;  // staged bitmap words: 3 z-words per (x,y) row of the region
  static_assert(NT % 64 == 0 && (TX * TY) % SLOTS == 0 && RPT <= 32 && RSIZE < 65536 && NW <= NT, "tile shape");
  // keys: lo half = obstacle word | flag, hi half = d^2. The halves are read through KW(j) = the array itself
  // reinterpreted in place (no pointer variable, no volatile): that keeps every access a DS instruction -- a
  // `volatile uint32_t *` alias of K compiled to FLAT loads.
  __shared__ unsigned long long K[RSIZE];
#define KW(j) (reinterpret_cast<uint32_t *>(K)[j])
#define K64(i) (K[i])
  __shared__ uint16_t Q[RSIZE];
  __shared__ uint32_t F[2][NW], E[NW], P[NW];
  __shared__ uint32_t FH[NW];  // level-0 sources that are not voxels of the tile: halo voxels, ghost cells of a shard
  __shared__ uint32_t rb[NROWW], cb[NROWW];
  __shared__ uint32_t qcount[2];
  __shared__ uint32_t n_oldvalid;
  __shared__ uint32_t nb_ok[27];
  __shared__ int32_t nb_page[27];
  __shared__ int nbr_dirty[27];
  constexpr int PAGE_VOX = TX * TY * TZ, PAGE_ROWS = TX * TY;

  const Geom &g = a.g;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;

  // Two ways to walk the active tiles. PAGED maps (huge, mostly empty directory): the compact list of the round.
  // Dense maps: every tile in SPATIAL order, skipping the inactive ones (flag_cur is the list) -- tiles are grouped
  // in chunks of 64 consecutive ids (a 16-tile z-column x 4 y-neighbours), chunk c goes to XCD c % 8 and block b runs
  // on XCD b % 8 (observed placement, used for speed only), so tiles that share halo lines meet in one XCD's L2
  // at about the same time instead of each missing to HBM.
  const uint32_t ntiles = (uint32_t)(a.tg.ntx * a.tg.nty * a.tg.ntz);
  const uint32_t n_list = a.n_cur_dev ? (uint32_t)*a.n_cur_dev : a.n_cur;
  if (a.n_cur_dev && n_list && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&a.counters[C_ROUNDS], 1ull);
  if (a.count_zero && blockIdx.x == 0 && threadIdx.x == 0) *a.count_zero = 0;
  // statistics are summed per work-group and flushed once after the walk (tens of thousands of visits would
  // otherwise queue their atomics on three hot addresses)
  uint32_t acc_writes = 0, acc_levels = 0, acc_visits = 0, acc_maxd2 = 0;
  for (uint32_t it = 0;; ++it) {
    uint32_t t;
    if (PAGED || a.spatial == 0) {
      const uint32_t li = blockIdx.x + it * gridDim.x;
      if (li >= n_list) break;
      t = a.list_cur[li];
    } else {
      const uint32_t xcd = blockIdx.x & 7u, v = (blockIdx.x >> 3) + it * (gridDim.x >> 3);
      const uint32_t chunk = (v >> 6) * 8u + xcd;
      if (chunk * 64u >= ntiles) break;
      t = chunk * 64u + (v & 63u);
      const bool active = t < ntiles && a.flag_cur[t] != 0u;
      __syncthreads();  // everybody has read the flag before thread 0 clears it below
      if (!active) continue;
    }
    const int tz = t % a.tg.ntz, ty = (t / a.tg.ntz) % a.tg.nty, tx = t / (a.tg.ntz * a.tg.nty);
    const int x0 = tx * TX, y0 = ty * TY, z0 = tz * TZ;
    const int bx = g.gx0 + x0 - H, by = g.gy0 + y0 - H, bz = g.gz0 + z0 - H;  // global coords of r-index 0
    const bool prof = a.prof != 0;
    // the thread id, opaque per visit: index arithmetic derived from it is recomputed per tile (a few VALU
    // instructions) instead of being hoisted out of the tile loop as dozens of invariants that spill to scratch
    int tl = tid;
    asm volatile("" : "+v"(tl));
    uint32_t n_items = 0;
    long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, ts_a = 0, ts_b = 0;  // (prof == 2: staging sub-phases)
    if (prof) t0 = clock64();
    if (tid == 0) {
      a.flag_cur[t] = 0;
      n_oldvalid = 0;
      qcount[0] = 0;
    }
    if (tid < 27) {
      nbr_dirty[tid] = 0;
      const int ox = tid / 9 - 1, oy = (tid / 3) % 3 - 1, oz = tid % 3 - 1;
      const int ux = tx + ox, uy = ty + oy, uz = tz + oz;
      uint32_t ok = 0;
      int32_t pg = -1;
      if ((unsigned)ux < (unsigned)a.tg.ntx && (unsigned)uy < (unsigned)a.tg.nty && (unsigned)uz < (unsigned)a.tg.ntz) {
        const uint32_t ot = (ux * a.tg.nty + uy) * a.tg.ntz + uz;
        if (PAGED) pg = a.dir[ot];
        if (!PAGED || pg >= 0) {
          if (a.tile_epoch[ot] == a.epoch) ok |= 1u;
          if (tid != 13 && a.cstamp_prev[ot] == a.serial - 1u) ok |= 2u;
        }
      }
      nb_ok[tid] = ok;
      nb_page[tid] = pg;
    }
    __syncthreads();  // nb_ok
    if (prof) ts_a = clock64();
    if (PAGED && nb_page[13] < 0) {  // a tile without a page holds no observed voxel: nothing to relax, nothing to write
      __syncthreads();
      continue;
    }
    // ---- stage the region: raw voxel word -> 64-bit key, and collect the level-0 frontier.
    // EVERY voxel load of the region is issued up front, before the bitmaps are staged: a lane owns 4 consecutive
    // tile-interior z of one (x,y) row (one 16-byte load, 8 lanes per row; the row arithmetic is shared by the four
    // voxels), the 2+2 z-halo voxels of a row are two 8-byte loads. One memory latency per visit instead of three.
    // Addresses are always legal, so no load sits behind a branch. (nz % 4 != 0: the same code with 4-byte loads.)
    // the whole region (tile + halo) lies inside the grid and the update window: no per-voxel range tests
    const bool region_inside = x0 - H >= max(0, g.wx0) && x0 + TX + H - 1 <= min(g.nx - 1, g.wx1) &&
                               y0 - H >= max(0, g.wy0) && y0 + TY + H - 1 <= min(g.ny - 1, g.wy1) &&
                               z0 - H >= max(0, g.wz0) && z0 + TZ + H - 1 <= min(g.nz - 1, g.wz1);
    const bool vec = PAGED || (g.nz & 3) == 0;
    constexpr int NROWS = RX * RY;
    constexpr int NQ = NROWS * (TZ / 4), UM = (NQ + NT - 1) / NT;  // interior quads
    constexpr int NHP = NROWS * 2, UH = (NHP + NT - 1) / NT;       // z-halo pairs
    auto loadn = [&](auto n_tag, const int rx, const int ry, const int rz, vox_t *out) {
      constexpr int N = decltype(n_tag)::value;
      const int x = x0 - H + rx, y = y0 - H + ry, z = z0 - H + rz;
      bool okxy = region_inside || ((unsigned)x < (unsigned)g.nx && (unsigned)y < (unsigned)g.ny && x >= g.wx0 &&
                                    x <= g.wx1 && y >= g.wy0 && y <= g.wy1);
      int64_t idx;
      if (PAGED) {
        const int ox = (rx < H) ? 0 : ((rx >= TX + H) ? 2 : 1), oy = (ry < H) ? 0 : ((ry >= TY + H) ? 2 : 1),
                  oz = (rz < H) ? 0 : ((rz >= TZ + H) ? 2 : 1);
        const int32_t pg = nb_page[ox * 9 + oy * 3 + oz];
        okxy = okxy && pg >= 0;
        idx = (int64_t)max(pg, 0) * PAGE_VOX + (((x & (TX - 1)) * TY + (y & (TY - 1))) * TZ + (z & (TZ - 1)));
      } else {
        idx = g.idx(x, y, z);
      }
      bool okz[N];
#pragma unroll
      for (int k = 0; k < N; ++k) okz[k] = region_inside || ((unsigned)(z + k) < (unsigned)g.nz && z + k >= g.wz0 && z + k <= g.wz1);
      if (vec) {  // z and nz are multiples of N: the run is inside the array or outside as a whole
        const bool in = okxy && (region_inside || (unsigned)z < (unsigned)g.nz);
        vox_t q[N];
        if constexpr (N == 4) {
          const uint4 t = *reinterpret_cast<const uint4 *>(a.coc + (in ? idx : 0));
          q[0] = t.x, q[1] = t.y, q[2] = t.z, q[3] = t.w;
        } else {
          const uint2 t = *reinterpret_cast<const uint2 *>(a.coc + (in ? idx : 0));
          q[0] = t.x, q[1] = t.y;
        }
#pragma unroll
        for (int k = 0; k < N; ++k) out[k] = (in && okz[k]) ? q[k] : kUnobserved;
      } else {
#pragma unroll
        for (int k = 0; k < N; ++k) {
          const bool ok = okxy && okz[k];
          const vox_t w = a.coc[ok ? idx + k : 0];
          out[k] = ok ? w : kUnobserved;
        }
      }
    };
    vox_t wq[UM][4], wh[UH][2];
#pragma unroll
    for (int u = 0; u < UM; ++u) {
      const int id = min(tl + u * NT, NQ - 1), row = id >> 3, quad = id & 7;
      loadn(std::integral_constant<int, 4>{}, row / RY, row % RY, H + 4 * quad, wq[u]);
    }
#pragma unroll
    for (int u = 0; u < UH; ++u) {
      const int id = min(tl + u * NT, NHP - 1), row = id >> 1;
      loadn(std::integral_constant<int, 2>{}, row / RY, row % RY, (id & 1) ? TZ + H : 0, wh[u]);
    }
    // ---- stage the frontier bitmaps of the region's rows (3 z-words per row) through LDS
    for (int j = tl; j < NROWW; j += NT) {
      const int k = j % 3, ry = (j / 3) % RY, rx = j / (3 * RY);
      const int x = x0 - H + rx, y = y0 - H + ry, zt = tz - 1 + k;
      const int ox = (rx < H) ? 0 : ((rx >= TX + H) ? 2 : 1), oy = (ry < H) ? 0 : ((ry >= TY + H) ? 2 : 1);
      const uint32_t ok = nb_ok[ox * 9 + oy * 3 + k];
      uint32_t r = 0, c = 0;
      if (ok && (unsigned)x < (unsigned)g.nx && (unsigned)y < (unsigned)g.ny) {
        const int64_t wi = PAGED ? (int64_t)nb_page[ox * 9 + oy * 3 + k] * PAGE_ROWS + ((x % TX) * TY + (y % TY))
                                 : ((int64_t)x * g.ny + y) * g.nzw + zt;
        if (ok & 1u) r = a.rbits[wi];
        if (ok & 2u) c = a.cbits_prev[wi];
      }
      rb[j] = r;
      cb[j] = c;
    }
    for (int j = tid; j < NW; j += NT) {
      E[j] = 0;
      P[j] = 0;
      F[0][j] = 0;
      F[1][j] = 0;
      FH[j] = 0;
    }
    __syncthreads();  // rb, cb staged; frontier bitmaps cleared
    if (prof) ts_b = clock64();
    const bool own_epoch = nb_ok[13] & 1u;

    uint32_t oldvalid = 0;
    // flags: 1 = already in the frontier (E), 2 = has no obstacle yet (P), 4 = level-0 frontier (F0), 8 = source only (FH)
    auto build = [&](auto zin_tag, const vox_t w, const int rx, const int ry, const int rz, uint32_t &flags) -> unsigned long long {
      constexpr bool ZIN = decltype(zin_tag)::value;  // rz is a tile-interior z
      flags = 0;
      uint32_t lo = kUnobserved, hi = 0;
      if (w != kUnobserved) {
        const bool act = (w & kAct) != 0;
        const bool valid = !(w & kNoCoc);
        const bool interior = ZIN && (unsigned)(rx - H) < (unsigned)TX && (unsigned)(ry - H) < (unsigned)TY;
        const int zq = rz + (TZ - H);  // z - (z0 - TZ)
        const int j = (rx * RY + ry) * 3 + (zq >> 5);
        const uint32_t bit = 1u << (zq & 31);
        const bool upd = interior && (!g.sharded || g.owned(x0 - H + rx, y0 - H + ry, z0 - H + rz));
        const bool src = !upd && valid && (act || (cb[j] & bit));
        // "joined the frontier during this update": own voxels from the tile's own bitmap; a halo voxel
        // only if it offers its obstacle in this very visit (a neighbour tile that runs concurrently may
        // already have published a newer bitmap than the value loaded above)
        const bool inR = upd ? (act || (rb[j] & bit)) : src;
        const vox_t c = w & ~kAct;
        lo = (valid ? c : kInf) | (inR ? kAct : 0u);
        if (upd) {
          hi = valid ? (uint32_t)dist2(g.wrap, bx + rx, by + ry, bz + rz, c) : (uint32_t)(g.wrap ? kD2Cap : kD2Inf);
          if (act) flags |= 1u | 4u;
          // a voxel without an obstacle asks its old-valid neighbours once, when it first joins the frontier:
          // now if it was orphaned by a delete (the re-seed of :308-321), else when a wave first reaches it
          if (!valid) flags |= 2u;
        } else if (src) {  // a source only: its d^2 field stays 0
          // a halo voxel can reach the tile through the stencil only from a face slab (one axis outside) or
          // from an edge at distance 1 on both outside axes (the +-1,+-1 diagonals); ghost cells sit inside
          const int ox_ = (rx < H) ? H - rx : ((rx >= TX + H) ? rx - (TX + H) + 1 : 0);
          const int oy_ = (ry < H) ? H - ry : ((ry >= TY + H) ? ry - (TY + H) + 1 : 0);
          const int oz_ = ZIN ? 0 : ((rz < H) ? H - rz : rz - (TZ + H) + 1);
          const int nout = (ox_ != 0) + (oy_ != 0) + (oz_ != 0);
          if (nout <= 1 || (nout == 2 && ox_ <= 1 && oy_ <= 1 && oz_ <= 1)) flags |= 8u;
        }
        if (valid && !inR) ++oldvalid;
      }
      return ((unsigned long long)hi << 32) | lo;
    };
#pragma unroll
    for (int u = 0; u < UM; ++u) {
      const int id = tl + u * NT;
      if (id < NQ) {
        const int row = id >> 3, quad = id & 7, rx = row / RY, ry = row % RY, rz0 = H + 4 * quad;
        const int i0 = row * RZ + rz0, w0 = i0 >> 5, sh = i0 & 31;
        uint32_t ne = 0, np = 0, nf = 0, nh = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          uint32_t flags;
          K64(i0 + k) = build(std::true_type{}, wq[u][k], rx, ry, rz0 + k, flags);
          ne |= (flags & 1u) << k;
          np |= ((flags >> 1) & 1u) << k;
          nf |= ((flags >> 2) & 1u) << k;
          nh |= ((flags >> 3) & 1u) << k;
        }
        auto put = [&](uint32_t *B, const uint32_t nib) {  // 4 frontier bits of this quad (a quad may straddle two words)
          if (nib) {
            __hip_atomic_fetch_or(&B[w0], nib << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t m2 = sh > 28 ? (nib >> (32 - sh)) : 0u;
            if (m2) __hip_atomic_fetch_or(&B[w0 + 1], m2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        };
        put(E, ne);
        put(P, np);
        put(F[0], nf);
        put(FH, nh);
      }
    }
#pragma unroll
    for (int u = 0; u < UH; ++u) {
      const int id = tl + u * NT;
      if (id < NHP) {
        const int row = id >> 1, rz0 = (id & 1) ? TZ + H : 0, i0 = row * RZ + rz0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          uint32_t flags;
          K64(i0 + k) = build(std::false_type{}, wh[u][k], row / RY, row % RY, rz0 + k, flags);
          if (flags & 8u)
            __hip_atomic_fetch_or(&FH[(i0 + k) >> 5], 1u << ((i0 + k) & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
    for (int off = 32; off > 0; off >>= 1) oldvalid += __shfl_down(oldvalid, off);
    if (lane == 0 && oldvalid) atomicAdd(&n_oldvalid, oldvalid);
    if (prof) t1 = clock64();

    // ---- level-synchronous propagation inside the tile
    long long tc = 0, tp = 0, tmark = 0;
    uint32_t n_pulls = 0, n_succ = 0;
    __syncthreads();  // keys, frontier bitmaps and n_oldvalid are complete
    const bool pulls_enabled = n_oldvalid != 0;
    // compact a frontier bitmap into the work queue Q; returns the number of items. Wave scan + one LDS atomicAdd per
    // wave for the wave's base (the order of Q does not matter: ds_min on whole keys is commutative), one barrier.
    // qcount[slot] is zero on entry; the other slot is cleared for the next call.
    auto compact = [&](uint32_t *Fsrc, const bool into_E, const int slot) -> uint32_t {
      if (wave * 64 < NW) {
        uint32_t bits = 0;
        if (tid < NW) {
          bits = Fsrc[tid];
          Fsrc[tid] = 0;
          if (bits && into_E) E[tid] |= bits;
        }
        uint32_t incl = __popc(bits);
        for (int off = 1; off < 64; off <<= 1) {
          const uint32_t o = __shfl_up(incl, off);
          if (lane >= off) incl += o;
        }
        uint32_t base = 0;
        if (lane == 63 && incl) base = atomicAdd(&qcount[slot], incl);
        base = __shfl(base, 63) + incl - __popc(bits);
        while (bits) {
          const int bpos = __ffs(bits) - 1;
          bits &= bits - 1;
          Q[base++] = (uint16_t)(tid * 32 + bpos);
        }
      }
      if (tid == NT - 1) qcount[slot ^ 1] = 0;
      __syncthreads();
      return qcount[slot];
    };
    // One frontier item = one voxel whose key changed. SRC: the item is a source-only voxel (halo voxel or ghost cell
    // of a shard; they only occur in the pass before level 0): no pull, its d^2 is recomputed, and a halo voxel only
    // probes the <= 6 directions that land inside the tile instead of 24.
    auto process = [&](auto src_tag, const uint32_t total, uint32_t *Fn) {
      constexpr bool SRC = decltype(src_tag)::value;
      for (uint32_t j = tid; j < total; j += NT) {
        const int v = Q[j];
        // all 24 neighbours as NON-NEGATIVE constant offsets from one base: they fold into the DS instructions'
        // unsigned immediate offset (a negative offset costs an address register and two VALU instructions each)
        constexpr int NB0 = 2 * RY * RZ;
        int vb = v - NB0;
        asm volatile("" : "+v"(vb));  // keep the compiler from folding vb + NB0 back into v
        __builtin_assume(vb >= 0 && vb < RSIZE);
        const uint32_t vbit = 1u << (v & 31);
        const int rz = v % RZ, ry = (v / RZ) % RY, rx = v / (RZ * RY);
        const int vx = bx + rx, vy = by + ry, vz = bz + rz;
        unsigned long long key = __hip_atomic_load(&K64(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
        if (a.prof == 1) ++n_items;
        if (!SRC && (P[v >> 5] & vbit)) {  // had no obstacle when the tile was staged, not asked yet
          __hip_atomic_fetch_and(&P[v >> 5], ~vbit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (pulls_enabled) {
            if (a.prof == 1) ++n_pulls;
            vox_t best = lo;
            uint32_t bestd = hi;
#define FIESTA_PULLL(DX, DY, DZ) un[q++] = KW(2 * (vb + (NB0 + ((DX)*RY + (DY)) * RZ + (DZ))));
#define FIESTA_PULLEVAL(NQ)                                 \
  _Pragma("unroll") for (int q = 0; q < (NQ); ++q) {        \
    const vox_t u = un[q];                                  \
    if (!(u & (kNoCoc | kAct))) {                           \
      const uint32_t d = (uint32_t)dist2(g.wrap, vx, vy, vz, u); \
      if (d < bestd) {                                      \
        bestd = d;                                          \
        best = u;                                           \
      }                                                     \
    }                                                       \
  }
            if (LOWREG) {
              vox_t un[12];
              {
                int q = 0;
                FIESTA_STENCIL12A(FIESTA_PULLL)
              }
              FIESTA_PULLEVAL(12)
              {
                int q = 0;
                FIESTA_STENCIL12B(FIESTA_PULLL)
              }
              FIESTA_PULLEVAL(12)
            } else {
              vox_t un[24];
              {
                int q = 0;
                FIESTA_STENCIL24(FIESTA_PULLL)
              }
              FIESTA_PULLEVAL(24)
            }
#undef FIESTA_PULLL
#undef FIESTA_PULLEVAL
            if (bestd < hi) {
              const unsigned long long mine = ((unsigned long long)bestd << 32) | best | kAct;
              const unsigned long long old = atomicMin(&K64(v), mine);
              key = old < mine ? old : mine;
              lo = (uint32_t)key;
              hi = (uint32_t)(key >> 32);
            }
          }
        }
        if (lo & kNoCoc) continue;
        // push: |v+e-c|^2 = d(v) + 2 e.(v-c) + |e|^2
        const vox_t c = lo & ~kAct;
        int rcx, rcy, rcz;
        coc_offset(g.wrap, vx, vy, vz, c, rcx, rcy, rcz);
        const int32_t dv = SRC ? rcx * rcx + rcy * rcy + rcz * rcz : (int32_t)hi;  // (source-only voxels keep d^2 = 0 in LDS)
        const unsigned long long keylo = (unsigned long long)(c | kAct);
        if (SRC) {
          const int ux = (rx < H) ? 1 : ((rx >= TX + H) ? -1 : 0), uy = (ry < H) ? 1 : ((ry >= TY + H) ? -1 : 0),
                    uz = (rz < H) ? 1 : ((rz >= TZ + H) ? -1 : 0);  // direction towards the tile per axis (0: inside)
          const int nout = (ux != 0) + (uy != 0) + (uz != 0);
          if (nout) {  // a halo voxel
            const int depth = max(max((rx < H) ? H - rx : rx - (TX + H) + 1, (ry < H) ? H - ry : ry - (TY + H) + 1),
                                  (rz < H) ? H - rz : rz - (TZ + H) + 1);
            auto probe = [&](const int ex, const int ey, const int ez) {
              const int n = v + (ex * RY + ey) * RZ + ez;
              const uint32_t cand = (uint32_t)(dv + 2 * (ex * rcx + ey * rcy + ez * rcz) + ex * ex + ey * ey + ez * ez);
              if (cand < KW(2 * n + 1)) {
                if (a.prof == 1) ++n_succ;
                __hip_atomic_fetch_min(&K64(n), ((unsigned long long)cand << 32) | keylo, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_or(&Fn[n >> 5], 1u << (n & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              }
            };
            if (nout == 1) {  // face slab: the step towards the tile, its four diagonals, and the double step
              if (depth == 1) {
                const int bx_ = (ux == 0) ? 1 : 0, by_ = (ux != 0) ? 1 : 0;  // first in-range axis
                const int cz_ = (uz == 0) ? 1 : 0, cy_ = (uz != 0) ? 1 : 0;  // second in-range axis
                probe(ux, uy, uz);
                probe(ux + bx_, uy + by_, uz);
                probe(ux - bx_, uy - by_, uz);
                probe(ux, uy + cy_, uz + cz_);
                probe(ux, uy - cy_, uz - cz_);
              }
              probe(2 * ux, 2 * uy, 2 * uz);
            } else if (nout == 2 && depth == 1) {
              probe(ux, uy, uz);  // edge: the one diagonal that lands inside
            }
            continue;
          }
        }
        const int ax = 2 * rcx, ay = 2 * rcy, az = 2 * rcz;
#define FIESTA_NIDX(DX, DY, DZ) (vb + (NB0 + ((DX)*RY + (DY)) * RZ + (DZ)))
#define FIESTA_PUSHL(DX, DY, DZ) dnv[q++] = KW(2 * FIESTA_NIDX(DX, DY, DZ) + 1);
#define FIESTA_PUSH(DX, DY, DZ)                                                                              \
  {                                                                                                          \
    const uint32_t cand = (uint32_t)(dv + (DX)*ax + (DY)*ay + (DZ)*az + ((DX) * (DX) + (DY) * (DY) + (DZ) * (DZ))); \
    if (cand < dnv[q++]) {                                                                                   \
      if (a.prof == 1) ++n_succ;                                                                             \
      const int n = FIESTA_NIDX(DX, DY, DZ);                                                                 \
      __hip_atomic_fetch_min(&K64(n), ((unsigned long long)cand << 32) | keylo, __ATOMIC_RELAXED,              \
                             __HIP_MEMORY_SCOPE_WORKGROUP);                                                  \
      __hip_atomic_fetch_or(&Fn[n >> 5], 1u << (n & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);    \
    }                                                                                                        \
  }
        if (LOWREG) {  // 128-VGPR budget: four batches of 6 filter reads
          uint32_t dnv[6];
#define FIESTA_BATCH(ST)  \
  {                       \
    int q = 0;            \
    ST(FIESTA_PUSHL)      \
  }                       \
  {                       \
    int q = 0;            \
    ST(FIESTA_PUSH)       \
  }
          FIESTA_BATCH(FIESTA_STENCIL6A)
          FIESTA_BATCH(FIESTA_STENCIL6B)
          FIESTA_BATCH(FIESTA_STENCIL6C)
          FIESTA_BATCH(FIESTA_STENCIL6D)
#undef FIESTA_BATCH
        } else {
          uint32_t dnv[24];
          {
            int q = 0;
            FIESTA_STENCIL24(FIESTA_PUSHL)
          }
          {
            int q = 0;
            FIESTA_STENCIL24(FIESTA_PUSH)
          }
        }
#undef FIESTA_PUSHL
#undef FIESTA_PUSH
#undef FIESTA_NIDX
      }
    };

    // Sparse levels: one frontier item = FOUR lanes, six directions each (the stencil's four groups of 6, in the
    // reference's order).  A sparse frontier -- a depth frame's update is a few hundred items per level -- leaves most
    // of the work-group idle, and what a level costs is one item's dependent chain: a quarter of the neighbour reads
    // and atomics per lane is a quarter of that chain.  The directions are per-lane data (no branch on the lane's
    // quarter, so a wave never diverges over them); the results are those of one lane doing all 24 (the pull keeps the
    // FIRST strictly smaller candidate in stencil order: the lower quarter wins ties; pushes are commutative ds_min).
    auto process_split = [&](const uint32_t total, uint32_t *Fn) {
      const int part = tl & 3;
      int ex[6], ey[6], ez[6], nd[6];
      {
        constexpr int8_t A[6][3] = {FIESTA_STENCIL6A(FIESTA_DIR3)}, B[6][3] = {FIESTA_STENCIL6B(FIESTA_DIR3)},
                         C[6][3] = {FIESTA_STENCIL6C(FIESTA_DIR3)}, D[6][3] = {FIESTA_STENCIL6D(FIESTA_DIR3)};
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          ex[q] = part == 0 ? A[q][0] : (part == 1 ? B[q][0] : (part == 2 ? C[q][0] : D[q][0]));
          ey[q] = part == 0 ? A[q][1] : (part == 1 ? B[q][1] : (part == 2 ? C[q][1] : D[q][1]));
          ez[q] = part == 0 ? A[q][2] : (part == 1 ? B[q][2] : (part == 2 ? C[q][2] : D[q][2]));
          nd[q] = (ex[q] * RY + ey[q]) * RZ + ez[q];
        }
      }
      for (uint32_t j = (uint32_t)tl >> 2; j < total; j += NT / 4) {
        const int v = Q[j];
        const uint32_t vbit = 1u << (v & 31);
        const int rz = v % RZ, ry = (v / RZ) % RY, rx = v / (RZ * RY);
        const int vx = bx + rx, vy = by + ry, vz = bz + rz;
        unsigned long long key = __hip_atomic_load(&K64(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
        if (a.prof == 1 && part == 0) ++n_items;
        if (P[v >> 5] & vbit) {  // had no obstacle when the tile was staged, not asked yet (the same answer in all four lanes)
          if (part == 0) __hip_atomic_fetch_and(&P[v >> 5], ~vbit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (pulls_enabled) {
            if (a.prof == 1 && part == 0) ++n_pulls;
            vox_t un[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) un[q] = KW(2 * (v + nd[q]));
            vox_t best = lo;
            uint32_t bestd = hi;
#pragma unroll
            for (int q = 0; q < 6; ++q) {
              const vox_t u = un[q];
              if (!(u & (kNoCoc | kAct))) {
                const uint32_t d = (uint32_t)dist2(g.wrap, vx, vy, vz, u);
                if (d < bestd) {
                  bestd = d;
                  best = u;
                }
              }
            }
#pragma unroll
            for (int m = 1; m <= 2; m <<= 1) {  // the four quarters' minima; on a tie the lower quarter's
              const uint32_t od = (uint32_t)__shfl_xor((int)bestd, m);
              const vox_t ob = (vox_t)__shfl_xor((int)best, m);
              if ((part & m) ? od <= bestd : od < bestd) {
                bestd = od;
                best = ob;
              }
            }
            if (bestd < hi) {
              const unsigned long long mine = ((unsigned long long)bestd << 32) | best | kAct;
              unsigned long long res = mine;
              if (part == 0) {
                const unsigned long long old = atomicMin(&K64(v), mine);
                res = old < mine ? old : mine;
              }
              lo = (uint32_t)__shfl((int)(uint32_t)res, lane & ~3);
              hi = (uint32_t)__shfl((int)(uint32_t)(res >> 32), lane & ~3);
            }
          }
        }
        if (lo & kNoCoc) continue;
        // push: |v+e-c|^2 = d(v) + 2 e.(v-c) + |e|^2
        const vox_t c = lo & ~kAct;
        int rcx, rcy, rcz;
        coc_offset(g.wrap, vx, vy, vz, c, rcx, rcy, rcz);
        const int32_t dv = (int32_t)hi;
        const unsigned long long keylo = (unsigned long long)(c | kAct);
        const int ax = 2 * rcx, ay = 2 * rcy, az = 2 * rcz;
        uint32_t dnv[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) dnv[q] = KW(2 * (v + nd[q]) + 1);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          const uint32_t cand = (uint32_t)(dv + ex[q] * ax + ey[q] * ay + ez[q] * az + (ex[q] * ex[q] + ey[q] * ey[q] + ez[q] * ez[q]));
          if (cand < dnv[q]) {
            if (a.prof == 1) ++n_succ;
            const int n = v + nd[q];
            __hip_atomic_fetch_min(&K64(n), ((unsigned long long)cand << 32) | keylo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_or(&Fn[n >> 5], 1u << (n & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    };

    // pass before level 0: the source-only voxels offer their obstacles; what they improve joins level 0
    if (prof) tmark = clock64();
    {
      const uint32_t nsrc = compact(FH, true, 0);  // (ghost cells of a shard sit inside the tile: their tag is cleared and
                                                //  their change published by the write-back, like any tile voxel)
      if (prof) {
        const long long now = clock64();
        tc += now - tmark;
        tmark = now;
      }
      if (nsrc) process(std::true_type{}, nsrc, F[0]);
      if (prof) tp += clock64() - tmark;
    }
    uint32_t level = 0;
    for (;; ++level) {
      const int cur = level & 1;
      __syncthreads();  // every push of the previous level has landed in F[cur]
      if (prof) tmark = clock64();
      const uint32_t total = compact(F[cur], true, cur ^ 1);
      if (prof) {
        const long long now = clock64();
        tc += now - tmark;
        tmark = now;
      }
      if (total == 0) break;
      if (total > (uint32_t)NT)  // (a dense level keeps every lane busy with one item each: fewer instructions per item)
        process(std::false_type{}, total, F[cur ^ 1]);
      else
        process_split(total, F[cur ^ 1]);
      if (prof) tp += clock64() - tmark;
    }
    if (prof) t2 = clock64();

    // ---- write back what changed, publish frontier membership, wake the neighbours
    const int lz = tl & 31, slot = tl >> 5;
    uint32_t nwrites = 0;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const int row = slot + SLOTS * r, lx = row / TY, ly = row % TY;
      const int x = x0 + lx, y = y0 + ly, z = z0 + lz;
      const int ri = ((lx + H) * RY + (ly + H)) * RZ + (lz + H);
      const bool e = (E[ri >> 5] >> (ri & 31)) & 1u;
      if (e) {
        const unsigned long long kfin = K64(ri);  // (one 8-byte read: the d^2 half feeds the distance bound below)
        const vox_t w = (vox_t)kfin & ~kAct;
        a.coc[PAGED ? (int64_t)nb_page[13] * PAGE_VOX + (lx * TY + ly) * TZ + lz : g.idx(x, y, z)] = w;
        ++nwrites;
        if (TRACK && !(w & kNoCoc)) acc_maxd2 = max(acc_maxd2, (uint32_t)(kfin >> 32));
        // Which neighbour tiles can this voxel's change reach through the 24-direction stencil (radius 2)?
        // Faces: within 2 of the face (the +-1 and +-2 axis steps). Edges: within 1 of both faces (the +-1,+-1
        // diagonals). Corners: never (the stencil has no 3-axis diagonals, include/parameters.h:54-68).
        if (!(w & kNoCoc)) {
          const int fx = (lx < H) ? -1 : ((lx >= TX - H) ? 1 : 0), fy = (ly < H) ? -1 : ((ly >= TY - H) ? 1 : 0),
                    fz = (lz < H) ? -1 : ((lz >= TZ - H) ? 1 : 0);
          if (fx | fy | fz) {
            const int ex = (lx == 0) ? -1 : ((lx == TX - 1) ? 1 : 0), ey = (ly == 0) ? -1 : ((ly == TY - 1) ? 1 : 0),
                      ez = (lz == 0) ? -1 : ((lz == TZ - 1) ? 1 : 0);
            if (fx) nbr_dirty[(fx + 1) * 9 + 4] = 1;
            if (fy) nbr_dirty[9 + (fy + 1) * 3 + 1] = 1;
            if (fz) nbr_dirty[9 + 3 + (fz + 1)] = 1;
            if (ex && ey) nbr_dirty[(ex + 1) * 9 + (ey + 1) * 3 + 1] = 1;
            if (ex && ez) nbr_dirty[(ex + 1) * 9 + 3 + (ez + 1)] = 1;
            if (ey && ez) nbr_dirty[9 + (ey + 1) * 3 + (ez + 1)] = 1;
          }
        }
      }
      const unsigned long long b = __ballot(e);
      if (lz == 0 && x < g.nx && y < g.ny && z < g.nz) {
        const uint32_t ebits = (tid & 32) ? (uint32_t)(b >> 32) : (uint32_t)b;
        const int64_t wi = PAGED ? (int64_t)nb_page[13] * PAGE_ROWS + lx * TY + ly : g.bitword(x, y, z);
        a.rbits[wi] = own_epoch ? (rb[((lx + H) * RY + (ly + H)) * 3 + 1] | ebits) : ebits;
        a.cbits_cur[wi] = ebits;
      }
    }
    acc_writes += nwrites;
    __syncthreads();
    if (prof) {
      t3 = clock64();
      uint32_t v = n_items;
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
      if (lane == 0 && v) atomicAdd(&a.counters[C_PROF0 + 3], (unsigned long long)v);
      v = n_pulls;
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
      if (lane == 0 && v) atomicAdd(&a.counters[C_PROF0 + 6], (unsigned long long)v);
      v = n_succ;
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
      if (lane == 0 && v) atomicAdd(&a.counters[C_PROF0 + 7], (unsigned long long)v);
      if (tid == 0 && a.prof == 2) {  // replaces the three event counts above by: flags+neighbour table / loads+bitmaps / keys
        atomicAdd(&a.counters[C_PROF0 + 3], (unsigned long long)(ts_a - t0));
        atomicAdd(&a.counters[C_PROF0 + 6], (unsigned long long)(ts_b - ts_a));
        atomicAdd(&a.counters[C_PROF0 + 7], (unsigned long long)(t1 - ts_b));
      }
      if (tid == 0) {
        atomicAdd(&a.counters[C_PROF0 + 4], (unsigned long long)tc);
        atomicAdd(&a.counters[C_PROF0 + 5], (unsigned long long)tp);
        atomicAdd(&a.counters[C_PROF0 + 0], (unsigned long long)(t1 - t0));
        atomicAdd(&a.counters[C_PROF0 + 1], (unsigned long long)(t2 - t1));
        atomicAdd(&a.counters[C_PROF0 + 2], (unsigned long long)(t3 - t2));
      }
    }
    if (tid == 0) {
      a.tile_epoch[t] = a.epoch;
      a.cstamp_cur[t] = a.serial;
      acc_levels += level;
      ++acc_visits;
    }
    if (tid < 27 && nbr_dirty[tid]) {
      const int ox = tid / 9 - 1, oy = (tid / 3) % 3 - 1, oz = tid % 3 - 1;
      const int ux = tx + ox, uy = ty + oy, uz = tz + oz;
      if ((unsigned)ux < (unsigned)a.tg.ntx && (unsigned)uy < (unsigned)a.tg.nty && (unsigned)uz < (unsigned)a.tg.ntz &&
          (!PAGED || nb_page[tid] >= 0))
        activate_tile((ux * a.tg.nty + uy) * a.tg.ntz + uz, a.flag_next, a.list_next, a.count_next);
    }
    __syncthreads();
  }
  for (int off = 32; off > 0; off >>= 1) acc_writes += __shfl_down(acc_writes, off);
  if (lane == 0 && acc_writes) atomicAdd(&a.counters[C_WRITES], (unsigned long long)acc_writes);
  if (TRACK) {  // largest d^2 ever stored: bounds the delete scan (k_invalidate); rises rarely, so test before the atomic
    for (int off = 32; off > 0; off >>= 1) acc_maxd2 = max(acc_maxd2, (uint32_t)__shfl_xor((int)acc_maxd2, off));
    if (lane == 0 && (unsigned long long)acc_maxd2 > a.counters[C_MAXD2]) atomicMax(&a.counters[C_MAXD2], (unsigned long long)acc_maxd2);
  }
  if (tid == 0 && acc_visits) {
    atomicAdd(&a.counters[C_SWEEPS], (unsigned long long)acc_levels);
    atomicAdd(&a.counters[C_VISITS], (unsigned long long)acc_visits);
  }
}

#undef K64
#undef KW


}  // namespace fiesta
