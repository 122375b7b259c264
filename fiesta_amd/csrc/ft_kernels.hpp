// fiesta_amd/csrc/ft_kernels.hpp -- gfx950 kernels of the BULK UpdateESDF path: the exact feature transform of the
// whole occupied set in three separable passes (see ft_core.hpp for why this equals the reference's fixed point on
// fully observed maps, and dense_map.hip: update_esdf for when it is used).
//
//   k_ft_rows    per x-plane: which z-rows hold an occupied voxel (ordered list + count)        reads 1 bit / voxel
//   k_ft_plane   pass A, per plane x: nearest occupied voxel of the SAME plane for every (y, z)  writes 4 B / voxel
//                (z: nearest set bit of a row, straight from the occupancy bitmap; y: lower envelope over the plane's
//                non-empty rows only -- a scalar loop that skips the ~85 % of rows that are empty in a scatter scene)
//   k_ft_x       pass B, per (y, z) column: lower envelope along x over the planes' candidates    reads 4 B, writes 4 B
//
// Work decomposition: one WAVE owns one column group -- 64 consecutive z (the lane axis, contiguous in HBM: every
// load and store of a step is one 256-byte row segment) -- and walks the scan axis serially; each lane runs its own
// LaneEnvelope with its deque in LDS (ring[slot][lane]: a lane always hits its own bank, no conflicts).  The deque is
// bounded because positions are emitted as soon as they are final; emission is lock-step across the wave (a position
// is written when it is final in all 64 lanes) so that stores stay coalesced.  A deque that outgrows its ring of S
// entries goes on in a backing store in global memory (ft_core.hpp, spill mode): one kernel per pass, whatever the scene.
#pragma once
#include <type_traits>
#include "common.hpp"
#include "ft_core.hpp"

namespace fiesta {

#pragma clang diagnostic ignored "-Winline-asm"  // (m0 on a clobber list: it IS written by the LDS-DMA prefetch of k_ft_x)

// The transform runs over a REGION (nx x ny x nz voxels; at most 1024 per axis with the plain site packing, 2048 with
// the WIDE one, see FtPack) whose occupancy comes from any bitmap -- the map's own (unsharded: region = the whole array) or a
// shard's replica of the GLOBAL bitmap (region = the shard's array grown by a margin, see DenseMap::run_bulk) -- and
// writes the voxels of an OUTPUT box inside the region.
struct FtArgs {
  int nx, ny, nz, nzw, nzc;  // region extents; nzw = 32-bit words, nzc = 64-voxel column groups per z-row
  int gx0, gy0, gz0;         // global coordinates of region voxel (0,0,0): ids are stored in global coordinates
  const uint32_t *src;       // occupancy bitmap; row (x, y) of the region starts at word ((x+sx0)*sny + y+sy0)*snzw + sw0
  int sx0, sy0, sw0, sny, snzw;
  uint16_t *rowlist;         // [nx][ny]: non-empty rows of plane x, ascending
  int32_t *rowcnt;           // [nx]
  uint32_t *inter;           // [nx][ny][nz]: pass A result, y' << 10 | z' (WIDE: << 11), region coordinates
  vox_t *coc;                // output array (the map's voxel words) with extents (., ony, onz); region voxel (x,y,z) is
  int ox0, oy0, oz0;         // output voxel (x - ox0, y - oy0, z - oz0), written iff inside [0,onx) x [0,ony) x [0,onz)
  int onx, ony, onz;
  uint32_t n_items;  // work items 0 .. n_items - 1 (pass A: x * nzc + c, pass B: y * nzc + c), walked with a grid stride
  // backing store of the rings (ft_core.hpp: a deque deeper than its ring): spill_stride bytes per wave of the grid, one
  // 512-byte slot row (64 lanes x 8 B) per counter value, i.e. (longest column + 2) x 512; the wave with linear index
  // blockIdx.x * WAVES + wave owns its slice for every item it walks
  char *spill;
  uint32_t spill_stride;
  unsigned long long *spill_count;  // statistics: items (column groups) that moved ring entries to the backing store
  unsigned long long *maxd2;  // optional: atomicMax of every d^2 written; a voxel without any site counts as 2^30
  __device__ __forceinline__ const uint32_t *row(int x, int y) const {
    return src + ((int64_t)(x + sx0) * sny + (y + sy0)) * snzw + sw0;
  }
};

// LDS ring of one wave: entry[slot][lane], two 32-bit words (ft_core.hpp: e1 = f << SB | start, e2 = the output word
// q << QSH | tag), read and written as one 8-byte access.
// Counters of the envelope advance by kStep = the byte stride of a slot, and a wave's ring is aligned to its own
// size: a slot address is then ONE v_and_or_b32 of the counter (r02: shift, mask, add).
typedef unsigned int ft_u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) ft_u32x2 lds_uint2;
template <int S, int LANES>
struct LdsRing {
  static constexpr int kStep = LANES * 8;                       // bytes from one slot of a lane to its next
  static constexpr uint32_t kBytes = (uint32_t)S * LANES * 8u;  // one wave's ring
  static constexpr bool kAligned = kBytes <= 16384u;            // (bigger rings are not worth the padding in LDS)
  static constexpr uint32_t kMask = (uint32_t)(S - 1) * kStep;
  uint32_t base;  // LDS byte address of this lane's slot 0 (kAligned: the wave's ring starts at a multiple of kBytes)
  char *bk;       // backing store: this lane's 8 bytes of slot row 0; the row of counter c starts c bytes further (LANES == 64)
  bool live;  // LANES < 64: the lanes beyond LANES carry no column and share the ring of lane % LANES -- they may read it
              // (and ignore what they read) but must not take part in the envelope's unconditional store
  __device__ __forceinline__ lds_uint2 *slot(int c) const {
    const uint32_t off = (uint32_t)c & kMask;
    return reinterpret_cast<lds_uint2 *>(kAligned ? (base | off) : (base + off));
  }
  __device__ __forceinline__ void get(int c, uint32_t &e1, uint32_t &e2) const {
    const ft_u32x2 v = *slot(c);
    e1 = v.x, e2 = v.y;
  }
  __device__ __forceinline__ void set(int c, uint32_t e1, uint32_t e2) {
    if (LANES == 64 || live) *slot(c) = ft_u32x2{e1, e2};
  }
  __device__ __forceinline__ void bget(int c, uint32_t &e1, uint32_t &e2) const {
    const uint2 v = *reinterpret_cast<const uint2 *>(bk + (uint32_t)c);
    e1 = v.x, e2 = v.y;
  }
  __device__ __forceinline__ void bset(int c, uint32_t e1, uint32_t e2) { *reinterpret_cast<uint2 *>(bk + (uint32_t)c) = make_uint2(e1, e2); }
};
template <int S, int LANES>
__device__ __forceinline__ LdsRing<S, LANES> make_ring(const uint2 *wave_ring, int lane, char *wave_spill) {
  static_assert(LANES == 64, "the backing store is laid out for full waves");
  return LdsRing<S, LANES>{(uint32_t)(size_t)(wave_ring + lane % LANES), wave_spill + 8 * lane, lane < LANES};  // (low half of a generic LDS pointer: the offset)
}
// Site packing.  Regions of at most 1024 voxels per axis (every unsharded map up to the plain-id limit): ABSOLUTE region
// coordinates, 10 bits each.  WIDE (regions up to 2048: a 1024^3 shard of config 5 plus its margin; grids beyond 1024 per
// axis, whose ids reach 512 voxels anyway, common.hpp): pass A keeps 11-bit absolute (y', z'); pass B carries the site's
// OFFSET from the column, (y' - y, z' - z) as signed 10-bit fields, as the envelope tag -- a candidate farther than 511
// voxels in y or z could never be stored and is dropped on arrival.
template <bool WIDE>
struct FtPack {
  static constexpr int SH = WIDE ? 11 : 10;  // pass A result: y' << SH | z'
};

// wave vote on a bool: the comparison result itself (no 0/1 round trip through a VGPR as with __ballot(int))
__device__ __forceinline__ unsigned long long ft_vote(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }

// ---- per plane: ordered list of the rows that hold at least one occupied voxel -------------------------------------
__global__ __launch_bounds__(256) void k_ft_rows(FtArgs a) {
  __shared__ int wsum[4];
  __shared__ int base;
  const int x = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) base = 0;
  __syncthreads();
  for (int y0 = 0; y0 < a.ny; y0 += 256) {
    const int y = y0 + tid;
    bool any = false;
    if (y < a.ny) {
      const uint32_t *row = a.row(x, y);
      uint32_t acc = 0;
      if ((a.nzw & 3) == 0 && (a.snzw & 3) == 0 && (a.sw0 & 3) == 0) {
        for (int w = 0; w < a.nzw; w += 4) {
          const uint4 q = *reinterpret_cast<const uint4 *>(row + w);
          acc |= q.x | q.y | q.z | q.w;
        }
      } else {
        for (int w = 0; w < a.nzw; ++w) acc |= row[w];
      }
      any = acc != 0;
    }
    const unsigned long long m = ft_vote(any);
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (any) a.rowlist[(int64_t)x * a.ny + off + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)y;
    __syncthreads();
    if (tid == 0) base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  if (tid == 0) {
    a.rowcnt[x] = base;
  }
}

// ---- pass A: in-plane nearest site.  item = x * nzc + c: plane x, lanes z = 64 c + lane ------------------------------
template <int S, int LANES, int WAVES, bool WIDE>
__global__ __launch_bounds__(64 * WAVES) void k_ft_plane(FtArgs a) {
  constexpr int RB = 16;               // bitmap rows staged per batch
  constexpr int RW = WIDE ? 64 : 32;   // 32-bit words of a staged row
  struct __attribute__((aligned(LdsRing<S, LANES>::kAligned ? S * LANES * 8 : 16))) Lds {
    uint2 ring[WAVES][S * LANES];
    uint32_t rowstage[WAVES][RB][RW];
  };
  __shared__ Lds lds;
  auto &ring = lds.ring;
  auto &rowstage = lds.rowstage;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  char *const wave_spill = a.spill + (size_t)(blockIdx.x * WAVES + wave) * a.spill_stride;
  for (uint32_t id = blockIdx.x * WAVES + wave; id < a.n_items; id += gridDim.x * WAVES) {
    constexpr int sub = 0;
    const int x = (int)(id / (uint32_t)a.nzc), c = (int)(id % (uint32_t)a.nzc);
    const int cnt = __builtin_amdgcn_readfirstlane(a.rowcnt[x]);
    if (cnt == 0) continue;  // pass B never reads an empty plane
    const int k = sub * LANES + lane;  // position inside the 64-voxel group
    const int z = 64 * c + k;
    const bool act = lane < LANES && z < a.nz && (unsigned)(z - a.oz0) < (unsigned)a.onz;  // (pass B only reads these)
    ft::LaneEnvelope<S, LdsRing<S, LANES>, FtPack<WIDE>::SH, WIDE> env;  // entries: q = row y', f = (z - z')^2, tag = z'
    env.r = make_ring<S, LANES>(&ring[wave][0], lane, wave_spill);
    env.init();
    env.set_idle(!act);
    int p_out = 0;
    bool sp = false;  // wave-uniform: some lane keeps entries in the backing store (ft_core.hpp, spill mode)
    bool careful = false, counted = false;
    uint32_t *out = a.inter + (int64_t)x * a.ny * a.nz + (act ? z : 0);
    const uint16_t *rows = a.rowlist + (int64_t)x * a.ny;
    for (int i0 = 0; i0 < cnt; i0 += RB) {
      // stage the bitmap words of the next RB non-empty rows: two dependent loads per batch instead of per row
      const int nb = min(RB, cnt - i0);
      const int ylist = rows[min(i0 + lane, cnt - 1)];  // lanes 0..nb-1: the rows of this batch; lane RB: the one after
      const int ynext_batch = (i0 + RB < cnt) ? __builtin_amdgcn_readlane(ylist, RB) : ft::kFarAhead;
      __builtin_amdgcn_wave_barrier();
      for (int j = lane; j < nb * RW; j += 64) {
        const int r = j / RW, w = j % RW;
        const int yr = __shfl(ylist, r);
        rowstage[wave][r][w] = w < a.nzw ? a.row(x, yr)[w] : 0u;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (int r = 0; r < nb; ++r) {
        const int yr = __builtin_amdgcn_readlane(ylist, r);
        const int ynext = (r + 1 < nb) ? __builtin_amdgcn_readlane(ylist, r + 1) : ynext_batch;
        // nearest occupied voxel of this row for every lane: own 64-bit chunk + nearest set bit outside it
        const uint32_t wv = lane < RW ? rowstage[wave][r][lane] : 0u;
        const unsigned long long nonempty = ft_vote(wv != 0u);
        unsigned long long chunk = rowstage[wave][r][2 * c];
        if (2 * c + 1 < a.nzw) chunk |= (unsigned long long)rowstage[wave][r][2 * c + 1] << 32;
        int left_out = -1, right_out = -1;
        const unsigned long long below = nonempty & ((1ull << (2 * c)) - 1ull);
        const unsigned long long above = (2 * c + 2 < 64) ? (nonempty >> (2 * c + 2)) << (2 * c + 2) : 0ull;
        if (below) {
          const int iw = 63 - __clzll((long long)below);
          left_out = 32 * iw + 31 - __clz((int)__builtin_amdgcn_readlane(wv, iw));
        }
        if (above) {
          const int iw = __ffsll((long long)above) - 1;
          right_out = 32 * iw + __ffs((int)__builtin_amdgcn_readlane(wv, iw)) - 1;
        }
        int d;
        const int zp = ft::nearest_in_row(chunk, 64 * c, k, left_out, right_out, d);
        // (WIDE: a site farther than an id reaches -- 512 voxels, common.hpp -- cannot matter: d^2 >= kD2Cap reads
        // "no obstacle" in the end; clamping keeps f inside its 21 bits)
        const int dd = WIDE ? min(d, 1023) : d;
        const int f = ft::mul24(dd, dd), key = yr * yr + f;
        const uint32_t word = ((uint32_t)yr << FtPack<WIDE>::SH) | (uint32_t)(zp & ((1 << FtPack<WIDE>::SH) - 1));
        // a batch of up to 8 rows begins (the rows between two emission runs): plain if every lane has 8 free ring slots
        // and nothing is out in the backing store, else careful -- spill mode (ft_core.hpp)
        if ((r & 7) == 0) {
          careful = sp || ft_vote(env.near_full(8)) != 0;
          if (careful && !sp) env.enter_spill();
        }
        if (!careful) {
          for (;;) {  // pop while any lane wants to
            const bool want = env.wants_pop(yr, key);  // (a lane without a column has an empty ring: never)
            if (!ft_vote(want)) break;
            env.pop(want);
          }
          env.place(act, yr, f, word, key, a.ny, p_out);
        } else {
          for (;;) {
            const bool want = env.wants_pop(yr, key);
            if (!ft_vote(want)) break;
            env.pop_sp(want);
          }
          const bool full = env.full_sp();  // a lane whose ring is full moves its oldest entry to the backing store
          if (ft_vote(full)) {
            env.evict(full);
            if (!counted && lane == 0 && a.spill_count) atomicAdd(a.spill_count, 1ull);
            counted = true;
          }
          env.place(act, yr, f, word, key, a.ny, p_out);
        }
        // positions are emitted every eighth site row (and at the end of a staged batch): a run of emissions ends with a
        // failed finality vote, and sites placed in between need not keep the cached bottom entry current
        if ((r & 7) != 7 && r + 1 < nb) continue;
        const int pend = min(a.ny, ynext);
        auto emit = [&]() {
          if (act && (unsigned)(p_out - a.oy0) < (unsigned)a.ony) *out = env.winner_word();
          out += a.nz;
          ++p_out;
        };
        auto run = [&](auto sp_tag) {
          constexpr bool SP = decltype(sp_tag)::value;
          if (SP)
            env.reload_bottom_sp();
          else
            env.reload_bottom();
          // four positions per finality vote while that holds (ft_core.hpp: finality is monotone) ...
          while (p_out + 3 < pend) {
            const bool fin4 = env.final_at(p_out + 3, ynext);
            if (ft_vote(fin4) != ~0ull) break;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if (SP)
                env.step_to_sp(p_out);
              else
                env.step_to(p_out);
              emit();
            }
          }
          // ... then one by one
          while (p_out < pend) {
            if (SP)
              env.step_to_sp(p_out);
            else
              env.step_to(p_out);
            const bool fin = env.final_at(p_out, ynext);
            if (ft_vote(fin) != ~0ull) break;
            emit();
          }
        };
        if (careful) {
          run(std::true_type{});
          sp = ft_vote(env.spilled()) != 0;  // (nothing left out there: the next batch may be a plain one again)
        } else {
          run(std::false_type{});
        }
      }
    }
  }
}

// ---- pass B: lower envelope along x.  item = y * nzc + c: row y, lanes z = 64 c + lane ------------------------------
// Memory choreography of one wave, per batch of P planes: wait for the batch's candidates (prefetched during the
// previous batch, one 256-byte row segment per plane, by LDS-DMA), copy them to registers, start the DMA of the next
// batch, push the P sites, then emit every position that has become final -- straight to HBM, one row segment each.
// gfx9's vmcnt is one counter for loads and stores, so the wait at the top of a batch also covers the stores of the
// batch before: they are issued last, right in front of it, and the other waves of the SIMD fill the gap.
template <int S, int LANES, int WAVES, bool WIDE, bool TRACK>
__global__ __launch_bounds__(64 * WAVES) void k_ft_x(FtArgs a) {
  constexpr int P = 8;
  // the landing zones of the prefetch (per wave: P planes x 64 lanes x 4 B) come first, so that the LDS address the DMA
  // takes from m0 stays below 64 KB; then the rings, each aligned to its own size
  constexpr int RA = LdsRing<S, LANES>::kAligned ? S * LANES * 8 : 16;
  struct __attribute__((aligned(RA))) Lds {
    uint32_t land[WAVES][P * 64];
    __attribute__((aligned(RA))) uint2 ring[WAVES][S * LANES];
  };
  __shared__ Lds lds;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  char *const wave_spill = a.spill + (size_t)(blockIdx.x * WAVES + wave) * a.spill_stride;
  uint32_t acc_maxd2 = 0;
  // which planes hold any site: bit x of the 2048-bit mask, word w in lane w
  // (lane w holds planes 32 w .. 32 w + 31, built from the planes' row counts: no mask to zero and fill with atomics)
  uint32_t pm = 0;
  if (32 * lane < a.nx) {  // (rowcnt is padded by 64 entries: eight 16-byte loads stay inside it)
    const int4 *rc = reinterpret_cast<const int4 *>(a.rowcnt + 32 * lane);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int4 v = rc[q];
      const int x = 32 * lane + 4 * q;
      pm |= ((x < a.nx && v.x) ? 1u : 0u) << (4 * q) | ((x + 1 < a.nx && v.y) ? 2u : 0u) << (4 * q) |
            ((x + 2 < a.nx && v.z) ? 4u : 0u) << (4 * q) | ((x + 3 < a.nx && v.w) ? 8u : 0u) << (4 * q);
    }
  }
  auto plane_has = [&](const int x) -> bool { return (__builtin_amdgcn_readlane(pm, (x >> 5) & 63) >> (x & 31)) & 1u; };
  for (uint32_t id = blockIdx.x * WAVES + wave; id < a.n_items; id += gridDim.x * WAVES) {
    constexpr int sub = 0;
    const int y = __builtin_amdgcn_readfirstlane((int)(id / (uint32_t)a.nzc)), c = __builtin_amdgcn_readfirstlane((int)(id % (uint32_t)a.nzc));
    const int z = 64 * c + sub * LANES + lane;
    const bool act = lane < LANES && z < a.nz && (unsigned)(z - a.oz0) < (unsigned)a.onz;
    if ((unsigned)(y - a.oy0) >= (unsigned)a.ony) continue;  // (the host only lists rows of the output box)
    // entries: q = plane x', f = (y - y')^2 + (z - z')^2, tag = y' << 10 | z' (WIDE: the offsets y' - y, z' - z)
    ft::LaneEnvelope<S, LdsRing<S, LANES>, 20, WIDE> env;
    env.r = make_ring<S, LANES>(&lds.ring[wave][0], lane, wave_spill);
    env.init();
    env.set_idle(!act);
    uint32_t *land = &lds.land[wave][0];
    int p_out = 0;
    bool sp = false;  // wave-uniform: some lane keeps entries in the backing store (ft_core.hpp, spill mode)
    bool counted = false;
    const int64_t plane = (int64_t)a.ny * a.nz, col = (int64_t)y * a.nz + (act ? z : 0);
    const uint32_t *in = a.inter + col;
    const int64_t oplane = (int64_t)a.ony * a.onz;
    // -> output voxel of region position p_out of this column (valid to dereference only inside the output box): a
    // wave-uniform row pointer that walks the planes (scalar arithmetic) + the lane's byte offset inside the row
    char *orow = reinterpret_cast<char *>(a.coc + ((int64_t)(0 - a.ox0) * a.ony + (y - a.oy0)) * a.onz);
    const uint32_t ooff = (uint32_t)(act ? z - a.oz0 : 0) * (uint32_t)sizeof(vox_t);
    const bool shifted = (a.gx0 | a.gy0 | a.gz0) != 0;
    // emits what is final, one 256-byte row segment per position
    bool no_site = false;  // WIDE: every site of the column was out of an id's reach for this lane (set before the last run)
    auto emit = [&]() {
      const uint32_t s = env.winner_word();
      // region coordinates -> the id: global coordinates modulo 1024 (common.hpp: pack_coc); plain when the region
      // starts at the global origin of a grid within the plain-id limit (every unsharded map up to 1024 per axis)
      vox_t word;
      if (WIDE) {
        const int dy = ((int)(s << 12)) >> 22, dz = ((int)(s << 22)) >> 22;
        word = pack_coc((int)(s >> 20) + a.gx0, y + dy + a.gy0, z + dz + a.gz0);
        if (no_site | (env.winner_cost(p_out) >= kD2Cap)) word = kInf;  // beyond the reach of an id on such grids
      } else {
        word = shifted ? pack_coc((int)(s >> 20) + a.gx0, (int)((s >> 10) & 1023u) + a.gy0, (int)(s & 1023u) + a.gz0) : s;
      }
      const bool inbox = (unsigned)(p_out - a.ox0) < (unsigned)a.onx;
      if (act && inbox) *reinterpret_cast<vox_t *>(orow + ooff) = word;
      if (TRACK && act && inbox) acc_maxd2 = max(acc_maxd2, (uint32_t)env.winner_cost(p_out));
      orow += oplane * (int64_t)sizeof(vox_t);
      ++p_out;
    };
    auto run = [&](auto sp_tag, const int x_next) {
      constexpr bool SP = decltype(sp_tag)::value;
      const int pend = min(a.nx, x_next);
      // (the sites of this batch were placed without keeping the cached bottom up to date)
      if (SP)
        env.reload_bottom_sp();
      else
        env.reload_bottom();
      // four positions per finality vote while that holds (ft_core.hpp: finality is monotone) ...
      while (p_out + 3 < pend) {
        const bool fin4 = env.final_at(p_out + 3, x_next);
        if (ft_vote(fin4) != ~0ull) break;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (SP)
            env.step_to_sp(p_out);
          else
            env.step_to(p_out);
          emit();
        }
      }
      // ... then one by one
      while (p_out < pend) {
        if (SP)
          env.step_to_sp(p_out);
        else
          env.step_to(p_out);
        const bool fin = env.final_at(p_out, x_next);
        if (ft_vote(fin) != ~0ull) break;
        emit();
      }
    };
    // The planes' candidates are prefetched a batch ahead by LDS-DMA (global_load_lds_dword: the data lands in the wave's
    // slice of LDS -- lane i at byte m0 + 4 i -- and no register is in flight), issued and waited for BY HAND in inline
    // asm.  hipcc must not see the transfer: it tracks an ordinary load by draining vmcnt to 0 in front of every inner
    // loop (one full HBM round trip per step), and a visible LDS-DMA by a vmcnt(0) in front of every LDS read.  (An
    // earlier version kept the batch in flight in VGPRs named by asm operands; the register allocator is free to copy
    // such a value before it has arrived, and with one more live range it did.)  Loads are unconditional: a plane
    // without sites is simply never looked at.
    uint32_t w[P];
    const uint32_t land_lds = (uint32_t)(size_t)land;  // LDS byte offset: the low half of the generic address
    auto issue = [&](const int xb) {
#pragma unroll
      for (int u = 0; u < P; ++u) {
        const uint32_t *ptr = in + (int64_t)min(xb + u, a.nx - 1) * plane;
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, off" : : "s"(land_lds + u * 256), "v"(ptr) : "memory", "m0");
      }
    };
    issue(0);
    for (int x0 = 0; x0 < a.nx; x0 += P) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the batch has landed (and the previous batch's stores are out)
#pragma unroll
      for (int u = 0; u < P; ++u) w[u] = land[u * 64 + lane];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // ... and is in registers before the next one may land
      issue(x0 + P);
      // the batch's sites, then the positions they made final.  A batch is PLAIN if every lane has P free ring slots and
      // nothing is out in the backing store -- no room check at any site -- else CAREFUL: spill mode (ft_core.hpp)
      auto sites = [&](auto sp_tag) {
        constexpr bool SP = decltype(sp_tag)::value;
#pragma unroll
        for (int u = 0; u < P; ++u) {
          const int x = x0 + u;
          if (x >= a.nx) break;
          if (plane_has(x)) {
            uint32_t tag;
            int dy, dz;
            bool use = act;
            if (WIDE) {  // offsets from the column; a site out of an id's reach in y or z is no candidate
              dy = (int)(w[u] >> 11) - y, dz = (int)(w[u] & 2047u) - z;
              use = use && (unsigned)(dy + 511) < 1023u && (unsigned)(dz + 511) < 1023u;
              tag = (((uint32_t)dy & 1023u) << 10) | ((uint32_t)dz & 1023u) | ((uint32_t)x << 20);
              dy = use ? dy : 0, dz = use ? dz : 0;  // (f stays inside its bit field)
            } else {
              tag = (w[u] & 0xFFFFFu) | ((uint32_t)x << 20);  // (the output word; x is wave-uniform: one v_and_or)
              dy = y - (int)((w[u] >> 10) & 1023u), dz = z - (int)(w[u] & 1023u);
            }
            const int f = ft::mul24(dy, dy) + ft::mul24(dz, dz), key = env.key_of(x, f);
            const int pkey = (WIDE && !use) ? env.kNoPop : key;  // (plain packing: use == act, and an idle lane's ring is empty)
            for (;;) {  // pop while any lane wants to
              const bool want = env.wants_pop(x, pkey);
              if (!ft_vote(want)) break;
              if (SP)
                env.pop_sp(want);
              else
                env.pop(want);
            }
            if (SP) {  // a lane whose ring is full moves its oldest entry to the backing store
              const bool full = env.full_sp();
              if (ft_vote(full)) {
                env.evict(full);
                if (!counted && lane == 0 && a.spill_count) atomicAdd(a.spill_count, 1ull);
                counted = true;
              }
            }
            env.place(use, x, f, tag, key, a.nx, p_out);
          }
        }
      };
      // positions are emitted once per batch: a failed finality vote (the way every run ends) costs as much as an
      // emission, and a batch's stores then sit right in front of the next batch's wait
      if (sp || ft_vote(env.near_full(P)) != 0) {
        if (!sp) env.enter_spill();
        sites(std::true_type{});
        run(std::true_type{}, min(x0 + P, a.nx));
        sp = ft_vote(env.spilled()) != 0;  // (nothing left out there: the next batch may be a plain one again)
      } else {
        sites(std::false_type{});
        run(std::false_type{}, min(x0 + P, a.nx));
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the last, unused prefetch must not land in the next item's batch)
    if (ft_vote(act && !env.empty())) {
      if (WIDE) {  // a lane that found no site in reach must not hold up its neighbours' last run: it reads "no obstacle"
        no_site = act & env.empty();
        if (no_site) env.set_idle(true);
        if (TRACK && no_site) acc_maxd2 = 1u << 30;
      }
      if (sp)
        run(std::true_type{}, ft::kFarAhead);
      else
        run(std::false_type{}, ft::kFarAhead);
    } else {  // no occupied voxel anywhere in the region: "observed, no obstacle"
      for (int p = 0; p < a.nx; ++p) {
        if (act && (unsigned)(p - a.ox0) < (unsigned)a.onx) *reinterpret_cast<vox_t *>(orow + ooff) = kInf;
        orow += oplane * (int64_t)sizeof(vox_t);
      }
      if (TRACK && act) acc_maxd2 = 1u << 30;
    }
  }
  if (TRACK) {
    for (int off = 32; off > 0; off >>= 1) acc_maxd2 = max(acc_maxd2, (uint32_t)__shfl_xor((int)acc_maxd2, off));
    if (lane == 0 && (unsigned long long)acc_maxd2 > *a.maxd2) atomicMax(a.maxd2, (unsigned long long)acc_maxd2);
  }
}

}  // namespace fiesta
