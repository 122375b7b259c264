// fiesta_amd/csrc/mask_kernels.hpp -- the MASKED transform: large deltas on PARTIALLY OBSERVED maps.
//
// The reference's BFS only passes through observed voxels (src/ESDFMap.cpp:345,382: a never-observed voxel holds -10000,
// fails every `>` test and is never queued), so on a partially observed map its field is not the transform T of the
// occupied set.  Measured on the verbatim reference (tools/dev/masked_transform_study.py, masked_engine_model.py): it IS T
// on every observed voxel whose straight segment to its nearest obstacle runs through observed voxels, and elsewhere -- the
// "shadows" of the unobserved space, a few percent of the voxels -- it is what 24-neighbour pulls from the voxels around
// make of it.  So a large delta on such a map is served as
//
//   k_obs_cells     per 8^3 cell: no voxel observed / all / some                               reads 1 bit / voxel
//   k_eff_occ       the occupancy bitmap without the obstacles NONE of whose 24 stencil neighbours is observed (they can
//                   hand their id to nobody, :375-391): the sites of the transform
//   <transform>     T of those sites into a SIDE buffer (cell transform nn_kernels.hpp, or the envelope passes)
//   k_mask_certify  per voxel of the side buffer: never observed -> 0xFFFFFFFF; an obstacle -> itself; CERTIFIED (every voxel of
//                   the discrete segment to the winner observed) -> T stays; every other observed voxel keeps what it held
//                   before the update if that obstacle still exists, else "no obstacle", and goes on the repair list
//                   (a second certificate through "portals" -- observed stencil neighbours of a winner hidden behind one
//                   unobserved voxel -- was modelled and dropped: it shrinks the list 4x and was wrong on 70 of 11.8 M voxels)
//   k_repair_*      Jacobi pulls (:349-367: 24 neighbours in stencil order, strict <) over the repair list until nothing
//                   changes: synchronous, deterministic, only listed voxels ever change
//
// and the side buffer becomes the field (pointer swap).  Nothing is committed before the end: a repair list that outgrows
// its buffer leaves the field untouched and the frontier rounds serve the update.
#pragma once
#include "common.hpp"
#include "relax_kernels.hpp"

namespace fiesta {

constexpr int kMaskIters = 48;  // repair iterations one chain of launches can hold (their change counters)
enum MaskCounter {
  MC_ULIST = 0,   // voxels appended to the repair list (may exceed its capacity: then the update is not committed)
  MC_WALKS,       // segment walks (statistics)
  MC_SPARE,       // (unused)
  MC_UNOBS,       // (unused)
  MC_CHANGED0,    // [kMaskIters]: voxels changed in iteration k of the current chain
  MC_COUNT = MC_CHANGED0 + kMaskIters
};

struct MaskArgs {
  Geom g;
  int ncx, ncy, ncz;         // 8^3 cells
  const uint32_t *occbits;   // Exist()
  const uint32_t *obsbits;   // observed at least once
  const uint32_t *effocc;    // k_eff_occ's result
  const uint8_t *cellobs;    // per cell: 0 nothing observed, 1 every voxel (of the grid) observed, 2 mixed
  const vox_t *old;          // the field before this update
  vox_t *out;                // T on entry, the new field on exit
  uint32_t *ulist;           // repair list: linear voxel indices
  uint32_t *uval;            // per entry: the value of the iteration under way
  uint32_t ucap;
  uint32_t *cstamp;          // per cell: tag of the last iteration that changed a voxel in or next to it
  unsigned long long *ctr;   // MaskCounter
  const unsigned long long *failed;  // the cell transform's failure counter (non-zero: T was not written)
};

__device__ __forceinline__ bool bit_test(const uint32_t *bits, const Geom &g, int x, int y, int z) {
  return (bits[g.bitword(x, y, z)] >> (z & 31)) & 1u;
}
__device__ __forceinline__ bool mask_observed(const MaskArgs &a, int x, int y, int z) {
  const uint32_t c = a.cellobs[((int64_t)(x >> 3) * a.ncy + (y >> 3)) * a.ncz + (z >> 3)];
  return c == 1u || (c == 2u && bit_test(a.obsbits, a.g, x, y, z));
}

// ---- observed bitmap from the field (after a restore / load) --------------------------------------------------------------
__global__ __launch_bounds__(256) void k_obs_rebuild(Geom g, const vox_t *coc, uint32_t *obsbits, int64_t nwords) {
  for (int64_t wi = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; wi < nwords; wi += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = wi / g.nzw;
    const int zw = (int)(wi - row * g.nzw);
    uint32_t m = 0;
    for (int b = 0; b < 32; ++b) {
      const int z = 32 * zw + b;
      if (z < g.nz && coc[row * g.nz + z] != kUnobserved) m |= 1u << b;
    }
    obsbits[wi] = m;
  }
}

// ---- late observations that a wave (or their own insertion) has healed since -------------------------------------------------
// A voxel first observed while obstacles exist holds "no obstacle" until a wave reaches it (src/ESDFMap.cpp:246-249: nobody
// queues it): k_fuse marks it in `latebits` and counts it; once it holds an obstacle, or is one, it is an ordinary voxel again.
__global__ __launch_bounds__(256) void k_late_rescan(Geom g, const vox_t *coc, const uint32_t *occbits, uint32_t *latebits, int64_t nwords,
                                                     unsigned long long *late) {
  unsigned healed = 0;
  for (int64_t wi = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; wi < nwords; wi += (int64_t)gridDim.x * blockDim.x) {
    uint32_t m = latebits[wi];
    if (!m) continue;
    const int64_t row = wi / g.nzw;
    const int zw = (int)(wi - row * g.nzw);
    const uint32_t occ = occbits[wi];
    uint32_t keep = 0;
    while (m) {
      const int b = __ffs((int)m) - 1;
      m &= m - 1;
      const vox_t w = coc[row * g.nz + 32 * zw + b];
      if (!(w & kNoCoc) || ((occ >> b) & 1u)) ++healed; else keep |= 1u << b;
    }
    latebits[wi] = keep;
  }
  for (int off = 32; off > 0; off >>= 1) healed += (unsigned)__shfl_xor((int)healed, off);
  if ((threadIdx.x & 63) == 0 && healed) atomicAdd(late, (unsigned long long)(-(long long)healed));
}

// ---- per-cell summary of the observed bitmap ---------------------------------------------------------------------------------
// One wave per (cx, cy) row of cells and chunk of 4 cells along z (= one 32-bit word per voxel row): lane = voxel row of the
// cell row (x = lane / 8, y = lane % 8).
__global__ __launch_bounds__(256) void k_obs_cells(Geom g, int ncx, int ncy, int ncz, const uint32_t *obsbits, uint8_t *cellobs) {
  const int lane = threadIdx.x & 63;
  const int nzw = g.nzw;
  const int64_t items = (int64_t)ncx * ncy * nzw;
  for (int64_t it = blockIdx.x * 4ll + (threadIdx.x >> 6); it < items; it += (int64_t)gridDim.x * 4) {
    const int zw = (int)(it % nzw);
    const int cy = (int)((it / nzw) % ncy), cx = (int)(it / ((int64_t)nzw * ncy));
    const int x = 8 * cx + (lane >> 3), y = 8 * cy + (lane & 7);
    const bool in = x < g.nx && y < g.ny;
    const uint32_t w = in ? obsbits[((int64_t)x * g.ny + y) * nzw + zw] : 0u;
    // bits of the row that lie inside the grid
    const int zleft = g.nz - 32 * zw;
    const uint32_t valid = in ? (zleft >= 32 ? 0xFFFFFFFFu : ((1u << zleft) - 1u)) : 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t wb = (w >> (8 * k)) & 255u, vb = (valid >> (8 * k)) & 255u;
      const bool any = __any((int)(wb != 0u)), all = __all((int)(wb == vb));
      const int cz = 4 * zw + k;
      if (lane == 0 && cz < ncz) cellobs[((int64_t)cx * ncy + cy) * ncz + cz] = (uint8_t)(!any ? 0 : (all ? 1 : 2));
    }
  }
}

// ---- the sites of the transform: obstacles that have somebody to hand their id to -----------------------------------------------
__global__ __launch_bounds__(256) void k_eff_occ(Geom g, const uint32_t *occbits, const uint32_t *obsbits, uint32_t *effocc, int64_t nwords) {
  for (int64_t wi = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; wi < nwords; wi += (int64_t)gridDim.x * blockDim.x) {
    uint32_t m = occbits[wi], keep = 0;
    if (m) {
      const int64_t row = wi / g.nzw;
      const int zw = (int)(wi - row * g.nzw);
      const int y = (int)(row % g.ny), x = (int)(row / g.ny);
      while (m) {
        const int b = __ffs((int)m) - 1;
        m &= m - 1;
        const int z = 32 * zw + b;
        bool any = false;
#define FIESTA_EFF(DX, DY, DZ)                                                                          \
  if (!any && g.in_grid(x + (DX), y + (DY), z + (DZ)) && bit_test(obsbits, g, x + (DX), y + (DY), z + (DZ))) any = true;
        FIESTA_STENCIL24(FIESTA_EFF)
#undef FIESTA_EFF
        if (any) keep |= 1u << b;
      }
    }
    effocc[wi] = keep;
  }
}

// ---- the certificate -------------------------------------------------------------------------------------------------------------
// Samples of the segment v -> s: n = 2 max|d| + 1 steps, sample i at v + round(d i / n) (never a tie: n is odd), i = 1 .. n - 1;
// every voxel of the discrete line is visited, most of them twice -- only a sample that MOVED is looked up.
__device__ inline bool mask_segment_observed(const MaskArgs &a, int vx, int vy, int vz, int sx, int sy, int sz) {
  const int dx = sx - vx, dy = sy - vy, dz = sz - vz;
  const int ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy, az = dz < 0 ? -dz : dz;
  const int m = max(ax, max(ay, az)), n = 2 * m + 1, n2 = 2 * n;
  const int ix = dx < 0 ? -1 : 1, iy = dy < 0 ? -1 : 1, iz = dz < 0 ? -1 : 1;
  int ex = n, ey = n, ez = n, px = vx, py = vy, pz = vz;
  for (int i = 1; i < n; ++i) {
    ex += 2 * ax, ey += 2 * ay, ez += 2 * az;
    bool moved = false;
    if (ex >= n2) ex -= n2, px += ix, moved = true;
    if (ey >= n2) ey -= n2, py += iy, moved = true;
    if (ez >= n2) ez -= n2, pz += iz, moved = true;
    if (moved && !mask_observed(a, px, py, pz)) return false;
  }
  return true;
}
constexpr int kMaskQueue = 1024;   // voxels of a quad waiting for their walk (per wave)
constexpr int kMaskUBuf = 1024;    // repair-list entries a wave collects before it takes a range of the list

// One WAVE per quad (four cells along z: 32 voxels = one 128-byte line per voxel row, one bitmap word per row); lane = (y, four
// consecutive z); waves are persistent and collect their repair-list entries in LDS (one atomic on the list cursor per ~1000
// entries: returning atomics on one address serialise at tens of ns each).
__global__ __launch_bounds__(256) void k_mask_certify(MaskArgs a) {
  __shared__ uint2 s_queue[4][kMaskQueue];
  __shared__ uint32_t s_ubuf[4][kMaskUBuf];
  __shared__ uint32_t s_qn[4], s_un[4];
  const Geom &g = a.g;
  if (a.failed && *a.failed) return;  // T was never written (a cell without a list): the host takes the envelope passes
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint2 *queue = s_queue[wave];
  uint32_t *ubuf = s_ubuf[wave];
  const int nqz = g.nzw;  // quads along z
  const int64_t nquads = (int64_t)a.ncx * a.ncy * nqz;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const int64_t per = (nquads + nwaves - 1) / nwaves;
  const int64_t q0 = (blockIdx.x * 4ll + wave) * per, q1 = min(q0 + per, nquads);
  const int y = lane >> 3, z4 = lane & 7;
  const bool vec = (g.nz & 3) == 0;
  if (lane == 0) s_qn[wave] = 0, s_un[wave] = 0;
  unsigned walks = 0;

  auto flush_u = [&]() {  // the wave's collected entries -> the list (every lane calls)
    const uint32_t n = s_un[wave];
    if (n) {
      uint32_t base = 0;
      if (lane == 0) base = (uint32_t)atomicAdd(&a.ctr[MC_ULIST], (unsigned long long)n);
      base = (uint32_t)__shfl((int)base, 0);
      for (uint32_t i = (uint32_t)lane; i < n; i += 64u)
        if (base + i < a.ucap) a.ulist[base + i] = ubuf[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane == 0) s_un[wave] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  };
  auto drain = [&]() {  // the walks of the queued voxels, 64 at a time (every lane calls)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const uint32_t n = min(s_qn[wave], (uint32_t)kMaskQueue);
    for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
      const uint32_t i = i0 + (uint32_t)lane;
      bool uncert = false;
      uint32_t idx = 0;
      if (i < n) {
        const uint2 e = queue[i];
        idx = e.x;
        const int vz = (int)(idx % (uint32_t)g.nz), vy = (int)((idx / (uint32_t)g.nz) % (uint32_t)g.ny), vx = (int)(idx / ((uint32_t)g.nz * (uint32_t)g.ny));
        int sx, sy, sz;
        unpack_coc(g.wrap, vx + g.gx0, vy + g.gy0, vz + g.gz0, e.y, sx, sy, sz);
        sx -= g.gx0, sy -= g.gy0, sz -= g.gz0;
        ++walks;
        const bool ok = mask_segment_observed(a, vx, vy, vz, sx, sy, sz);
        if (!ok) {  // keeps what it held if that obstacle still exists; repaired from its neighbours afterwards
          vox_t o = a.old[idx] & ~kAct;
          if (!(o & kNoCoc)) {
            int ox, oy, oz;
            unpack_coc(g.wrap, vx + g.gx0, vy + g.gy0, vz + g.gz0, o, ox, oy, oz);
            ox -= g.gx0, oy -= g.gy0, oz -= g.gz0;
            if (!(g.in_grid(ox, oy, oz) && bit_test(a.occbits, g, ox, oy, oz))) o = kInf;
          } else {
            o = kInf;
          }
          a.out[idx] = o;
          uncert = true;
        }
      }
      const unsigned long long mu = __ballot(uncert);
      if (mu) {
        const uint32_t at = s_un[wave];
        if (uncert) ubuf[at + (uint32_t)__popcll(mu & ((1ull << lane) - 1ull))] = idx;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane == 0) s_un[wave] = at + (uint32_t)__popcll(mu);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (at + (uint32_t)__popcll(mu) > (uint32_t)(kMaskUBuf - 64)) flush_u();
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane == 0) s_qn[wave] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  };

  for (int64_t q = q0; q < q1; ++q) {
    const int qz = (int)(q % nqz);
    const int cy = (int)((q / nqz) % a.ncy), cx = (int)(q / ((int64_t)nqz * a.ncy));
    // the four cells' summaries
    uint32_t cs[4];
    bool any_obs = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int cz = 4 * qz + k;
      cs[k] = cz < a.ncz ? (uint32_t)a.cellobs[((int64_t)cx * a.ncy + cy) * a.ncz + cz] : 0u;
      any_obs |= cs[k] != 0u;
    }
    const int Y = 8 * cy + y, Z = 32 * qz + 4 * z4;
    const bool row_in = Y < g.ny && Z < g.nz;
    for (int x = 0; x < 8; ++x) {
      const int X = 8 * cx + x;
      if (X >= g.nx) break;  // (wave-uniform)
      const int64_t base = ((int64_t)X * g.ny + Y) * g.nz + Z;
      if (!any_obs) {  // nothing of this quad was ever observed
        if (row_in) {
          if (vec) *reinterpret_cast<uint4 *>(a.out + base) = uint4{kUnobserved, kUnobserved, kUnobserved, kUnobserved};
          else
            for (int k = 0; k < 4 && Z + k < g.nz; ++k) a.out[base + k] = kUnobserved;
        }
        continue;
      }
      uint32_t w[4] = {kUnobserved, kUnobserved, kUnobserved, kUnobserved}, w0[4];
      uint32_t ob = 0, oc = 0;
      if (row_in) {
        if (vec) {
          const uint4 v = *reinterpret_cast<const uint4 *>(a.out + base);
          w[0] = v.x, w[1] = v.y, w[2] = v.z, w[3] = v.w;
        } else {
          for (int k = 0; k < 4 && Z + k < g.nz; ++k) w[k] = a.out[base + k];
        }
        const int64_t bw = ((int64_t)X * g.ny + Y) * g.nzw + qz;
        ob = (a.obsbits[bw] >> (4 * z4)) & 15u, oc = (a.occbits[bw] >> (4 * z4)) & 15u;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) w0[k] = w[k];
      bool want[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        want[k] = false;
        const int vz = Z + k;
        if (!row_in || vz >= g.nz) continue;
        if (!((ob >> k) & 1u)) {
          w[k] = kUnobserved;
        } else if ((oc >> k) & 1u) {
          w[k] = pack_coc(X + g.gx0, Y + g.gy0, vz + g.gz0);  // an obstacle is its own closest obstacle (:284-286)
        } else if (w[k] & kNoCoc) {
          w[k] = kInf;  // (a map without a site)
        } else {
          // coarse test: every cell the box of v and its winner touches fully observed -> certified without a walk
          int sx, sy, sz;
          unpack_coc(g.wrap, X + g.gx0, Y + g.gy0, vz + g.gz0, w[k], sx, sy, sz);
          sx -= g.gx0, sy -= g.gy0, sz -= g.gz0;
          const int cx0 = min(X, sx) >> 3, cx1 = max(X, sx) >> 3, cy0 = min(Y, sy) >> 3, cy1 = max(Y, sy) >> 3, cz0 = min(vz, sz) >> 3,
                    cz1 = max(vz, sz) >> 3;
          bool full = (cx1 - cx0 + 1) * (cy1 - cy0 + 1) * (cz1 - cz0 + 1) <= 27;
          for (int ux = cx0; ux <= cx1 && full; ++ux)
            for (int uy = cy0; uy <= cy1 && full; ++uy)
              for (int uz = cz0; uz <= cz1 && full; ++uz) full = a.cellobs[((int64_t)ux * a.ncy + uy) * a.ncz + uz] == 1u;
          want[k] = !full;
        }
      }
      // changed words go out now (a walk that ends uncertified overwrites its voxel later: same wave, stores in order)
      if (row_in && (w[0] != w0[0] || w[1] != w0[1] || w[2] != w0[2] || w[3] != w0[3])) {
        if (vec) *reinterpret_cast<uint4 *>(a.out + base) = uint4{w[0], w[1], w[2], w[3]};
        else
          for (int k = 0; k < 4 && Z + k < g.nz; ++k) a.out[base + k] = w[k];
      }
      // queue the voxels that need a walk
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned long long mq = __ballot(want[k]);
        if (mq) {
          const uint32_t at = s_qn[wave];
          if (want[k]) queue[at + (uint32_t)__popcll(mq & ((1ull << lane) - 1ull))] = uint2{(uint32_t)(base + k), w[k]};
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          if (lane == 0) s_qn[wave] = at + (uint32_t)__popcll(mq);
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
      }
      if (s_qn[wave] > (uint32_t)(kMaskQueue - 256)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        drain();
      }
    }
    if (s_qn[wave]) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      drain();
    }
  }
  flush_u();
  for (int off = 32; off > 0; off >>= 1) {
    walks += (unsigned)__shfl_xor((int)walks, off);
  }
  if (lane == 0 && walks) atomicAdd(&a.ctr[MC_WALKS], (unsigned long long)walks);
}

// ---- the repair: Jacobi pulls over the list ------------------------------------------------------------------------------------------
// Iteration `it` of a chain (tag = the stamp of this update's iteration it): pull evaluates the 24 neighbours of every listed
// voxel in or next to a cell that changed in iteration it - 1 (all of them in the first iteration of an update) and leaves the
// better obstacle in uval; commit stores the changes, stamps the cells around them and counts them.  A chain is launched
// whole; an iteration whose predecessor changed nothing returns at once.
__global__ __launch_bounds__(256) void k_repair_pull(MaskArgs a, int it, uint32_t tag_prev, int first) {
  const Geom &g = a.g;
  if (a.failed && *a.failed) return;
  if (it > 0 && a.ctr[MC_CHANGED0 + it - 1] == 0) return;
  const unsigned long long nu = a.ctr[MC_ULIST];
  if (nu > a.ucap) return;  // (the list overflowed: the host gives this update to the rounds)
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < (uint32_t)nu; i += gridDim.x * blockDim.x) {
    const uint32_t idx = a.ulist[i];
    const int z = (int)(idx % (uint32_t)g.nz), y = (int)((idx / (uint32_t)g.nz) % (uint32_t)g.ny), x = (int)(idx / ((uint32_t)g.nz * (uint32_t)g.ny));
    if (!first && a.cstamp[((int64_t)(x >> 3) * a.ncy + (y >> 3)) * a.ncz + (z >> 3)] != tag_prev) {
      a.uval[i] = kUnobserved;
      continue;
    }
    const vox_t cur = a.out[idx];
    int32_t best = (cur & kNoCoc) ? kD2Inf : dist2(g.wrap, x + g.gx0, y + g.gy0, z + g.gz0, cur);
    vox_t bw = cur;
#define FIESTA_PULL(DX, DY, DZ)                                                                      \
  {                                                                                                  \
    const int ux = x + (DX), uy = y + (DY), uz = z + (DZ);                                           \
    if (g.in_grid(ux, uy, uz)) {                                                                     \
      const vox_t w = a.out[g.idx(ux, uy, uz)];                                                      \
      if (!(w & kNoCoc)) {                                                                           \
        int ox, oy, oz;                                                                              \
        unpack_coc(g.wrap, ux + g.gx0, uy + g.gy0, uz + g.gz0, w & ~kAct, ox, oy, oz);               \
        const int32_t d = (x + g.gx0 - ox) * (x + g.gx0 - ox) + (y + g.gy0 - oy) * (y + g.gy0 - oy) + (z + g.gz0 - oz) * (z + g.gz0 - oz); \
        if (d < best && (!g.wrap || d < kD2Cap)) best = d, bw = pack_coc(ox, oy, oz);                 \
      }                                                                                              \
    }                                                                                                \
  }
    FIESTA_STENCIL24(FIESTA_PULL)
#undef FIESTA_PULL
    a.uval[i] = bw != cur ? bw : kUnobserved;
  }
}

__global__ __launch_bounds__(256) void k_repair_commit(MaskArgs a, int it, uint32_t tag) {
  __shared__ uint32_t s_changed;
  const Geom &g = a.g;
  if (a.failed && *a.failed) return;
  if (it > 0 && a.ctr[MC_CHANGED0 + it - 1] == 0) return;
  const unsigned long long nu = a.ctr[MC_ULIST];
  if (nu > a.ucap) return;
  if (threadIdx.x == 0) s_changed = 0;
  __syncthreads();
  unsigned mine = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < (uint32_t)nu; i += gridDim.x * blockDim.x) {
    const vox_t w = a.uval[i];
    if (w == kUnobserved) continue;
    const uint32_t idx = a.ulist[i];
    a.out[idx] = w;
    ++mine;
    const int z = (int)(idx % (uint32_t)g.nz), y = (int)((idx / (uint32_t)g.nz) % (uint32_t)g.ny), x = (int)(idx / ((uint32_t)g.nz * (uint32_t)g.ny));
    // the cells whose voxels have this one in their stencil (radius 2)
    const int cx0 = max(x - 2, 0) >> 3, cx1 = min(x + 2, g.nx - 1) >> 3, cy0 = max(y - 2, 0) >> 3, cy1 = min(y + 2, g.ny - 1) >> 3,
              cz0 = max(z - 2, 0) >> 3, cz1 = min(z + 2, g.nz - 1) >> 3;
    for (int ux = cx0; ux <= cx1; ++ux)
      for (int uy = cy0; uy <= cy1; ++uy)
        for (int uz = cz0; uz <= cz1; ++uz) a.cstamp[((int64_t)ux * a.ncy + uy) * a.ncz + uz] = tag;
  }
  for (int off = 32; off > 0; off >>= 1) mine += (unsigned)__shfl_xor((int)mine, off);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&s_changed, mine);
  __syncthreads();
  if (threadIdx.x == 0 && s_changed) atomicAdd(&a.ctr[MC_CHANGED0 + it], (unsigned long long)s_changed);
}

}  // namespace fiesta
