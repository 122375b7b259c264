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
//   k_cell_dist     per cell: how far the nearest cell that is not fully observed is, and which of its 26 neighbours are
//   k_eff_occ       the occupancy bitmap without the obstacles NONE of whose 24 stencil neighbours is observed (they can
//                   hand their id to nobody, :375-391): the sites of the transform
//   <transform>     T of those sites into a SIDE buffer (cell transform nn_kernels.hpp, or the envelope passes)
//   k_mask_classify per voxel of the side buffer, streaming: never observed -> 0xFFFFFFFF; an obstacle -> itself; CERTIFIED at once
//                   where every cell between the voxel and its winner is fully observed; the others are queued for
//   k_mask_walk     the certificate proper: every voxel of the discrete segment to the winner observed -> T stays; an
//                   uncertified voxel is marked for repair (ubits)
//                   -- or, second chance, the winner has a PORTAL (an observed stencil neighbour that can only hold it) the
//                   voxel's way to is clear (mask_portal_certificate)
//   k_mask_cells    the cells that hold a marked voxel; a marked voxel keeps what it held before the update if that obstacle
//                   still exists, else "no obstacle"
//   k_repair_cell / k_repair_commit   Jacobi pulls (:349-367: 24 neighbours in stencil order, strict <) on the marked voxels:
//                   a WAVE stages a cell + its 2-voxel halo in LDS and iterates it to local quiescence against the halo as
//                   the global iteration found it (block Jacobi); results go to a second buffer and are committed by a
//                   launch of their own -- synchronous, deterministic, only marked voxels ever change
//
// and the side buffer becomes the field (pointer swap).  tools/dev/masked_engine_model.py is the same algorithm in numpy; the
// GPU field equals it voxel for voxel (tools/dev/masked_gpu_check.py).
#pragma once
#include <climits>

#include "common.hpp"
#include "relax_kernels.hpp"

namespace fiesta {

constexpr int kMaskIters = 24;  // repair iterations one chain of launches can hold (their change counters)
constexpr int kMaskSegs = 256;  // segments of the walk list (one cursor each, 128 bytes apart: appends spread over the L2 channels)
constexpr int kMaskSub = 6;     // cell-local Jacobi steps per global repair iteration
enum MaskCounter {
  MC_MARKED = 0,  // voxels marked for repair
  MC_WALKS,       // segment walks
  MC_OVERFLOW,    // walk-list segments that ran out of room (non-zero: the update is not committed)
  MC_QUADS,       // cells on the repair list
  MC_CHANGED0,    // [kMaskIters]: voxels changed in iteration k of the current chain
  MC_SEG0 = MC_CHANGED0 + kMaskIters,  // [kMaskSegs * 16]: the segments' cursors (every 16th word)
  MC_COUNT = MC_SEG0 + kMaskSegs * 16
};

struct MaskArgs {
  Geom g;
  int ncx, ncy, ncz;         // 8^3 cells
  const uint32_t *occbits;   // Exist()
  const uint32_t *obsbits;   // observed at least once
  const uint32_t *effocc;    // the sites of the transform (k_eff_occ)
  uint2 *ptab;               // the portals of the HIDDEN sites (k_portal_sites): open addressing, {site word, 24-bit mask of stencil
  uint32_t ptab_mask;        //   directions}, 0xFFFFFFFF = empty; slots - 1
  const uint8_t *cellobs;    // per cell: 0 nothing observed, 1 every voxel (of the grid) observed, 2 mixed
  const uint8_t *celldist;   // per cell: 0 not fully observed, k = every cell within k - 1 cells (Chebyshev) is (k <= 3)
  const uint32_t *cellnb;    // per cell: bit (dx + 1) * 9 + (dy + 1) * 3 + (dz + 1) = the neighbour cell (dx, dy, dz) is fully observed
                             // (cells outside the grid count as observed: nothing to cross there)
  const unsigned long long *cellst;  // per cell: two bits per neighbour cell (same order): its cellobs (a cell outside the grid: 1)
  vox_t *old;                // the field before this update; the repair's second buffer afterwards
  vox_t *out;                // T on entry, the new field on exit
  uint32_t *ubits;           // 1 bit / voxel: marked for repair
  uint2 *walks;              // walk list: (voxel x << 20 | y << 10 | z -- no axis of such a map exceeds 1024 --, winner), kMaskSegs segments of seg_cap entries
  uint32_t seg_cap;
  uint32_t *uq;              // cells that hold a marked voxel
  uint32_t *qstamp[2];       // per cell: tag of the last iteration (by parity) that changed a voxel in or next to it
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
        // (all 24 words asked for at once: a quarter of config 2-partial's obstacles stand inside never-observed blocks, and a lane
        //  that stops at the first observed neighbour goes through two dozen dependent loads for each of them)
        uint32_t any = 0;
#define FIESTA_EFF(DX, DY, DZ)                                                                          \
  {                                                                                                     \
    const int ux = x + (DX), uy = y + (DY), uz = z + (DZ);                                              \
    const bool in = g.in_grid(ux, uy, uz);                                                              \
    const uint32_t w = obsbits[in ? g.bitword(ux, uy, uz) : wi];                                        \
    any |= in ? (w >> (uz & 31)) & 1u : 0u;                                                             \
  }
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
// every voxel of the discrete line is visited, most of them twice -- only a sample that MOVED is looked up.  Walked from the
// winner's end: more than half of the walks that fail do so next to a winner that sits behind an unobserved voxel.
// The cells around the voxel a walk starts from: their summaries in a register (a walk rarely leaves the 3^3 cells around its
// voxel -- a load per cell crossed otherwise, each behind the one before).
struct CellView {
  int cx, cy, cz;
  unsigned long long st;
};
__device__ __forceinline__ CellView cell_view(const MaskArgs &a, int vx, int vy, int vz) {
  CellView cv{vx >> 3, vy >> 3, vz >> 3, 0ull};
  cv.st = a.cellst[((int64_t)cv.cx * a.ncy + cv.cy) * a.ncz + cv.cz];
  return cv;
}
__device__ __forceinline__ uint32_t cell_state(const MaskArgs &a, const CellView &cv, int ccx, int ccy, int ccz) {
  const unsigned ox = (unsigned)(ccx - cv.cx + 1), oy = (unsigned)(ccy - cv.cy + 1), oz = (unsigned)(ccz - cv.cz + 1);
  if (ox < 3u && oy < 3u && oz < 3u) return (uint32_t)(cv.st >> (2u * (ox * 9u + oy * 3u + oz))) & 3u;
  return a.cellobs[((int64_t)ccx * a.ncy + ccy) * a.ncz + ccz];
}
// (a build with -DFIESTA_PROBE counts the walks' work in spare words of the counter block; run_masked prints them)
#if defined(FIESTA_PROBE)
#define PROBE_ADD(K, N) atomicAdd(&a.ctr[MC_CHANGED0 + 12 + (K)], (unsigned long long)(N))
#else
#define PROBE_ADD(K, N)
#endif
// (the reference's stencil has no (1, 1, 1) step, src/ESDFMap.cpp:33-60: where two consecutive samples differ along all three axes
//  an id crosses in two hops, through one of the six voxels between them -- one of those has to be observed as well.  On a map
//  observed through view cones two voxels in 1.5 million were certified across such a diagonal with nothing observed between, and
//  the repair carried their ids to 630 voxels no run of the reference gives them to, tests/golden/c2_sensor_256_envelope.npz)
__device__ __forceinline__ bool vox_observed(const MaskArgs &a, const CellView &, int x, int y, int z) { return bit_test(a.obsbits, a.g, x, y, z); }
__device__ __forceinline__ bool mask_segment_samples(const MaskArgs &a, const CellView &cv, int vx, int vy, int vz, int sx, int sy, int sz) {
  const int dx = sx - vx, dy = sy - vy, dz = sz - vz;
  const int ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy, az = dz < 0 ? -dz : dz;
  const int m = max(ax, max(ay, az)), n = 2 * m + 1, n2 = 2 * n;
  const int ix = dx < 0 ? -1 : 1, iy = dy < 0 ? -1 : 1, iz = dz < 0 ? -1 : 1;
  // sample i: v + sign * floor((2 |d| i + n) / 2n) per axis; start at i = n - 1 -- that is s itself: 2 |d| (n - 1) + n =
  // |d| 2n + (n - 2 |d|) and 0 < n - 2 |d| < 2n -- and step down
  int ex = n - 2 * ax, ey = n - 2 * ay, ez = n - 2 * az;
  int px = sx, py = sy, pz = sz;
  // the summary of the cell the walk is in stays in a register
  int ccx = px >> 3, ccy = py >> 3, ccz = pz >> 3;
  uint32_t cst = cell_state(a, cv, ccx, ccy, ccz);
  if (cst == 0u || (cst == 2u && !bit_test(a.obsbits, a.g, px, py, pz))) return false;
  bool diagonals = false;
  for (int i = n - 2; i >= 1; --i) {
    ex -= 2 * ax, ey -= 2 * ay, ez -= 2 * az;
    const bool mvx = ex < 0, mvy = ey < 0, mvz = ez < 0;
    if (!(mvx || mvy || mvz)) continue;
    if (mvx) ex += n2, px -= ix;
    if (mvy) ey += n2, py -= iy;
    if (mvz) ez += n2, pz -= iz;
    if ((px >> 3) != ccx || (py >> 3) != ccy || (pz >> 3) != ccz) {
      ccx = px >> 3, ccy = py >> 3, ccz = pz >> 3;
      cst = cell_state(a, cv, ccx, ccy, ccz);
    }
    if (cst == 0u || (cst == 2u && !bit_test(a.obsbits, a.g, px, py, pz))) return false;
    diagonals |= mvx && mvy && mvz;
  }
  if (!diagonals) return true;
  // every sample is observed; the (1, 1, 1)-diagonals between samples in a walk of their own (few segments have one, fewer fail here)
  ex = n - 2 * ax, ey = n - 2 * ay, ez = n - 2 * az;
  px = sx, py = sy, pz = sz;
  for (int i = n - 2; i >= 1; --i) {
    ex -= 2 * ax, ey -= 2 * ay, ez -= 2 * az;
    const bool mvx = ex < 0, mvy = ey < 0, mvz = ez < 0;
    const int qx = px, qy = py, qz = pz;  // (the sample before)
    if (mvx) ex += n2, px -= ix;
    if (mvy) ey += n2, py -= iy;
    if (mvz) ez += n2, pz -= iz;
    if (mvx && mvy && mvz &&
        !(vox_observed(a, cv, px, qy, qz) || vox_observed(a, cv, qx, py, qz) || vox_observed(a, cv, qx, qy, pz) ||
          vox_observed(a, cv, px, py, qz) || vox_observed(a, cv, px, qy, pz) || vox_observed(a, cv, qx, py, pz)))
      return false;
  }
  return true;
}

// The same answer from the CELLS the samples cross, where that decides it: axis a of sample i is v + sign floor((2 |d_a| i + n) /
// 2n), so the walk enters its k-th voxel along a at sample ceil(n (2k - 1) / 2 |d_a|) -- the cell boundaries it crosses, in the
// order it crosses them, cost a division each instead of a step per sample.  A cell nothing was observed in refuses the
// segment, a fully observed one passes; the first PARTLY observed cell sends the walk to the samples (mask_segment_samples).
__device__ __forceinline__ uint32_t ceil_div_small(uint32_t num, uint32_t den) {  // ceil(num / den), num < 2^16, 0 < den
  const uint32_t x = num + den - 1u;
  uint32_t q = (uint32_t)((float)x * __builtin_amdgcn_rcpf((float)den));  // (within one of the quotient: x < 2^17 is exact in a float)
  int r = (int)(x - q * den);
  if (r < 0) --q, r += (int)den;
  if (r >= (int)den) ++q;
  return q;
}
__device__ __forceinline__ int mask_segment_cells(const MaskArgs &a, const CellView &cv, int vx, int vy, int vz, int sx, int sy, int sz) {
  const int dx = sx - vx, dy = sy - vy, dz = sz - vz;
  const int ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy, az = dz < 0 ? -dz : dz;
  const int m = max(ax, max(ay, az)), n = 2 * m + 1;
  const uint32_t st0 = (uint32_t)(cv.st >> 26) & 3u;  // (v's own cell: the 14th of the 27)
  if (m > 127 || st0 != 1u) return 2;
  const int ix = dx < 0 ? -1 : 1, iy = dy < 0 ? -1 : 1, iz = dz < 0 ? -1 : 1;
  // the next voxel offset along each axis that lies in another cell, and the sample that reaches it
  int kx = dx < 0 ? (vx & 7) + 1 : 8 - (vx & 7), ky = dy < 0 ? (vy & 7) + 1 : 8 - (vy & 7), kz = dz < 0 ? (vz & 7) + 1 : 8 - (vz & 7);
  int tx = kx <= ax ? (int)ceil_div_small((uint32_t)(n * (2 * kx - 1)), (uint32_t)(2 * ax)) : INT_MAX;
  int ty = ky <= ay ? (int)ceil_div_small((uint32_t)(n * (2 * ky - 1)), (uint32_t)(2 * ay)) : INT_MAX;
  int tz = kz <= az ? (int)ceil_div_small((uint32_t)(n * (2 * kz - 1)), (uint32_t)(2 * az)) : INT_MAX;
  int ccx = cv.cx, ccy = cv.cy, ccz = cv.cz;
  for (;;) {
    const int t = min(tx, min(ty, tz));
    if (t == INT_MAX) return 1;  // (every crossing lies at a sample <= n - 1, the winner itself)
    PROBE_ADD(0, 1);
    if (tx == t && ty == t && tz == t) {
      PROBE_ADD(1, 1);
      // through the corner of a cell: the two samples lie a (1, 1, 1)-diagonal apart and the six voxels between them in the six
      // other cells around that corner -- one of them has to be observed (any other diagonal step has such a voxel in the cell it
      // leaves, fully observed here); one walk in a thousand comes this way
      uint32_t any1 = 0, any2 = 0;
#pragma unroll
      for (int h = 1; h < 7; ++h) {  // the six ways to have moved along one or two of the axes
        // (from the register alone: a corner beyond the 3^3 cells around the voxel is left to the samples)
        const unsigned ox = (unsigned)(ccx + ((h & 1) ? ix : 0) - cv.cx + 1), oy = (unsigned)(ccy + ((h & 2) ? iy : 0) - cv.cy + 1), oz = (unsigned)(ccz + ((h & 4) ? iz : 0) - cv.cz + 1);
        const uint32_t st = (ox < 3u && oy < 3u && oz < 3u) ? (uint32_t)(cv.st >> (2u * (ox * 9u + oy * 3u + oz))) & 3u : 2u;
        any1 |= (uint32_t)(st == 1u), any2 |= (uint32_t)(st == 2u);
      }
      if (!any1) return any2 ? 2 : 0;
    }
    if (tx == t) {
      ccx += ix, kx += 8;
      tx = kx <= ax ? (int)ceil_div_small((uint32_t)(n * (2 * kx - 1)), (uint32_t)(2 * ax)) : INT_MAX;
    }
    if (ty == t) {
      ccy += iy, ky += 8;
      ty = ky <= ay ? (int)ceil_div_small((uint32_t)(n * (2 * ky - 1)), (uint32_t)(2 * ay)) : INT_MAX;
    }
    if (tz == t) {
      ccz += iz, kz += 8;
      tz = kz <= az ? (int)ceil_div_small((uint32_t)(n * (2 * kz - 1)), (uint32_t)(2 * az)) : INT_MAX;
    }
    const uint32_t cst = cell_state(a, cv, ccx, ccy, ccz);
    if (cst == 0u) return 0;
    if (cst == 2u) return 2;
  }
}

// The second certificate, for a winner s hidden behind an unobserved voxel: s hands its id to an observed stencil neighbour p --
// a PORTAL -- and through it to everybody whose way to p is clear.  Only a HIDDEN site has portals (a never-observed voxel among its
// 26 neighbours); they are found once per update (k_portal_sites) and looked up by the walk.  v keeps T(v) = s if for some stencil direction e (in stencil
// order) p = s + e is inside the grid, observed, free, nearer to v than s, has NO other site within |e| of it (then p can only
// hold s: s pushes it there itself, src/ESDFMap.cpp:375-391), and every voxel of the discrete segment v -> p is observed, free
// and has s as its own winner (`out` still holds T for every observed free voxel: nobody writes it between k_mask_classify and
// k_mask_cells).  On config 2's partially observed scene four fifths of the voxels the straight segment leaves uncertified have
// such a winner; against the envelope of the reference's runs (tests/golden/c2_partial_256_envelope.npz) this certificate takes
// the voxels closer than every run from 247 / 180 to 0 / 0 -- the straight segment alone repaired those regions by pulls, whose
// ties fall differently from the reference's arrivals.
// (walked from the portal's end, where the never-observed voxels are; the words of the samples go out six at a time.  The straight
//  certificate's rule for (1, 1, 1)-diagonals is not applied here: every sample of this path holds the winner, each got it from
//  somewhere, and on the fixtures the rule changes nothing here -- tests/golden/c2_sensor_256_envelope.npz -- at +50 % walk time.  A check of the
//  cells' summaries ahead of the loads would spare a refused path its words -- one path in nine -- at a third more instructions
//  for every path: the walk kernel is bound by its VALU instructions, not by these loads)
__device__ __forceinline__ int mask_path_in_cell(const MaskArgs &a, int vx, int vy, int vz, int ux, int uy, int uz, vox_t ws) {  // 0 no, 1 yes, 2 yes if its diagonals hold
  const Geom &g = a.g;
  const int dx = ux - vx, dy = uy - vy, dz = uz - vz;
  const int ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy, az = dz < 0 ? -dz : dz;
  const int m = max(ax, max(ay, az)), n = 2 * m + 1, n2 = 2 * n;
  if (m == 0) return 1;
  const int ix = dx < 0 ? -1 : 1, iy = dy < 0 ? -1 : 1, iz = dz < 0 ? -1 : 1;
  int ex = n - 2 * ax, ey = n - 2 * ay, ez = n - 2 * az;  // sample n - 1 is the portal itself (see mask_segment_samples)
  // the walk only needs the sample's WORD: its linear index moves by a stride when an axis steps (no axis of such a map exceeds
  // 1024: the index fits 30 bits)
  const int stx = ix * g.ny * g.nz, sty = iy * g.nz, stz = iz;
  int pidx = (ux * g.ny + uy) * g.nz + uz;
  int i = n - 1;     // the sample pidx is
  bool pend = true;  // ... and it has not been looked at yet
  int diagonals = 0;
  PROBE_ADD(2, 1);
  for (;;) {
    uint32_t w[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      w[k] = ws;
#pragma unroll
      for (int t = 0; t < 2; ++t)
        if (!pend && i > 1) {
          --i;
          ex -= 2 * ax, ey -= 2 * ay, ez -= 2 * az;
          const int mx = ex >> 31, my = ey >> 31, mz = ez >> 31;  // (all ones where the axis steps)
          ex += mx & n2, ey += my & n2, ez += mz & n2;
          pidx -= (mx & stx) + (my & sty) + (mz & stz);
          diagonals |= mx & my & mz;
          pend = (mx | my | mz) != 0;
        }
      if (pend) {
        PROBE_ADD(3, 1);
        // (observed and free follow from the word: a never-observed voxel holds kUnobserved, an obstacle itself)
        w[k] = a.out[pidx] & ~kAct;
        pend = false;
      }
    }
    if (w[0] != ws || w[1] != ws || w[2] != ws || w[3] != ws || w[4] != ws || w[5] != ws) return 0;
    if (i <= 1) return diagonals ? 2 : 1;
  }
}
// The (1, 1, 1)-diagonals of a path whose samples all hold the winner: the reference's stencil has no such step (src/ESDFMap.cpp:33-60),
// the id crosses one in two hops -- one of the six voxels between the two samples has to hold it too (or be the winner itself).
// A walk of its own, for the third of the paths that have such a step (on config 2-partial at 512^3 two voxels in 2.8 million were
// certified across a diagonal nobody relays the id over, and the repair carried it on to 700 more).
__device__ __forceinline__ bool mask_path_diagonals(const MaskArgs &a, int vx, int vy, int vz, int ux, int uy, int uz, vox_t ws) {
  const Geom &g = a.g;
  const int dx = ux - vx, dy = uy - vy, dz = uz - vz;
  const int ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy, az = dz < 0 ? -dz : dz;
  const int m = max(ax, max(ay, az)), n = 2 * m + 1, n2 = 2 * n;
  const int ix = dx < 0 ? -1 : 1, iy = dy < 0 ? -1 : 1, iz = dz < 0 ? -1 : 1;
  int ex = n - 2 * ax, ey = n - 2 * ay, ez = n - 2 * az;
  const int stx = ix * g.ny * g.nz, sty = iy * g.nz, stz = iz;
  int pidx = (ux * g.ny + uy) * g.nz + uz;
  for (int i = n - 2; i >= 1; --i) {
    ex -= 2 * ax, ey -= 2 * ay, ez -= 2 * az;
    const int mx = ex >> 31, my = ey >> 31, mz = ez >> 31;
    ex += mx & n2, ey += my & n2, ez += mz & n2;
    if (mx & my & mz) {
      const bool via = (a.out[pidx - stx] & ~kAct) == ws || (a.out[pidx - sty] & ~kAct) == ws || (a.out[pidx - stz] & ~kAct) == ws ||
                       (a.out[pidx - stx - sty] & ~kAct) == ws || (a.out[pidx - stx - stz] & ~kAct) == ws || (a.out[pidx - sty - stz] & ~kAct) == ws;
      if (!via) return false;
    }
    pidx -= (mx & stx) + (my & sty) + (mz & stz);
  }
  return true;
}
__device__ __forceinline__ uint32_t portal_hash(vox_t w) { return (w * 0x9E3779B1u) ^ (w >> 15); }
// the stencil's directions by index, from registers: (d + 2) of twelve directions, three bits each, per 64-bit constant
struct StencilCode {
  unsigned long long x[2], y[2], z[2];
};
constexpr StencilCode stencil_code() {
  StencilCode c{{0, 0}, {0, 0}, {0, 0}};
  int k = 0;
#define FIESTA_CODE(DX, DY, DZ)                                          \
  c.x[k / 12] |= (unsigned long long)((DX) + 2) << (3 * (k % 12));       \
  c.y[k / 12] |= (unsigned long long)((DY) + 2) << (3 * (k % 12));       \
  c.z[k / 12] |= (unsigned long long)((DZ) + 2) << (3 * (k % 12));       \
  ++k;
  FIESTA_STENCIL24(FIESTA_CODE)
#undef FIESTA_CODE
  return c;
}
__device__ __forceinline__ void stencil_dir(int bit, int &dx, int &dy, int &dz) {
  constexpr StencilCode code = stencil_code();
  const int hi = bit >= 12, sh = 3 * (bit - 12 * hi);
  dx = (int)(((hi ? code.x[1] : code.x[0]) >> sh) & 7ull) - 2;
  dy = (int)(((hi ? code.y[1] : code.y[0]) >> sh) & 7ull) - 2;
  dz = (int)(((hi ? code.z[1] : code.z[0]) >> sh) & 7ull) - 2;
}
// the winner's portals that are nearer to v than the winner (0: it is not hidden, or no neighbour qualifies)
__device__ inline uint32_t mask_portals_of(const MaskArgs &a, int vx, int vy, int vz, int sx, int sy, int sz, vox_t ws) {
  uint32_t mask = 0;
  for (uint32_t slot = portal_hash(ws) & a.ptab_mask;; slot = (slot + 1u) & a.ptab_mask) {
    const uint2 e = a.ptab[slot];
    if (e.x == ws) {
      mask = e.y;
      break;
    }
    if (e.x == 0xFFFFFFFFu) break;
  }
  // nearer to v than the winner: |s + e - v|^2 < |s - v|^2, that is 2 e.(s - v) + |e|^2 < 0 -- all 24 at once, no loop
  const int wx = sx - vx, wy = sy - vy, wz = sz - vz;
  uint32_t keep = 0;
  int bit = 0;
#define FIESTA_NEARER(DX, DY, DZ)                                                                                  \
  keep |= (uint32_t)(2 * ((DX) * wx + (DY) * wy + (DZ) * wz) + ((DX) * (DX) + (DY) * (DY) + (DZ) * (DZ)) < 0) << bit; \
  ++bit;
  FIESTA_STENCIL24(FIESTA_NEARER)
#undef FIESTA_NEARER
  return keep & mask;
}
// the candidate nearest to v (any order gives the same answer to "does one of them certify v": the nearest does most often);
// |s + e - v|^2 - |s - v|^2 = 2 e.(s - v) + |e|^2, the lowest stencil index among equals
__device__ inline int mask_best_portal(uint32_t cand, int vx, int vy, int vz, int sx, int sy, int sz) {
  const int wx = sx - vx, wy = sy - vy, wz = sz - vz;
  int best = -1, bd = INT_MAX, bit = 0;
#define FIESTA_BEST(DX, DY, DZ)                                                                           \
  {                                                                                                       \
    const int f = 2 * ((DX) * wx + (DY) * wy + (DZ) * wz) + ((DX) * (DX) + (DY) * (DY) + (DZ) * (DZ));    \
    const bool take = ((cand >> bit) & 1u) && f < bd;                                                     \
    bd = take ? f : bd, best = take ? bit : best;                                                         \
    ++bit;                                                                                                \
  }
  FIESTA_STENCIL24(FIESTA_BEST)
#undef FIESTA_BEST
  return best;
}

// The portals of every HIDDEN site (one with a never-observed voxel among its 26 neighbours inside the grid), once per update: bit k
// of a site's mask = its k-th stencil neighbour is inside the grid, observed, free, and has no other site within that distance.
// A WAVE per 64 words of the site bitmap; for each site found, the wave together: the 9 x 9 voxel rows around the site (x, y within
// 4) as 9-bit windows of the three bitmaps -- one round of independent loads into LDS --, then lane k < 24 judges direction k
// from those windows with shifts and popcounts (a thread per word with nested probe loops took 0.41 ms: one lane in sixty-four
// working through ~600 dependent loads per site).
__global__ __launch_bounds__(256) void k_portal_sites(MaskArgs a, int64_t nwords) {
  __shared__ uint32_t s_eff[4][81], s_obs[4][81], s_occ[4][81];  // row (dx + 4) * 9 + dy + 4: bits z - 4 .. z + 4 of row (x + dx, y + dy) in bits 0 .. 8
  const Geom &g = a.g;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t *we = s_eff[wave], *wo = s_obs[wave], *wc = s_occ[wave];
  const int64_t nchunks = (nwords + 63) / 64;
  for (int64_t ch = blockIdx.x * 4ll + wave; ch < nchunks; ch += (int64_t)gridDim.x * 4) {
    const int64_t wi = ch * 64 + lane;
    const uint32_t mine = wi < nwords ? a.effocc[wi] : 0u;
    unsigned long long todo = __ballot(mine != 0u);
    while (todo) {  // (wave-uniform: the lanes' words one after the other)
      const int src = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      uint32_t m = (uint32_t)__shfl((int)mine, src);
      const int64_t w0 = ch * 64 + src, row = w0 / g.nzw;
      const int zw = (int)(w0 - row * g.nzw), y = (int)(row % g.ny), x = (int)(row / g.ny);
      while (m) {
        const int z = 32 * zw + __ffs((int)m) - 1;
        m &= m - 1;
        // the 81 rows' windows: bit j of a window = voxel z - 4 + j (zero outside the grid; obs: ONE outside the grid for the
        // hidden test below is handled there)
        for (int r = lane; r < 81; r += 64) {
          const int X = x + r / 9 - 4, Y = y + r % 9 - 4;
          uint32_t be = 0, bo = 0, bc = 0;
          if ((unsigned)X < (unsigned)g.nx && (unsigned)Y < (unsigned)g.ny) {
            const int64_t rb = ((int64_t)X * g.ny + Y) * g.nzw;
            const int z0 = z - 4;  // window start (may be negative)
            const int wlo = (z0 < 0 ? 0 : z0) >> 5, whi = min(z + 4, g.nz - 1) >> 5;
            unsigned long long e2 = a.effocc[rb + wlo], o2 = a.obsbits[rb + wlo], c2 = a.occbits[rb + wlo];
            if (whi != wlo) {
              e2 |= (unsigned long long)a.effocc[rb + whi] << 32, o2 |= (unsigned long long)a.obsbits[rb + whi] << 32;
              c2 |= (unsigned long long)a.occbits[rb + whi] << 32;
            }
            const int sh = z0 - 32 * wlo;  // bit of the pair that is window bit 0 (negative: the window starts before the grid)
            if (sh >= 0) be = (uint32_t)(e2 >> sh) & 511u, bo = (uint32_t)(o2 >> sh) & 511u, bc = (uint32_t)(c2 >> sh) & 511u;
            else be = (uint32_t)(e2 << -sh) & 511u, bo = (uint32_t)(o2 << -sh) & 511u, bc = (uint32_t)(c2 << -sh) & 511u;
            // (bits beyond the grid's last voxel are zero in the bitmaps: rows are padded with zeros)
          }
          we[r] = be, wo[r] = bo, wc[r] = bc;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // hidden: a never-observed voxel among the 26 neighbours inside the grid (lane = neighbour)
        bool unobs = false;
        if (lane < 27) {
          const int dx = lane / 9 - 1, dy = (lane / 3) % 3 - 1, dz = lane % 3 - 1;
          unobs = g.in_grid(x + dx, y + dy, z + dz) && !((wo[(dx + 4) * 9 + dy + 4] >> (dz + 4)) & 1u);
        }
        const bool hidden = __any((int)unobs);
        uint32_t mask = 0;
        if (hidden) {
          bool good = false;
          if (lane < 24) {
            int ex = 0, ey = 0, ez = 0, k = 0;
#define FIESTA_PDIR(DX, DY, DZ) \
  if (k++ == lane) ex = (DX), ey = (DY), ez = (DZ);
            FIESTA_STENCIL24(FIESTA_PDIR)
#undef FIESTA_PDIR
            const int ux = x + ex, uy = y + ey, uz = z + ez;
            const int r0 = (ex + 4) * 9 + ey + 4;
            if (g.in_grid(ux, uy, uz) && ((wo[r0] >> (ez + 4)) & 1u) && !((wc[r0] >> (ez + 4)) & 1u)) {
              const int R2 = ex * ex + ey * ey + ez * ez;
              int sites = 0;
              for (int ddx = -2; ddx <= 2; ++ddx)
                for (int ddy = -2; ddy <= 2; ++ddy) {
                  const int rem = R2 - ddx * ddx - ddy * ddy;
                  if (rem < 0) continue;
                  const int dzm = rem >= 4 ? 2 : (rem >= 1 ? 1 : 0);  // |ddz| <= dzm
                  const uint32_t win = we[(ex + ddx + 4) * 9 + ey + ddy + 4] >> (ez + 4 - dzm);
                  uint32_t bits = win & ((1u << (2 * dzm + 1)) - 1u);
                  if (ddx == 0 && ddy == 0) bits &= ~(1u << dzm);  // (the portal voxel itself does not count)
                  sites += __popc(bits);
                }
              good = sites == 1;
            }
          }
          mask = (uint32_t)__ballot(good) & 0xFFFFFFu;
        }
        if (mask && lane == 0) {
          const vox_t ws = pack_coc(x + g.gx0, y + g.gy0, z + g.gz0);
          for (uint32_t slot = portal_hash(ws) & a.ptab_mask;; slot = (slot + 1u) & a.ptab_mask) {
            const uint32_t old = atomicCAS(&a.ptab[slot].x, 0xFFFFFFFFu, ws);
            if (old == 0xFFFFFFFFu || old == ws) {
              a.ptab[slot].y = mask;
              break;
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the next site's windows overwrite these)
      }
    }
  }
}

// Chebyshev distance (in cells, capped at 3) from every fully observed cell to the nearest cell that is not, and the 3^3
// neighbourhood as a bit mask; cells outside the grid do not count (nothing to cross there).
__global__ __launch_bounds__(256) void k_cell_dist(int ncx, int ncy, int ncz, const uint8_t *cellobs, uint8_t *celldist, uint32_t *cellnb,
                                                   unsigned long long *cellst) {
  const int64_t n = (int64_t)ncx * ncy * ncz;
  for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < n; c += (int64_t)gridDim.x * blockDim.x) {
    const int cz = (int)(c % ncz), cy = (int)((c / ncz) % ncy), cx = (int)(c / ((int64_t)ncz * ncy));
    int d = 3;
    uint32_t nb = 0x7FFFFFFu;
    unsigned long long st = 0x15555555555555ull;  // (27 x 01)
    for (int dx = -2; dx <= 2; ++dx)
      for (int dy = -2; dy <= 2; ++dy)
        for (int dz = -2; dz <= 2; ++dz) {
          const int ux = cx + dx, uy = cy + dy, uz = cz + dz;
          if ((unsigned)ux >= (unsigned)ncx || (unsigned)uy >= (unsigned)ncy || (unsigned)uz >= (unsigned)ncz) continue;
          const uint32_t co = cellobs[((int64_t)ux * ncy + uy) * ncz + uz];
          if (co != 1u) {
            const int r = max(max(dx < 0 ? -dx : dx, dy < 0 ? -dy : dy), dz < 0 ? -dz : dz);
            d = min(d, r);
            if (r <= 1) {
              const int b = (dx + 1) * 9 + (dy + 1) * 3 + (dz + 1);
              nb &= ~(1u << b);
              st = (st & ~(3ull << (2 * b))) | ((unsigned long long)co << (2 * b));
            }
          }
        }
    celldist[c] = (uint8_t)d;
    cellnb[c] = nb;
    cellst[c] = st;
  }
}

#ifndef FIESTA_CLASSIFY_SLABS
#define FIESTA_CLASSIFY_SLABS 2
#endif
constexpr int kClassifySlabs = FIESTA_CLASSIFY_SLABS;  // voxel slabs (x) of a quad whose loads are in flight together
constexpr int kMaskQueue = 512;  // walks a wave collects before it takes a range of a segment

// One WAVE per quad (four cells along z: 32 voxels = one 128-byte line per voxel row, one bitmap word per row); lane = (y, four
// consecutive z); waves are persistent.  Voxels that need the segment walk go to the walk list, through an LDS queue per wave
// (one atomic on a segment's cursor per flush: returning atomics on one address serialise at tens of ns each).  No load depends
// on a voxel's word: the cells between a voxel and its winner are judged from the cell's own summaries, held in registers.
#ifndef FIESTA_CLASSIFY_WAVES
#define FIESTA_CLASSIFY_WAVES 5  /* (96 VGPRs without a spill; 6 spills 14) */
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FIESTA_CLASSIFY_WAVES, FIESTA_CLASSIFY_WAVES))) void k_mask_classify(MaskArgs a) {
  __shared__ uint2 s_queue[4][kMaskQueue];
  __shared__ uint32_t s_qn[4], s_need[27];
  const Geom &g = a.g;
  if (a.failed && *a.failed) return;  // T was never written (a cell without a list): the host takes the envelope passes
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint2 *queue = s_queue[wave];
  const int nqz = g.nzw;  // quads along z
  const int64_t nquads = (int64_t)a.ncx * a.ncy * nqz;
  const int64_t nwaves = (int64_t)gridDim.x * 4, gw = blockIdx.x * 4ll + wave;
  const int64_t per = (nquads + nwaves - 1) / nwaves;
  const int64_t q0 = gw * per, q1 = min(q0 + per, nquads);
  const int y = lane >> 3, z4 = lane & 7;
  const bool vec = (g.nz & 3) == 0;
  if (lane == 0) s_qn[wave] = 0;
  unsigned nwalk = 0;
  // the cells of the 3^3 around a voxel's cell that the box of the cell and a winner's cell (one cell away) covers, by the offset
  if (threadIdx.x < 27) {
    const uint32_t mx = 2u | (1u << (threadIdx.x / 9)), my = 2u | (1u << ((threadIdx.x / 3) % 3)), mz = 2u | (1u << (threadIdx.x % 3));
    const uint32_t pm = ((my & 1u) ? mz : 0u) | ((my & 2u) ? mz << 3 : 0u) | ((my & 4u) ? mz << 6 : 0u);
    s_need[threadIdx.x] = ((mx & 1u) ? pm : 0u) | ((mx & 2u) ? pm << 9 : 0u) | ((mx & 4u) ? pm << 18 : 0u);
  }
  __syncthreads();

  // (a quad's walks go to segment quad % kMaskSegs: neighbouring quads feed different segments, the segments fill evenly)
  auto flush = [&](const int seg) {  // the wave's queued walks -> a segment of the list (every lane calls)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const uint32_t n = s_qn[wave];
    if (n) {
      uint32_t base = 0;
      if (lane == 0) {
        base = (uint32_t)atomicAdd(&a.ctr[MC_SEG0 + 16 * seg], (unsigned long long)n);
        if (base + n > a.seg_cap) atomicAdd(&a.ctr[MC_OVERFLOW], 1ull);
      }
      base = (uint32_t)__shfl((int)base, 0);
      for (uint32_t i = (uint32_t)lane; i < n; i += 64u)
        if (base + i < a.seg_cap) a.walks[(size_t)seg * a.seg_cap + base + i] = queue[i];
      nwalk += lane == 0 ? n : 0u;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane == 0) s_qn[wave] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  };

  for (int64_t q = q0; q < q1; ++q) {
    const int qz = (int)(q % nqz);
    const int cy = (int)((q / nqz) % a.ncy), cx = (int)(q / ((int64_t)nqz * a.ncy));
    uint32_t cd[4], cn[4];
    bool any_obs = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int cz = 4 * qz + k;
      const int64_t c = ((int64_t)cx * a.ncy + cy) * a.ncz + cz;
      any_obs |= cz < a.ncz && a.cellobs[c] != 0u;
      cd[k] = cz < a.ncz ? (uint32_t)a.celldist[c] : 0u;
      cn[k] = cz < a.ncz ? a.cellnb[c] : 0u;
    }
    const int Y = 8 * cy + y, Z = 32 * qz + 4 * z4;
    const bool row_in = Y < g.ny && Z < g.nz;
    const uint32_t mycd = cd[z4 >> 1], mycn = cn[z4 >> 1];  // (the cell of this lane's four voxels)
    // the quad's eight slabs: every load of the quad in flight before the first word is looked at (a slab at a time, each wave
    // went through eight dependent round trips to memory per quad)
    for (int xh = 0; xh < 8; xh += kClassifySlabs) {
    uint32_t ww[kClassifySlabs][4], obx[kClassifySlabs], ocx[kClassifySlabs];
    if (any_obs) {
#pragma unroll
      for (int x = 0; x < kClassifySlabs; ++x) {
        const int X = 8 * cx + xh + x;
        const int64_t base = ((int64_t)X * g.ny + Y) * g.nz + Z;
        ww[x][0] = ww[x][1] = ww[x][2] = ww[x][3] = kUnobserved;
        obx[x] = ocx[x] = 0;
        if (row_in && X < g.nx) {
          if (vec) {
            const uint4 v = *reinterpret_cast<const uint4 *>(a.out + base);
            ww[x][0] = v.x, ww[x][1] = v.y, ww[x][2] = v.z, ww[x][3] = v.w;
          } else {
            for (int k = 0; k < 4 && Z + k < g.nz; ++k) ww[x][k] = a.out[base + k];
          }
          const int64_t bw = ((int64_t)X * g.ny + Y) * g.nzw + qz;
          obx[x] = a.obsbits[bw], ocx[x] = a.occbits[bw];
        }
      }
    }
#pragma unroll
    for (int x = 0; x < kClassifySlabs; ++x) {
      const int X = 8 * cx + xh + x;
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
      uint32_t w[4] = {ww[x][0], ww[x][1], ww[x][2], ww[x][3]}, w0[4];
      const uint32_t ob = (obx[x] >> (4 * z4)) & 15u, oc = (ocx[x] >> (4 * z4)) & 15u;
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
          // (the masked transform never runs on a map whose ids wrap: the word's fields ARE the winner's global coordinates)
          const int sx = (int)((w[k] >> 20) & 1023u) - g.gx0, sy = (int)((w[k] >> 10) & 1023u) - g.gy0, sz = (int)(w[k] & 1023u) - g.gz0;
          const int cx1 = sx >> 3, cy1 = sy >> 3, cz1 = sz >> 3, cz0 = vz >> 3;
          const int co = max(max(cx1 > cx ? cx1 - cx : cx - cx1, cy1 > cy ? cy1 - cy : cy - cy1), cz1 > cz0 ? cz1 - cz0 : cz0 - cz1);
          // every cell the box of v and its winner touches fully observed -> certified without a walk: from the cell's distance
          // to the nearest cell that is not, else (a box within the 3^3 cells around) from the cell's neighbour mask
          bool full = (uint32_t)co < mycd;
          if (!full && co == 1) {
            const uint32_t need = s_need[(cx1 - cx + 1) * 9 + (cy1 - cy + 1) * 3 + (cz1 - cz0 + 1)];
            full = (mycn & need) == need;
          }
          want[k] = !full;
        }
      }
      if (row_in && (w[0] != w0[0] || w[1] != w0[1] || w[2] != w0[2] || w[3] != w0[3])) {
        if (vec) *reinterpret_cast<uint4 *>(a.out + base) = uint4{w[0], w[1], w[2], w[3]};
        else
          for (int k = 0; k < 4 && Z + k < g.nz; ++k) a.out[base + k] = w[k];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned long long mq = __ballot(want[k]);
        if (mq) {
          const uint32_t at = s_qn[wave];
          if (want[k]) queue[at + (uint32_t)__popcll(mq & ((1ull << lane) - 1ull))] = uint2{((uint32_t)X << 20) | ((uint32_t)Y << 10) | (uint32_t)(Z + k), w[k]};
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          if (lane == 0) s_qn[wave] = at + (uint32_t)__popcll(mq);
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
      }
      if (s_qn[wave] > (uint32_t)(kMaskQueue - 256)) flush((int)(q % kMaskSegs));
    }
    }
    flush((int)(q % kMaskSegs));
  }
  if (lane == 0 && nwalk) atomicAdd(&a.ctr[MC_WALKS], (unsigned long long)nwalk);
}

// The certificate proper: one lane per queued voxel, three kinds of batches of 256 per work-group.  STRAIGHT: the segment judged from
// the cells it crosses (mask_segment_cells); the voxels that need the samples -- a partly observed cell on the way, a step through
// a cell's corner -- wait in an LDS list and are walked sample by sample 256 at a time (SAMPLES), so that the long walk never
// runs for one lane of a wave.  The voxels the straight segment refuses wait in an LDS queue, and the portal certificate runs in
// ROUNDS over that queue: a round = every lane takes a waiting voxel and tries ONE portal, the nearest it has not tried; a voxel
// that fails and has candidates left goes back into the queue.  (One lane trying its up to 24 portals in a row: a wave took as
// long as its unluckiest lane -- a quarter of the waiting voxels end up uncertified, after nine walks on average -- while three
// quarters are done after one.)
constexpr int kWalkQueue = 768, kWalkDeferred = 768;
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_mask_walk(MaskArgs a) {
  __shared__ uint32_t s_marked, s_qn, s_dn;
  __shared__ uint2 s_q[kWalkQueue];
  __shared__ uint32_t s_qm[kWalkQueue];     // candidates left (0xFFFFFFFF: not looked up yet)
  __shared__ uint2 s_d[kWalkDeferred];      // voxels put aside for a long walk
  __shared__ uint32_t s_dm[kWalkDeferred];  // 0xFFFFFFFF: the samples of its segment; else the diagonals of the path to portal (code >> 24), candidates left in the low 24 bits
  const Geom &g = a.g;
  if (a.failed && *a.failed) return;
  if (a.ctr[MC_OVERFLOW]) return;
  if (threadIdx.x == 0) s_marked = 0, s_qn = 0, s_dn = 0;
  unsigned marked = 0;
  // work-group b takes segment b % kMaskSegs, its share of it
  const int seg = (int)(blockIdx.x % kMaskSegs);
  const uint32_t part = blockIdx.x / kMaskSegs, parts = (gridDim.x + kMaskSegs - 1 - seg) / kMaskSegs;
  const uint32_t n = (uint32_t)min(a.ctr[MC_SEG0 + 16 * seg], (unsigned long long)a.seg_cap);
  uint32_t i0 = part * blockDim.x;
  for (;;) {
    __syncthreads();
    const uint32_t qn = s_qn, dn = s_dn;
    __syncthreads();  // (both counts are read by everybody before anybody adds to them)
    // (room: the queue takes at most 256 more where it is below 512, the list where it is below 512 -- both hold 768)
    if (qn < 512u && (dn >= 256u || (dn > 0u && i0 >= n && qn < 256u))) {  // (uniform) LONG WALKS: up to 256 voxels put aside
      const uint32_t take = min(dn, 256u);
      uint2 e{0u, 0u};
      uint32_t code = 0;
      if (threadIdx.x < take) e = s_d[dn - take + threadIdx.x], code = s_dm[dn - take + threadIdx.x];
      __syncthreads();
      if (threadIdx.x == 0) s_dn = dn - take;
      if (threadIdx.x < take) {
        const int vx = (int)(e.x >> 20), vy = (int)((e.x >> 10) & 1023u), vz = (int)(e.x & 1023u);
        const int sx = (int)((e.y >> 20) & 1023u) - g.gx0, sy = (int)((e.y >> 10) & 1023u) - g.gy0, sz = (int)(e.y & 1023u) - g.gz0;
        if (code == 0xFFFFFFFFu) {  // the samples of the straight segment
          if (!mask_segment_samples(a, cell_view(a, vx, vy, vz), vx, vy, vz, sx, sy, sz)) {
            const uint32_t at = atomicAdd(&s_qn, 1u);
            s_q[at] = e, s_qm[at] = 0xFFFFFFFFu;
          }
        } else {  // the diagonals of a path that holds the winner on every sample
          int dx, dy, dz;
          stencil_dir((int)(code >> 24), dx, dy, dz);
          if (!mask_path_diagonals(a, vx, vy, vz, sx + dx, sy + dy, sz + dz, e.y & ~kAct)) {
            const uint32_t cand = code & 0xFFFFFFu;
            if (cand) {
              const uint32_t at = atomicAdd(&s_qn, 1u);
              s_q[at] = e, s_qm[at] = cand;
            } else {
              atomicOr(&a.ubits[g.bitword(vx, vy, vz)], 1u << (vz & 31));
              ++marked;
            }
          }
        }
      }
      continue;
    }
    if (qn < 256u && dn < 256u && i0 < n) {  // (uniform) STRAIGHT: another 256 voxels of the list
      const uint32_t i = i0 + threadIdx.x;
      i0 += parts * blockDim.x;
      if (i < n) {
        const uint2 e = a.walks[(size_t)seg * a.seg_cap + i];
        const int vx = (int)(e.x >> 20), vy = (int)((e.x >> 10) & 1023u), vz = (int)(e.x & 1023u);
        const int sx = (int)((e.y >> 20) & 1023u) - g.gx0, sy = (int)((e.y >> 10) & 1023u) - g.gy0, sz = (int)(e.y & 1023u) - g.gz0;
        const int r = mask_segment_cells(a, cell_view(a, vx, vy, vz), vx, vy, vz, sx, sy, sz);  // 0 refused, 1 observed, 2 ask the samples
        if (r == 0) {
          const uint32_t at = atomicAdd(&s_qn, 1u);
          s_q[at] = e, s_qm[at] = 0xFFFFFFFFu;
        } else if (r == 2) {
          const uint32_t at = atomicAdd(&s_dn, 1u);
          s_d[at] = e, s_dm[at] = 0xFFFFFFFFu;
        }
      }
      continue;
    }
    if (qn == 0u) break;  // (uniform: the list is exhausted and nothing is put aside)
    const uint32_t take = min(qn, 256u);  // a ROUND of the portal certificate
    const bool aside = dn < 512u;         // (uniform) room to put the paths with diagonals aside
    uint2 e{0u, 0u};
    uint32_t cand = 0;
    if (threadIdx.x < take) e = s_q[qn - take + threadIdx.x], cand = s_qm[qn - take + threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) s_qn = qn - take;
    __syncthreads();
    if (threadIdx.x < take) {
      const int vx = (int)(e.x >> 20), vy = (int)((e.x >> 10) & 1023u), vz = (int)(e.x & 1023u);
      const int sx = (int)((e.y >> 20) & 1023u) - g.gx0, sy = (int)((e.y >> 10) & 1023u) - g.gy0, sz = (int)(e.y & 1023u) - g.gz0;
      const vox_t ws = e.y & ~kAct;
      if (cand == 0xFFFFFFFFu) {
        cand = mask_portals_of(a, vx, vy, vz, sx, sy, sz, ws);
        PROBE_ADD(6, 1);
        PROBE_ADD(7, __popc(cand));
        if (!cand) PROBE_ADD(8, 1);
      }
      bool certified = false;
      if (cand) {
        const int bit = mask_best_portal(cand, vx, vy, vz, sx, sy, sz);
        cand &= ~(1u << bit);
        int dx, dy, dz;
        stencil_dir(bit, dx, dy, dz);
        const int r = mask_path_in_cell(a, vx, vy, vz, sx + dx, sy + dy, sz + dz, ws);
        certified = r == 1;
        if (r == 2) {  // every sample holds the winner: its (1, 1, 1)-diagonals in a batch of their own
          if (aside) {
            const uint32_t at = atomicAdd(&s_dn, 1u);
            s_d[at] = e, s_dm[at] = ((uint32_t)bit << 24) | cand;
            certified = true;  // (nothing more to do for it in this round)
          } else {
            certified = mask_path_diagonals(a, vx, vy, vz, sx + dx, sy + dy, sz + dz, ws);
          }
        }
      }
      if (certified) PROBE_ADD(9, 1);
      if (!certified) {
        if (cand) {  // back into the queue
          const uint32_t at = atomicAdd(&s_qn, 1u);
          s_q[at] = e, s_qm[at] = cand;
        } else {  // uncertified: marked for repair (k_mask_cells gives it the word it starts from)
          atomicOr(&a.ubits[g.bitword(vx, vy, vz)], 1u << (vz & 31));
          ++marked;
        }
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) marked += (unsigned)__shfl_xor((int)marked, off);
  if ((threadIdx.x & 63) == 0 && marked) atomicAdd(&s_marked, marked);
  __syncthreads();
  if (threadIdx.x == 0 && s_marked) atomicAdd(&a.ctr[MC_MARKED], (unsigned long long)s_marked);
}

// The CELLS (8^3 voxels) that hold a marked voxel -> the repair list; their marked voxels' words also go into the second buffer
// (the repair keeps the two buffers equal on marked voxels between iterations).  One wave per quad (four cells along z): lane =
// voxel row (x = lane / 8, y), one word of marks per row, a byte of it per cell.
constexpr int kCellsList = 2048;
__global__ __launch_bounds__(256) void k_mask_cells(MaskArgs a) {
  __shared__ uint32_t s_n, s_base, s_list[kCellsList];  // the cells this work-group finds: ONE atomic on the list's cursor at its end
  __shared__ uint16_t s_vox[4][2048];  // a wave's marked voxels of its quad: row << 5 | z
  const Geom &g = a.g;
  if (a.failed && *a.failed) return;
  if (a.ctr[MC_OVERFLOW]) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint16_t *vox = s_vox[wave];
  const int nqz = g.nzw;
  const int64_t nquads = (int64_t)a.ncx * a.ncy * nqz;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  for (int64_t q = blockIdx.x * 4ll + wave; q < nquads; q += (int64_t)gridDim.x * 4) {
    {
      const int qz = (int)(q % nqz);
      const int cy = (int)((q / nqz) % a.ncy), cx = (int)(q / ((int64_t)nqz * a.ncy));
      const int X = 8 * cx + (lane >> 3), Y = 8 * cy + (lane & 7);
      uint32_t m = (X < g.nx && Y < g.ny) ? a.ubits[((int64_t)X * g.ny + Y) * g.nzw + qz] : 0u;
      if (__any((int)(m != 0u))) {  // (wave-uniform)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const bool has = __any((int)(((m >> (8 * k)) & 255u) != 0u));
          if (has && lane == 0) {
            const uint32_t cell = (uint32_t)((((int64_t)cx * a.ncy + cy) * a.ncz) + 4 * qz + k);
            const uint32_t at = atomicAdd(&s_n, 1u);
            if (at < (uint32_t)kCellsList) s_list[at] = cell;
            else a.uq[atomicAdd(&a.ctr[MC_QUADS], 1ull)] = cell;  // (a map far larger than the launch was sized for)
          }
        }
        // the quad's marked voxels, spread over the lanes: a row's marks one after the other -- each a load of the old word, its
        // obstacle looked up behind it -- kept a wave as long as its fullest row
        uint32_t incl = (uint32_t)__popc(m);
        for (int off = 1; off < 64; off <<= 1) {
          const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
          if (lane >= off) incl += up;
        }
        const uint32_t nm = (uint32_t)__shfl((int)incl, 63);
        {
          uint32_t at = incl - (uint32_t)__popc(m);
          while (m) {
            const int b = __ffs((int)m) - 1;
            m &= m - 1;
            vox[at++] = (uint16_t)((lane << 5) | b);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (uint32_t i = (uint32_t)lane; i < nm; i += 64u) {
          const uint32_t code = vox[i];
          const int row = (int)(code >> 5), b = (int)(code & 31u);
          const int VX = 8 * cx + (row >> 3), VY = 8 * cy + (row & 7), VZ = 32 * qz + b;
          const int64_t idx = ((int64_t)VX * g.ny + VY) * g.nz + VZ;
          // an uncertified voxel keeps what it held before the update if that obstacle still exists, else "no obstacle" -- the
          // word the repair starts from, in both buffers
          vox_t o = a.old[idx] & ~kAct;
          if (!(o & kNoCoc)) {
            int ox, oy, oz;
            unpack_coc(g.wrap, VX + g.gx0, VY + g.gy0, VZ + g.gz0, o, ox, oy, oz);
            ox -= g.gx0, oy -= g.gy0, oz -= g.gz0;
            if (!(g.in_grid(ox, oy, oz) && bit_test(a.occbits, g, ox, oy, oz))) o = kInf;
          } else {
            o = kInf;
          }
          a.out[idx] = o;
          a.old[idx] = o;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
    }
  }
  __syncthreads();
  const uint32_t n = min(s_n, (uint32_t)kCellsList);
  if (threadIdx.x == 0 && n) s_base = (uint32_t)atomicAdd(&a.ctr[MC_QUADS], (unsigned long long)n);
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) a.uq[s_base + i] = s_list[i];
}

// ---- the repair: block-Jacobi pulls on the marked voxels ------------------------------------------------------------------------------
// Global iteration `it` of a chain.  k_repair_cell: ONE WAVE per listed cell that changed, or has a neighbour that changed, in the
// iteration before (all of them in an update's first) -- no barrier anywhere, a CU keeps sixteen cells in flight: the cell + its
// 2-voxel halo (12^3 words) from `out` into the wave's LDS tile, its marked voxels compacted into a list (a lane per marked voxel),
// up to kMaskSub Jacobi steps on them against that halo, the voxels that end different into `old` (the second buffer).
// k_repair_commit: the same cells' differences old -> out, counted, and the cells around each changed voxel stamped for the next
// iteration.  Nobody writes `out` while a k_repair_cell launch reads it: the result does not depend on the order the waves run in.
constexpr int kTileE = 12, kTileN = kTileE * kTileE * kTileE;
__global__ __launch_bounds__(256) void k_repair_cell(MaskArgs a, int it, int rd, uint32_t tag_prev, int first) {
  __shared__ vox_t s_tile[4][kTileN];
  __shared__ uint16_t s_voxl[4][512];  // marked voxels of the cell: x << 6 | y << 3 | z
  __shared__ vox_t s_newv[4][512];
  const Geom &g = a.g;
  if (a.failed && *a.failed) return;
  if (a.ctr[MC_OVERFLOW]) return;
  if (it > 0 && a.ctr[MC_CHANGED0 + it - 1] == 0) return;
  const uint32_t nuq = (uint32_t)a.ctr[MC_QUADS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  vox_t *tile = s_tile[wave];
  uint16_t *voxl = s_voxl[wave];
  vox_t *newv = s_newv[wave];
  const uint32_t *stamp_r = a.qstamp[rd];  // (written by the commit of the iteration before)
  for (uint32_t ci = blockIdx.x * 4u + (uint32_t)wave; ci < nuq; ci += gridDim.x * 4u) {
    const uint32_t c = a.uq[ci];
    if (!first && stamp_r[c] != tag_prev) continue;  // (wave-uniform)
    const int cz = (int)(c % (uint32_t)a.ncz);
    const int cy = (int)((c / (uint32_t)a.ncz) % (uint32_t)a.ncy), cx = (int)(c / ((uint32_t)a.ncz * (uint32_t)a.ncy));
    const int X0 = 8 * cx - 2, Y0 = 8 * cy - 2, Z0 = 8 * cz - 2;
    // (LDS is in order within a wave: no barrier between the phases below)
    for (int i = lane; i < kTileN; i += 64) {
      const int zz = i % kTileE, r = i / kTileE, yy = r % kTileE, xx = r / kTileE;
      const int X = X0 + xx, Y = Y0 + yy, Z = Z0 + zz;
      tile[i] = g.in_grid(X, Y, Z) ? (a.out[g.idx(X, Y, Z)] & ~kAct) : kUnobserved;
    }
    // the cell's marks: lane = voxel row (x = lane / 8, y = lane % 8), a byte of the row's word
    const int X = 8 * cx + (lane >> 3), Y = 8 * cy + (lane & 7);
    uint32_t m = (X < g.nx && Y < g.ny) ? ((a.ubits[((int64_t)X * g.ny + Y) * g.nzw + (cz >> 2)] >> (8 * (cz & 3))) & 255u) : 0u;
    uint32_t incl = (uint32_t)__popc(m);
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
      if (lane >= off) incl += up;
    }
    const uint32_t nm = (uint32_t)__shfl((int)incl, 63);
    {
      uint32_t at = incl - (uint32_t)__popc(m);
      while (m) {
        const int b = __ffs((int)m) - 1;
        m &= m - 1;
        voxl[at++] = (uint16_t)((lane << 3) | b);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    uint32_t ever = 0;  // bit k: the lane's k-th voxel (lane + 64 k of the list) has changed -- a pull only ever lowers a distance
    for (int sub = 0; sub < kMaskSub; ++sub) {
      bool changed = false;
      int k = 0;
      for (uint32_t i = (uint32_t)lane; i < nm; i += 64u, ++k) {
        const uint32_t code = voxl[i];
        const int x = (int)(code >> 6), y = (int)((code >> 3) & 7u), z = (int)(code & 7u);
        const int vx = 8 * cx + x + g.gx0, vy = 8 * cy + y + g.gy0, vz = 8 * cz + z + g.gz0;
        const int c0 = ((x + 2) * kTileE + y + 2) * kTileE + z + 2;
        const vox_t cw = tile[c0];
        int32_t best = (cw & kNoCoc) ? kD2Inf : dist2(0, vx, vy, vz, cw);  // (the masked transform never runs on a map whose ids wrap)
        vox_t bw = cw;
        // (a neighbour's word names the same obstacle for this voxel: ids are global coordinates, modulo 1024 on larger grids)
#define FIESTA_PULL(DX, DY, DZ)                                                          \
  {                                                                                      \
    const vox_t w = tile[c0 + ((DX) * kTileE + (DY)) * kTileE + (DZ)];                   \
    if (!(w & kNoCoc)) {                                                                 \
      const int32_t d = dist2(0, vx, vy, vz, w);                                         \
      if (d < best) best = d, bw = w;                                                    \
    }                                                                                    \
  }
        FIESTA_STENCIL24(FIESTA_PULL)
#undef FIESTA_PULL
        newv[i] = bw;
        changed |= bw != cw;
        ever |= (uint32_t)(bw != cw) << k;
      }
      if (!__any((int)changed)) break;  // (wave-uniform)
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (every pull of this step has read the tile)
      for (uint32_t i = (uint32_t)lane; i < nm; i += 64u) {
        const uint32_t code = voxl[i];
        tile[(((int)(code >> 6) + 2) * kTileE + (int)((code >> 3) & 7u) + 2) * kTileE + (int)(code & 7u) + 2] = newv[i];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    // the voxels that differ from what `out` holds -> the second buffer (the two were equal on every marked voxel)
    for (uint32_t i = (uint32_t)lane; ever; i += 64u, ever >>= 1) {
      if (!(ever & 1u)) continue;
      const uint32_t code = voxl[i];
      const int x = (int)(code >> 6), y = (int)((code >> 3) & 7u), z = (int)(code & 7u);
      a.old[g.idx(8 * cx + x, 8 * cy + y, 8 * cz + z)] = tile[((x + 2) * kTileE + y + 2) * kTileE + z + 2];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the next cell's staging overwrites the tile)
  }
}

__global__ __launch_bounds__(256) void k_repair_commit(MaskArgs a, int it, int rd, uint32_t tag_prev, uint32_t tag, int first) {
  __shared__ uint16_t s_voxl[4][512];  // marked voxels of the wave's cell: x << 6 | y << 3 | z
  const Geom &g = a.g;
  if (a.failed && *a.failed) return;
  if (a.ctr[MC_OVERFLOW]) return;
  if (it > 0 && a.ctr[MC_CHANGED0 + it - 1] == 0) return;
  const uint32_t nuq = (uint32_t)a.ctr[MC_QUADS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;  // one wave per cell
  uint16_t *voxl = s_voxl[wave];
  const uint32_t *stamp_r = a.qstamp[rd];
  uint32_t *stamp_w = a.qstamp[rd ^ 1];
  unsigned total = 0;
  for (uint32_t ci = blockIdx.x * 4u + (uint32_t)wave; ci < nuq; ci += gridDim.x * 4u) {
    const uint32_t c = a.uq[ci];
    if (!first && stamp_r[c] != tag_prev) continue;  // (the cells k_repair_cell worked on)
    const int cz = (int)(c % (uint32_t)a.ncz);
    const int cy = (int)((c / (uint32_t)a.ncz) % (uint32_t)a.ncy), cx = (int)(c / ((uint32_t)a.ncz * (uint32_t)a.ncy));
    // the cell's marks, spread over the lanes (lane = voxel row for the load: x = lane / 8, y = lane % 8, a byte of the row's word)
    const int X = 8 * cx + (lane >> 3), Y = 8 * cy + (lane & 7);
    uint32_t m = (X < g.nx && Y < g.ny) ? ((a.ubits[((int64_t)X * g.ny + Y) * g.nzw + (cz >> 2)] >> (8 * (cz & 3))) & 255u) : 0u;
    uint32_t incl = (uint32_t)__popc(m);
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
      if (lane >= off) incl += up;
    }
    const uint32_t nm = (uint32_t)__shfl((int)incl, 63);
    {
      uint32_t at = incl - (uint32_t)__popc(m);
      while (m) {
        const int b = __ffs((int)m) - 1;
        m &= m - 1;
        voxl[at++] = (uint16_t)((lane << 3) | b);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    unsigned mine = 0;
    uint32_t near = 0;  // bit (dx + 1) * 9 + (dy + 1) * 3 + (dz + 1): a neighbour cell has a changed voxel in its halo
    for (uint32_t i = (uint32_t)lane; i < nm; i += 64u) {
      const uint32_t code = voxl[i];
      const int x = (int)(code >> 6), y = (int)((code >> 3) & 7u), b = (int)(code & 7u);
      const int64_t idx = g.idx(8 * cx + x, 8 * cy + y, 8 * cz + b);
      const vox_t nv = a.old[idx];
      if (nv == a.out[idx]) continue;
      a.out[idx] = nv;
      ++mine;
      const int dxl = x < 2 ? -1 : 0, dxh = x > 5 ? 1 : 0, dyl = y < 2 ? -1 : 0, dyh = y > 5 ? 1 : 0, dzl = b < 2 ? -1 : 0, dzh = b > 5 ? 1 : 0;
      for (int dx = dxl; dx <= dxh; ++dx)
        for (int dy = dyl; dy <= dyh; ++dy)
          for (int dz = dzl; dz <= dzh; ++dz) near |= 1u << ((dx + 1) * 9 + (dy + 1) * 3 + (dz + 1));
    }
    for (int off = 32; off > 0; off >>= 1) {
      mine += (unsigned)__shfl_xor((int)mine, off);
      near |= (uint32_t)__shfl_xor((int)near, off);
    }
    if (mine && lane < 27 && ((near >> lane) & 1u)) {
      const int ux = cx + lane / 9 - 1, uy = cy + (lane / 3) % 3 - 1, uz = cz + lane % 3 - 1;
      if ((unsigned)ux < (unsigned)a.ncx && (unsigned)uy < (unsigned)a.ncy && (unsigned)uz < (unsigned)a.ncz)
        stamp_w[((int64_t)ux * a.ncy + uy) * a.ncz + uz] = tag;
    }
    total += mine;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the next cell's list overwrites this one)
  }
  if (lane == 0 && total) atomicAdd(&a.ctr[MC_CHANGED0 + it], (unsigned long long)total);
}

}  // namespace fiesta
