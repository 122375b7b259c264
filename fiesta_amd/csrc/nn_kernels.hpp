// fiesta_amd/csrc/nn_kernels.hpp -- gfx950 kernels of the CELL transform (nn_core.hpp): UpdateESDF on a fully observed map
// with a sparse obstacle set, as three launches that never read a per-voxel array:
//
//   k_nn_cells   occupancy bitmap -> the obstacles ("sites") ordered by cell row, and per cell the index of its first
//                site.  A wave per row of cells (cx, cy): lane c reads byte c of each of the row's 64 voxel rows -- one
//                coalesced 64-byte segment per load.                                            reads 1 bit / voxel
//   k_nn_lists   FOUR LANES per cell: the competitor (nearest site to the cell centre), then every site of the search window
//                that the competitor does not dominate over the whole cell -> the cell's list.  A row of window cells is
//                ONE contiguous range of the site array (that is what the ordering is for).  The work-group first copies
//                its neighbourhood -- the table rows and sites within 3 cells of its 64 x 4 cells -- into LDS: a lane's
//                ~150 dependent reads then cost LDS latency, not L2 latency (the first version, straight from memory,
//                spent 130 us on config 2 waiting).                                              reads sites (L2-resident)
//   k_nn_fill    a WAVE per cell, a lane per (y, z) column of the cell, eight x-slabs in registers: per list entry one
//                v_dot4_i32_i8 + one v_lshl_add_u32 give the lane's key at x = 0, one add per slab moves it along x,
//                v_min3_u32 keeps the best of two entries; the winner's word is fetched from the list by ds_bpermute and
//                stored.  Waves are persistent (32 cells each on config 2) and fetch the NEXT cell's list while they
//                work on this one: a wave per cell spent its life waiting for six dependent loads.   writes 4 B / voxel
//
// A cell without a list (nothing within the window's reach, more candidates than a list holds) fails the whole
// transform: k_nn_lists counts it, k_nn_fill then leaves the field alone and the host runs the envelope passes instead
// (dense_map.hip: run_cells).
#pragma once
#include "common.hpp"
#include "nn_core.hpp"

namespace fiesta {

struct NnArgs {
  nn::Geom g;
  const uint8_t *cellobs;  // nullable; a masked transform (mask_kernels.hpp): per cell 0 = no voxel ever observed -> no list wanted
  const uint32_t *occ;  // the occupancy bitmap the region is cut out of: row (X, Y) at (X * sny + Y) * snzw words, bit Z;
  int sx0, sy0, szb;    // region voxel (x, y, z) is bit z % 8 of byte szb + z / 8 of row (sx0 + x, sy0 + y)
  int sny, snzw;        // (an unsharded map: its own bitmap, offsets 0; a shard: its replica of the global one)
  uint32_t *ctab;       // [ncx * ncy][ncz + 1]: first site of the cell; entry ncz: end of the row
  uint32_t *sites;      // packed x << 20 | y << 10 | z
  uint32_t sites_cap;
  uint32_t *lists;      // [cells][nn::kStride]: the cells' records (nn_core.hpp), dword 0 = entries (0: the cell has no list)
  unsigned long long *cursor;   // sites handed out so far
  unsigned long long *failed;   // cells without a list (+ 1 if the site array overflowed)
  unsigned long long *entries;  // list entries in total (statistics, and what the engine choice learns from)
  unsigned long long *maxd2;    // TRACK: atomicMax of every d^2 written
  vox_t *coc;
  // k_nn_close (one thread, behind the fill): copies the results the host wants into pinned host memory `pub` (indexed like
  // the counter array), `tag` last; leaves the transform's counters -- and, if no cell failed, the two drained queue lengths
  // `queues` -- at zero for the next update; hands the distance bound to `track_dst`.
  uint32_t *dump;  // 128 dwords nobody reads: lanes outside the array store here, so that every wave issues the same stores
  unsigned long long *pub, *queues, *track_dst;
  unsigned long long tag;
  int pub_failed, pub_entries, pub_maxd2, pub_tag, pub_dirty, pub_brute;  // slots of pub
  // an INCREMENTAL transform (k_nn_mark, k_nn_lists_dirty, k_nn_fill_dirty): the lists of the last transform are still valid except
  // in the cells whose search window holds a voxel that changed occupancy since -- only those get a new list and a new fill
  const uint32_t *chg[2];       // the insert and the delete queue: linear voxel indices of the array
  uint32_t nchg[2];
  uint32_t *dirty_flag;         // per cell: on the dirty list
  uint32_t *dirty_list;         // cells to redo (linear cell index)
  unsigned long long *dirty_count;
  uint32_t dirty_cap;           // (more dirty cells than this: the transform fails and the full one runs)
  // cells WITHOUT a list (nothing within reach, more survivors than a list holds): up to fail_cap of them are served one by one
  // (k_nn_brute, behind the fill); more than that -- or any on a shard with an open face -- fails the transform (`failed`)
  uint32_t *fail_list;          // linear cell index
  unsigned long long *nfail;
  uint32_t fail_cap;
  unsigned long long *ticket;   // k_nn_close: work-groups that have finished (zero between launches)
};


// ---- sites by cell row ---------------------------------------------------------------------------------------------------
// One batch of loads per wave: the 64 voxel rows of the cell row, a byte per lane and row (two on a 1024-voxel axis), kept
// in registers for both halves -- counting, and after the wave's range of the site array is known, writing the sites.
// ROWS (the region's first byte of a row dword-aligned: any unsharded map; a shard's region starts on whole bitmap words along
// z): the same bytes arrive as FOUR 16-byte loads per lane -- lane r takes voxel row r's 64 bytes of the chunk -- and are turned round in LDS (lane c then
// reads byte c of each row: 64 ds_read_u8 from consecutive addresses) instead of 64 one-byte loads per lane, each a load
// instruction of its own through the texture addresser.
constexpr int kCellChunks = (nn::kRegionMax / nn::kB + 63) / 64;  // 64-cell chunks of the longest row of cells
constexpr int kCellTileRow = 80;  // bytes between two voxel rows of a wave's tile (16-byte aligned, and 20 r spreads over the banks)
template <bool ROWS>
__global__ __launch_bounds__(1024) void k_nn_cells(NnArgs a) {  // sixteen cell rows per work-group, one per wave
  __shared__ uint32_t s_tot[16], s_start[16];
  __shared__ __attribute__((aligned(16))) uint8_t s_tile[ROWS ? 16 * 64 * kCellTileRow : 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 16 + wave;
  const nn::Geom &g = a.g;
  const bool rlive = row < g.ncx * g.ncy;  // (wave-uniform)
  const int cx = rlive ? row / g.ncy : 0, cy = rlive ? row % g.ncy : 0;
  const int nchunk = (g.ncz + 63) >> 6;  // cells of the row per lane (<= kCellChunks: a region has at most nn::kRegionMax voxels along z)
  uint32_t pk[kCellChunks][16];  // pk[k][r / 4] byte r % 4: the 8 z-bits of voxel row r (x = r / 8, y = r % 8) in cell lane + 64 k
  uint32_t cnt[kCellChunks] = {};
#pragma unroll
  for (int k = 0; k < kCellChunks; ++k) {
    const int c = lane + 64 * k;
    const bool has = rlive && k < nchunk && c < g.ncz;
    // (the bits of a row's last byte beyond the region: beyond the grid they are zero anyway, inside a shard's replica they
    //  belong to voxels the region does not hold)
    const uint32_t zmask = (nn::kB * c + nn::kB <= g.nz) ? 0xFFu : (0xFFu >> (nn::kB * c + nn::kB - g.nz));
#pragma unroll
    for (int q = 0; q < 16; ++q) pk[k][q] = 0;
    if (k < nchunk) {
      // the row of the cell row's voxel (0, 0); row r lies (r / 8) x-strides and (r % 8) y-strides farther
      const int64_t ystride = (int64_t)a.snzw * 4, xstride = ystride * a.sny;
      const uint8_t *row0 = reinterpret_cast<const uint8_t *>(a.occ) + (int64_t)(a.sx0 + nn::kB * cx) * xstride +
                            (int64_t)(a.sy0 + nn::kB * cy) * ystride + a.szb + (has ? c : 0);
      const int xin = g.nx - nn::kB * cx, yin = g.ny - nn::kB * cy;  // rows of the cell row inside the region (wave-uniform)
      if constexpr (ROWS) {
        const int rbytes = (int)ystride - a.szb;  // bytes of the bitmap's row from the region's first byte on (loads stay inside the row)
        const bool rin = rlive && (lane >> 3) < xin && (lane & 7) < yin;
        const uint8_t *src = reinterpret_cast<const uint8_t *>(a.occ) + (int64_t)(a.sx0 + nn::kB * cx + (rin ? lane >> 3 : 0)) * xstride +
                             (int64_t)(a.sy0 + nn::kB * cy + (rin ? lane & 7 : 0)) * ystride + a.szb + 64 * k;
        uint8_t *t = s_tile + wave * (64 * kCellTileRow);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // (a piece that would leave the row is read dword by dword as far as the row goes: rows are whole dwords)
          uint4 v{0u, 0u, 0u, 0u};
          const int left = rin ? rbytes - (64 * k + 16 * j) : 0;
          if (left >= 16 && (reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
            v = *reinterpret_cast<const uint4 *>(src + 16 * j);  // (an unsharded map whose rows are whole 16-byte pieces)
          } else if (left >= 16) {
            const uint32_t *p = reinterpret_cast<const uint32_t *>(src + 16 * j);  // dword-aligned: szb is a multiple of 4
            v = uint4{p[0], p[1], p[2], p[3]};
          } else if (left > 0) {
            const uint32_t *p = reinterpret_cast<const uint32_t *>(src + 16 * j);
            v.x = p[0];
            if (left > 4) v.y = p[1];
            if (left > 8) v.z = p[2];
          }
          *reinterpret_cast<uint4 *>(t + lane * kCellTileRow + 16 * j) = v;
        }
        // (LDS is in order within a wave: the reads below see the writes above, the next chunk's writes come after these reads)
#pragma unroll
        for (int r = 0; r < 64; ++r) {
          const uint32_t m = has ? ((uint32_t)t[r * kCellTileRow + lane] & zmask) : 0u;
          pk[k][r >> 2] |= m << (8 * (r & 3));
        }
      } else
#pragma unroll
      for (int r = 0; r < 64; ++r) {
        const bool in = (r >> 3) < xin && (r & 7) < yin;
        const uint32_t m = (in && has) ? ((uint32_t)row0[in ? (r >> 3) * xstride + (r & 7) * ystride : 0] & zmask) : 0u;
        pk[k][r >> 2] |= m << (8 * (r & 3));
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) cnt[k] += (uint32_t)__popc(pk[k][q]);
    }
  }
  uint32_t first[kCellChunks];
  uint32_t base = 0;
#pragma unroll
  for (int k = 0; k < kCellChunks; ++k) {
    uint32_t incl = cnt[k];
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
      if (lane >= off) incl += up;
    }
    first[k] = base + incl - cnt[k];
    base += (uint32_t)__shfl((int)incl, 63);
  }
  const uint32_t total = base;
  // the work-group's range of the site array: ONE atomic on the cursor (4096 of them, one per row, serialised for 60 us:
  // device-scope atomics on one address cross the XCDs)
  if (lane == 0) s_tot[wave] = total;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t sum = 0;
    for (int w = 0; w < 16; ++w) s_start[w] = sum, sum += s_tot[w];
    const uint32_t at = sum ? (uint32_t)atomicAdd(a.cursor, (unsigned long long)sum) : 0u;
    if ((unsigned long long)at + sum > a.sites_cap) atomicAdd(a.failed, 1ull);
    for (int w = 0; w < 16; ++w) s_start[w] += at;
  }
  __syncthreads();
  if (!rlive) return;
  const uint32_t start = s_start[wave];
  uint32_t *tab = a.ctab + (int64_t)row * (g.ncz + 1);
#pragma unroll
  for (int k = 0; k < kCellChunks; ++k) {
    const int c = lane + 64 * k;
    if (k < nchunk && c < g.ncz) tab[c] = start + first[k];
  }
  if (lane == 0) tab[g.ncz] = start + total;
  if (total == 0) return;  // (wave-uniform)
#pragma unroll
  for (int k = 0; k < kCellChunks; ++k) {
    if (cnt[k] == 0) continue;
    const int c = lane + 64 * k;
    uint32_t at = start + first[k];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      uint32_t m = pk[k][q];
      while (m) {  // in a cell: x-major, then y, then z
        const int bit = __ffs((int)m) - 1;
        m &= m - 1;
        const int r = 4 * q + (bit >> 3);
        if (at < a.sites_cap)
          a.sites[at] = (((uint32_t)(nn::kB * cx + (r >> 3)) & 1023u) << 20) | (((uint32_t)(nn::kB * cy + (r & 7)) & 1023u) << 10) |
                        ((uint32_t)(nn::kB * c + (bit & 7)) & 1023u);  // (region coordinates; modulo 1024 in a region beyond that)
        ++at;
      }
    }
  }
}

// ---- one list per cell ------------------------------------------------------------------------------------------------------
// The neighbourhood of a work-group's 64 (z) x 4 (y) cells in LDS: table rows and sites of the cells within kStageK = 4 of it
// (9 x 12 rows of cells).  A window that reaches farther (a cell whose nearest obstacle is more than ~24 voxels away) reads
// those rows from memory.
constexpr int kStageK = 4;
constexpr int kStageNX = 1 + 2 * kStageK, kStageNY = 4 + 2 * kStageK, kStageNR = kStageNX * kStageNY;  // 9 x 12 rows of cells
constexpr int kStageNZ = 64 + 2 * kStageK + 1;                                                          // table entries per row
constexpr int kStageSites = 3072;
template <bool WRAP>
struct StagedSrcT {  // the staged neighbourhood only: no range checks (rows outside the map are staged empty, entries are
  static constexpr bool wrap = WRAP;
  static constexpr int reach = kStageK;  // clamped into their row), no second path -- a window that needs more is served by
  const uint16_t *tab;     // HybridSrc in a second attempt.  [kStageNR][kStageNZ]: entry e of a row = sites of the row before cell Zf + e
  const uint32_t *soff;    // [kStageNR]: LDS index of the row's first staged site
  const uint32_t *lsites;
  int rbase, ebase;        // row index of (cx, cy) and entry index of cz for this lane's cell
  __device__ __forceinline__ void bounds(int X, int Y, int z0, int z1, uint32_t &i0, uint32_t &i1) const {
    const int row = X * kStageNY + Y + rbase;  // (X, Y arrive as cx + dx, cy + dy: rbase takes the origin out)
    const uint32_t d = soff[row];
    const uint16_t *t = tab + row * kStageNZ + ebase;
    i0 = t[z0] + d, i1 = t[z1 + 1] + d;
  }
  __device__ __forceinline__ uint32_t site(uint32_t i) const { return lsites[i]; }
};

// The second attempt's source for a cell whose window leaves the staged neighbourhood (its nearest obstacle is more than ~16
// voxels away: one cell in a thousand on config 2's scene): staged rows from LDS, the others from memory.  (Everything from
// memory is a chain of ~60 dependent L2 round trips per lane, and one such team holds its whole work-group back.)
template <bool WRAP>
struct HybridSrcT {
  static constexpr bool wrap = WRAP;
  static constexpr int reach = 1 << 20;
  StagedSrcT<WRAP> st;
  nn::PlainSrcT<WRAP> plain;
  int X0, Y0, Zf;
  __device__ __forceinline__ void bounds(int X, int Y, int z0, int z1, uint32_t &i0, uint32_t &i1) const {
    const int rx = X - X0, ry = Y - Y0;
    if (st.tab && (unsigned)rx < (unsigned)kStageNX && (unsigned)ry < (unsigned)kStageNY && z0 >= Zf && z1 + 1 - Zf < kStageNZ) {
      st.bounds(X, Y, z0, z1, i0, i1);
      i0 |= 0x80000000u, i1 |= 0x80000000u;
    } else {
      plain.bounds(X, Y, z0, z1, i0, i1);
    }
  }
  __device__ __forceinline__ uint32_t site(uint32_t i) const { return (i & 0x80000000u) ? st.lsites[i & 0x7FFFFFFFu] : plain.sites[i]; }
};

// A team of four adjacent lanes builds one list (nn_core.hpp: Team): the rows of the window dealt out among them.
struct QuadTeam {
  static constexpr int lanes = 4;
  int rank;
  uint32_t *counter;  // LDS: slots handed out so far (the lanes of a quad run in lock step: slots are deterministic)
  uint32_t *raw;      // LDS: the team's scratch, nn::kRaw words
  __device__ __forceinline__ void nearest(int &e2, uint32_t &w) const {
#pragma unroll
    for (int off = 1; off <= 2; off <<= 1) {
      const int oe = __shfl_xor(e2, off);
      const uint32_t ow = (uint32_t)__shfl_xor((int)w, off);
      if (oe < e2 || (oe == e2 && ow < w)) e2 = oe, w = ow;
    }
  }
  __device__ __forceinline__ void restart() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (rank == 0) (void)atomicExch(counter, 0u);  // (LDS atomics rather than volatile accesses: those go out as flat, system-scope)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  __device__ __forceinline__ int slot() { return (int)atomicAdd(counter, 1u); }
  // a team's lanes are lanes of one wave in lock step: a slot handed out by an earlier instruction has been put by the time
  // a later one reads it -- except the three other lanes' puts of the SAME step, which a lane may miss (harmless: it then
  // keeps a candidate it could have dropped)
  __device__ __forceinline__ int count_now() const { return (int)min(atomicAdd(counter, 0u), (uint32_t)nn::kRaw); }
  // (plain LDS accesses: what one lane of a team put is read by the others behind count()'s fence, or as count_now() says)
  __device__ __forceinline__ void put(int k, uint32_t v) { raw[k] = v; }
  __device__ __forceinline__ uint32_t get(int k) const { return raw[k]; }
  __device__ __forceinline__ int count() const {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the quad's lanes have left their loops: same wave, program order)
    return (int)atomicAdd(counter, 0u);
  }
};

// Sixteen adjacent lanes per list (the incremental transform: few cells, each a chain of dependent reads straight from memory --
// sixteen lanes make the chain a quarter as long as four do).
struct WideTeam {
  static constexpr int lanes = 16;
  int rank;
  uint32_t *counter;
  uint32_t *raw;
  __device__ __forceinline__ void nearest(int &e2, uint32_t &w) const {
#pragma unroll
    for (int off = 1; off <= 8; off <<= 1) {
      const int oe = __shfl_xor(e2, off);
      const uint32_t ow = (uint32_t)__shfl_xor((int)w, off);
      if (oe < e2 || (oe == e2 && ow < w)) e2 = oe, w = ow;
    }
  }
  __device__ __forceinline__ void restart() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (rank == 0) (void)atomicExch(counter, 0u);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  __device__ __forceinline__ int slot() { return (int)atomicAdd(counter, 1u); }
  __device__ __forceinline__ int count_now() const { return (int)min(atomicAdd(counter, 0u), (uint32_t)nn::kRaw); }
  __device__ __forceinline__ void put(int k, uint32_t v) { raw[k] = v; }
  __device__ __forceinline__ uint32_t get(int k) const { return raw[k]; }
  __device__ __forceinline__ int count() const {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    return (int)atomicAdd(counter, 0u);
  }
};

// WRAP: a region of more than 1024 voxels along an axis, its sites stored modulo 1024 (nn_core.hpp: site_offset)
template <bool WRAP>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_nn_lists(NnArgs a) {  // (two work-groups per CU: 64 VGPRs, < 80 KB of LDS)  // 64 (z) x 4 (y) cells, four lanes each
  __shared__ uint16_t s_tab[kStageNR * kStageNZ];
  __shared__ uint32_t s_first[kStageNR], s_soff[kStageNR], s_cnt[kStageNR];
  __shared__ uint32_t s_sites[kStageSites];
  __shared__ uint32_t s_slots[256];
  __shared__ uint32_t s_raw[256 * (nn::kRaw + 1)];  // (+ 1: the sixteen teams of a wave on different banks)
  __shared__ uint32_t s_total, s_bad, s_sum, s_done;
  __shared__ uint32_t s_te2[256], s_tw[256], s_bin[8];
  __shared__ uint16_t s_order[256];
  __shared__ uint8_t s_key[256];
  const nn::Geom &g = a.g;
  if (*a.failed) return;  // k_nn_cells ran out of room for the sites (a shard's region: its table points past the array)
  const int tid = (int)threadIdx.x;
  const int cz0 = g.lz0 + (int)blockIdx.x * 64, cy0 = g.ly0 + (int)blockIdx.y * 4, cx = g.lx0 + (int)blockIdx.z;  // (the cells that get a list)
  const int X0 = cx - kStageK, Y0 = cy0 - kStageK, Zf = cz0 - kStageK;
  const nn::Frame fr = nn::frame_of(g);
  const int64_t rowlen = g.ncz + 1;
  if (tid < 256) s_slots[tid] = 0;
  if (tid == 0) s_bad = 0, s_sum = 0, s_done = 0;
  if (tid < 8) s_bin[tid] = 0;
  // staging, step 1: per row of cells its range of the site array inside the neighbourhood's z-extent
  if (tid < kStageNR) {
    const int X = X0 + tid / kStageNY, Y = Y0 + tid % kStageNY;
    uint32_t f = 0, l = 0;
    if ((unsigned)X < (unsigned)g.ncx && (unsigned)Y < (unsigned)g.ncy) {
      const uint32_t *row = a.ctab + ((int64_t)X * g.ncy + Y) * rowlen;
      f = row[min(max(Zf, 0), g.ncz)], l = row[min(max(Zf + kStageNZ - 1, 0), g.ncz)];
    }
    s_first[tid] = f, s_cnt[tid] = l - f;
  }
  __syncthreads();
  if (tid < 64) {  // LDS offsets of the rows' sites: a scan over kStageNR <= 128 counts
    const uint32_t c0 = tid < kStageNR ? s_cnt[tid] : 0u, c1 = tid + 64 < kStageNR ? s_cnt[tid + 64] : 0u;
    uint32_t i0 = c0, i1 = c1;
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t u0 = (uint32_t)__shfl_up((int)i0, off), u1 = (uint32_t)__shfl_up((int)i1, off);
      if (tid >= off) i0 += u0, i1 += u1;
    }
    const uint32_t t0 = (uint32_t)__shfl((int)i0, 63);
    if (tid < kStageNR) s_soff[tid] = i0 - c0;
    if (tid + 64 < kStageNR) s_soff[tid + 64] = t0 + i1 - c1;
    if (tid == 63) s_total = t0 + i1;
  }
  __syncthreads();
  const bool staged = s_total <= (uint32_t)kStageSites;
  if (staged) {  // step 2: the table (16-bit, relative to the row's first staged site) and the sites
    for (int idx = tid; idx < kStageNR * kStageNZ; idx += 1024) {
      const int row = idx / kStageNZ, e = idx - row * kStageNZ;
      const int X = X0 + row / kStageNY, Y = Y0 + row % kStageNY;
      uint32_t v = 0;
      if ((unsigned)X < (unsigned)g.ncx && (unsigned)Y < (unsigned)g.ncy)
        v = a.ctab[((int64_t)X * g.ncy + Y) * rowlen + min(max(Zf + e, 0), g.ncz)] - s_first[row];
      s_tab[idx] = (uint16_t)v;
    }
    for (int idx = tid; idx < kStageNR * 16; idx += 1024) {
      const int row = idx >> 4;
      const uint32_t cnt = s_cnt[row], first = s_first[row], at = s_soff[row];
      for (uint32_t j = (uint32_t)(idx & 15); j < cnt; j += 16) s_sites[at + j] = a.sites[first + j];
    }
  }
  __syncthreads();
  // Phase A: team i finds the competitor of cell i of the work-group (16 cells per wave, consecutive in z) and how far its
  // search window reaches.  Then the work-group's cells are SORTED by that reach, and in phase B team i builds the list of the
  // i-th cell in that order: the lanes of a wave walk their windows in lock step, so a wave costs what its widest window
  // costs -- 13 row steps if one of its sixteen cells needs the 7 x 7 rows, 7 if all get by with 5 x 5 (four cells in five
  // do).  Sorted, thirteen waves of sixteen take the short walk.
  const int team_i = tid >> 2;
  QuadTeam team{tid & 3, &s_slots[team_i], &s_raw[team_i * (nn::kRaw + 1)]};
  const StagedSrcT<WRAP> ssrc{staged ? s_tab : nullptr, s_soff, s_sites, -(X0 * kStageNY + Y0), -Zf};
  {
    const int cz = cz0 + (team_i & 63), cy = cy0 + (team_i >> 6);
    const bool live = cz < g.lz1 && cy < g.ly1 && !(a.cellobs && a.cellobs[((int64_t)cx * g.ncy + cy) * g.ncz + cz] == 0);
    int te2 = nn::kNone, key = 7;  // (7: not a cell of the map, or one that wants no list)
    uint32_t tw = 0xFFFFFFFFu;
    if (live) {
      if (staged) nn::first_competitor(ssrc, team, cx, cy, cz, nn::kKfirst, te2, tw);
      key = min(nn::window_reach<WRAP>(te2, tw, cx, cy, cz), 6);
    }
    if (team.rank == 0) {
      s_te2[team_i] = (uint32_t)te2, s_tw[team_i] = tw, s_key[team_i] = (uint8_t)key;
      atomicAdd(&s_bin[key], 1u);
    }
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t at = 0;
    for (int k = 0; k < 8; ++k) {
      const uint32_t c = s_bin[k];
      s_bin[k] = at, at += c;
    }
  }
  __syncthreads();
  if (tid < 256) s_order[atomicAdd(&s_bin[s_key[tid]], 1u)] = (uint16_t)tid;  // (which team gets which cell does not change any list)
  __syncthreads();
  const int ci = s_order[team_i];
  const int cz = cz0 + (ci & 63), cy = cy0 + (ci >> 6);
  int n = 0;
  bool live = cz < g.lz1 && cy < g.ly1;  // (the same for the four lanes of a team)
  if (live && a.cellobs && a.cellobs[((int64_t)cx * g.ncy + cy) * g.ncz + cz] == 0) {  // an empty record: the fill writes words nobody keeps
    live = false;
    if (team.rank == 0) a.lists[(((int64_t)cx * g.ncy + cy) * g.ncz + cz) * nn::kStride] = 0u;
  }
  if (live) {
    const int64_t cell = ((int64_t)cx * g.ncy + cy) * g.ncz + cz;
    uint32_t *rec = a.lists + cell * nn::kStride;
    n = -1;
    if (staged) n = nn::build_list(ssrc, team, cx, cy, cz, rec, true, (int)s_te2[ci], s_tw[ci], fr);
    if (n < 0) {  // (a team decides together: the window needs more than the staged cells -- or nothing was staged)
      const HybridSrcT<WRAP> src{ssrc, nn::PlainSrcT<WRAP>{a.ctab, a.sites, g.ncx, g.ncy, g.ncz}, X0, Y0, Zf};
      n = nn::build_list(src, team, cx, cy, cz, rec, false, nn::kNone, 0xFFFFFFFFu, fr);
    }
  }
  // a cell without a list: on the list of cells k_nn_brute serves one by one, while that has room and the region has no open
  // face (a plain load first: a scene that leaves EVERY cell without a list must not queue 10^5 atomics on one address)
  bool hard = live && team.rank == 0 && n == 0;
  if (hard && a.fail_list && fr.open == 0 && *a.nfail < a.fail_cap) {
    const unsigned long long at = atomicAdd(a.nfail, 1ull);
    if (at < a.fail_cap) {
      a.fail_list[at] = (uint32_t)(((int64_t)cx * g.ncy + cy) * g.ncz + cz);
      hard = false;
    }
  }
  // statistics: failures and entries, one atomic each per work-group -- by the LAST wave to get here, not behind a barrier
  // (a wave that waits for its work-group's slowest team keeps its registers from the next work-group)
  const unsigned long long bad = __ballot(hard);
  int sum = (live && team.rank == 0) ? n : 0;
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
  if ((tid & 63) == 0) {
    if (bad) atomicAdd(&s_bad, (uint32_t)__popcll(bad));
    if (sum) atomicAdd(&s_sum, (uint32_t)sum);
    __threadfence_block();
    if (atomicAdd(&s_done, 1u) == 15u) {
      const uint32_t b = atomicAdd(&s_bad, 0u), t = atomicAdd(&s_sum, 0u);
      if (b) atomicAdd(a.failed, (unsigned long long)b);
      if (t) atomicAdd(a.entries, (unsigned long long)t);
    }
  }
}

// ---- every voxel's minimum over its cell's list -------------------------------------------------------------------------------
// Sums of the four signed byte products of a with b0 and with b1 (v_dot4_i32_i8, VOP3P).  Written as ONE asm statement that
// ends in s_nop 2: on gfx940+ a VALU instruction must not read a DOT result within three wait states of the DOT, hipcc's
// hazard recognizer pads its own DOTs but cannot look into inline asm, and without the padding the sums read back wrong
// (tools/dev/serve_probe.hip reproduces it: the same loop is right with the nops, and with __builtin_amdgcn_sdot4 -- which
// costs a mov + v_dot4c per product, two instructions more per pair of entries).  Early-clobber outputs: a DOT must not be
// allocated onto its own sources either.
__device__ __forceinline__ void nn_dot4_pair(uint32_t a, uint32_t b0, uint32_t b1, int &d0, int &d1) {
  asm("v_dot4_i32_i8 %0, %2, %3, 0\n\tv_dot4_i32_i8 %1, %2, %4, 0\n\ts_nop 2" : "=&v"(d0), "=&v"(d1) : "v"(a), "v"(b0), "v"(b1));
}

// persistent work-groups of four waves.  Measured on config 2 (512^3, r06): 1024 -> 162 us, 2048 -> 157 (r05's choice: 8 quads per
// wave), 4096 -> 148, 8192 -> 145, 16384 -> 136: one quad (four cells) per wave, the tail of the launch evens out
#ifndef FIESTA_FILL_BLOCKS
#define FIESTA_FILL_BLOCKS 16384
#endif
constexpr int kFillBlocks = FIESTA_FILL_BLOCKS;
constexpr int kListPad = 256;      // dwords the list array is over-allocated by (the last 128: where masked lanes of the offset fill store)

// A QUAD is four cells adjacent in z: 32 voxels, one 128-byte line per (x, y) row.  ONE WAVE serves a quad -- its four cells
// one after the other, the 4 x 8 winners' words kept in registers -- and then stores it slab by slab through a 1.25 KB LDS
// tile of its own: 16 bytes per lane, eight whole lines per store instruction.  (Stored cell by cell, eight 32-byte pieces
// per instruction, the same bytes took 270 us instead of 120; exchanged between the four waves of a work-group behind two
// barriers per cell, 170 us: a wave stalled on its fetch stalled the other three.)  The quads of the map in row-major
// order are dealt out in contiguous runs, one run per wave.
//
// Memory choreography of one wave (the k_ft_x recipe, ft_kernels.hpp): the next cell's record -- 512 bytes: the count, then
// 16 bytes (b, K, m, W) per entry -- is fetched by LDS-DMA (global_load_lds_dword: lane i's dword lands at m0 + 4 i, no
// register is in flight) into the other of two landing zones BEFORE this cell's arithmetic, issued and waited for by hand:
// hipcc never sees a load, so it never drains vmcnt for one, and a quad's stores stay in flight behind the next fetch
// (gfx9 retires the loads and stores of a wave in issue order).  An entry is read back as ONE ds_read_b128 from a
// wave-uniform address (a broadcast): b, K and m arrive in VGPRs holding the same value in every lane -- no scalar loads,
// no readlane.  FULL maps only: every extent a multiple of the cell edge, rows a multiple of four cells long, the same
// number of quads for every wave -- no predicate anywhere, so the number of memory operations between a fetch and its wait
// is the same on every path.  Other maps take the simple predicated variant below.
// OFFSET: the array lies at an offset inside its region (a shard: owned box + ghost layers), cut by cells on every side, but
// pairs of voxels stay aligned (even row length, even z-offset).  The same wave-per-quad choreography over the cells that
// hold a voxel of the array, 8-byte stores (16 lanes per 32-voxel row piece, two store instructions per slab), and -- since
// the counted waits need the same number of memory operations on every path -- NO predicated store: a lane whose pair lies
// outside the array stores into a dump area instead.  Runs may differ in length by one quad (each wave counts its own).
template <bool TRACK, bool OFFSET = false>
__global__ __launch_bounds__(256) void k_nn_fill_full(NnArgs a) {
  constexpr int kTileRow = 40;  // dwords between two rows of the tile: 32 + padding that spreads the rows over the banks
  __shared__ __attribute__((aligned(16))) uint32_t land[4][2][nn::kStride];  // per wave, two zones of one record each
  __shared__ __attribute__((aligned(16))) uint32_t tile[4][nn::kB * kTileRow];  // per wave, one x-slab of a quad: [y][32 z]
  const nn::Geom &g = a.g;
  if (*a.failed) return;  // some cell has no list: the envelope passes serve this update (dense_map.hip)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // the quads: of the whole map, or (OFFSET) of the cells that got a list -- lx0.., coordinates relative to that range
  const int quads = OFFSET ? (g.lz1 - g.lz0 + 3) >> 2 : g.ncz >> 2;
  const int rows_y = OFFSET ? g.ly1 - g.ly0 : g.ncy;
  const uint32_t nq = (uint32_t)((OFFSET ? g.lx1 - g.lx0 : g.ncx) * rows_y) * (uint32_t)quads;
  // (full maps: gridDim.x * 4 * per == nq, the host's choice; OFFSET: the last waves' runs are shorter or empty)
  const uint32_t per = OFFSET ? (nq + gridDim.x * 4u - 1u) / (gridDim.x * 4u) : nq / (gridDim.x * 4u);
  const uint32_t q0 = (blockIdx.x * 4u + (uint32_t)wave) * per, q1 = OFFSET ? min(q0 + per, nq) : q0 + per;
  if (OFFSET && q0 >= q1) return;
  const int row0 = (int)(q0 / (uint32_t)quads);
  const int y = lane >> 3, z = lane & 7;
  const uint32_t ayz = (uint32_t)y | ((uint32_t)z << 8);
  const int64_t plane = (int64_t)g.ny * g.nz;
  const int loff4 = y * g.nz + 4 * z;  // the lane's 16 bytes of a stored slab: row y, voxels 4 z .. 4 z + 3 of the quad's 32
  uint32_t dmax = 0;
  const uint32_t zone0 = (uint32_t)(size_t)&land[wave][0][0];  // LDS byte offsets: the low half of the generic address
  uint32_t *const mytile = &tile[wave][0];
  struct At { int cx, cy, q; };  // a quad: cell row (cx, cy), quad q of it
  auto fetch = [&](const At &c, const int cell_of_quad, const int zone) {
    // (OFFSET: a quad cut by the end of the range fetches its last cell again -- never stored: no branch around a load)
    const int64_t cell = OFFSET ? ((int64_t)(g.lx0 + c.cx) * g.ncy + (g.ly0 + c.cy)) * g.ncz + min(g.lz0 + 4 * c.q + cell_of_quad, g.lz1 - 1)
                                : ((int64_t)c.cx * g.ncy + c.cy) * g.ncz + 4 * c.q + cell_of_quad;
    const uint32_t *lp = a.lists + cell * nn::kStride + lane, *hp = lp + 64;
    const uint32_t at = zone0 + (uint32_t)zone * (uint32_t)(nn::kStride * 4);
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, off\n\t"
                 "s_mov_b32 m0, %2\n\tglobal_load_lds_dword %3, off"
                 : : "s"(at), "v"(lp), "s"(at + 256u), "v"(hp) : "memory", "m0");
  };
  auto advance = [&](At &c) {
    if (++c.q == quads) {
      c.q = 0;
      if (++c.cy == rows_y) c.cy = 0, ++c.cx;
    }
  };
  // one cell: every voxel's minimum key over the record in `zone`, the winners' words into ww[0..7]
  auto serve = [&](const int zone, uint32_t *ww, const bool col_in, const int xin_lo, const int xin_hi) {
    const uint32_t *lz = &land[wave][zone][0];
    const int cnt = __builtin_amdgcn_readfirstlane((int)lz[0]);
    uint32_t best[nn::kB];
#pragma unroll
    for (int x = 0; x < nn::kB; ++x) best[x] = 0xFFFFFFFFu;
    for (int i = 0; i < cnt; i += 2) {  // two entries per step (an odd list ends with a padding entry)
      const uint4 e0 = *reinterpret_cast<const uint4 *>(lz + 4 + 4 * i), e1 = *reinterpret_cast<const uint4 *>(lz + 8 + 4 * i);
      int d0, d1;
      nn_dot4_pair(ayz, e0.x, e1.x, d0, d1);
      uint32_t k0 = ((uint32_t)d0 << nn::kSH) + e0.y, k1 = ((uint32_t)d1 << nn::kSH) + e1.y;
      best[0] = min(best[0], min(k0, k1));
#pragma unroll
      for (int x = 1; x < nn::kB; ++x) {
        k0 += e0.z, k1 += e1.z;
        best[x] = min(best[x], min(k0, k1));
      }
    }
#pragma unroll
    for (int x = 0; x < nn::kB; ++x) {
      ww[x] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(lz + 7) + (best[x] & 0x1F0u));
      // (OFFSET: only voxels of the array count for the distance bound)
      if (TRACK && (!OFFSET || (col_in && x >= xin_lo && x < xin_hi)))
        dmax = max(dmax, (best[x] >> nn::kSH) - (uint32_t)nn::kBias + (uint32_t)(x * x + y * y + z * z));
    }
  };
  // The wave's VMEM operations in issue order, per quad:  F1 | F2 | F3 | F0' | S x 8   (F: the two loads of a cell's fetch,
  // issued as the cell before it begins; F0': the next quad's first cell; S: the quad's stores).  A fetch is needed one cell
  // after it was issued: behind it then lie the next fetch, and -- for a quad's first cell -- the eight stores of the quad before.
  At c{row0 / rows_y, row0 % rows_y, (int)(q0 % (uint32_t)quads)};
  fetch(c, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (uint32_t q = q0; q < q1; ++q) {
    uint32_t ww[4][nn::kB];
    At nxt = c;
    if (q + 1 < q1) advance(nxt);  // (the run's last fetch re-reads its last quad's first cell: no branch around a load)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // zone (k + 1) & 1 was last read by the cell before: its LDS reads are done (lgkmcnt(0) below)
      if (k < 3) fetch(c, k + 1, (k + 1) & 1); else fetch(nxt, 0, 0);
      // behind this cell's fetch: the fetch just issued, and for a quad's first cell the stores of the quad before (8; OFFSET: 16)
      if (k == 0) {
        if (OFFSET) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      }
      bool col_in = true;
      int xin_lo = 0, xin_hi = nn::kB;
      if (OFFSET && TRACK) {  // which of this lane's voxels of the cell lie in the array
        const int cz = g.lz0 + 4 * c.q + k;
        const int Y = nn::kB * (g.ly0 + c.cy) + y - g.fy, Z = nn::kB * cz + z - g.fz, X0 = nn::kB * (g.lx0 + c.cx) - g.fx;
        col_in = cz < g.lz1 && (unsigned)Y < (unsigned)g.ay && (unsigned)Z < (unsigned)g.az;
        xin_lo = max(0, -X0), xin_hi = min(nn::kB, g.ax - X0);
      }
      serve(k & 1, ww[k], col_in, xin_lo, xin_hi);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (!OFFSET) {
      vox_t *slab = a.coc + ((int64_t)(nn::kB * c.cx) * g.ny + nn::kB * c.cy) * g.nz + 4 * nn::kB * c.q;
#pragma unroll
      for (int x = 0; x < nn::kB; ++x) {
#pragma unroll
        for (int k = 0; k < 4; ++k) mytile[y * kTileRow + nn::kB * k + z] = ww[k][x];
        const uint4 v = *reinterpret_cast<const uint4 *>(&mytile[y * kTileRow + 4 * z]);  // (LDS is in order within a wave)
        *reinterpret_cast<uint4 *>(slab + loff4) = v;  // lane (y, z): row y of the slab, z-voxels 4 z .. 4 z + 3 of the quad
        slab += plane;
      }
    } else {
      // lane -> row (lane / 16) + 4 h of the slab, voxels 2 (lane % 16), + 1 of the quad's 32; the array's coordinates
      const int srow = lane >> 4, spair = lane & 15;
      const int X0 = nn::kB * (g.lx0 + c.cx) - g.fx, Y0 = nn::kB * (g.ly0 + c.cy) - g.fy;
      const int Zq = nn::kB * (g.lz0 + 4 * c.q) + 2 * spair;  // region z of the pair (even; fz is even too)
      const bool zin = (unsigned)(Zq - g.fz) < (unsigned)g.az && Zq < nn::kB * g.lz1;
      const int64_t aplane = (int64_t)g.ay * g.az;
      uint32_t *const mydump = a.dump + 2 * lane;
#pragma unroll
      for (int x = 0; x < nn::kB; ++x) {
#pragma unroll
        for (int k = 0; k < 4; ++k) mytile[y * kTileRow + nn::kB * k + z] = ww[k][x];
        const bool xin = (unsigned)(X0 + x) < (unsigned)g.ax;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int r = srow + 4 * h;
          const uint2 v = *reinterpret_cast<const uint2 *>(&mytile[r * kTileRow + 2 * spair]);  // (LDS is in order within a wave)
          const bool in = xin && zin && (unsigned)(Y0 + r) < (unsigned)g.ay;
          uint32_t *dst = in ? a.coc + (int64_t)(X0 + x) * aplane + (int64_t)(Y0 + r) * g.az + (Zq - g.fz) : mydump;
          *reinterpret_cast<uint2 *>(dst) = v;  // (always issued: the waits above count it)
        }
      }
    }
    c = nxt;
  }
  if (TRACK) {
    for (int off = 32; off > 0; off >>= 1) dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, off));
    if (lane == 0 && (unsigned long long)dmax > *a.maxd2) atomicMax(a.maxd2, (unsigned long long)dmax);
  }
}

// The predicated variant for maps with cells cut by the array's faces: one wave per cell, the record through the scalar cache.
typedef __attribute__((address_space(4))) const uint32_t nn_cu32;  // constant address space: wave-uniform reads become s_load
typedef uint32_t nn_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(4))) const nn_u32x4 nn_cu32x4;
// one cell's fill by ONE wave (lane = the cell's (y, z) column): the record through the scalar cache, predicated stores
template <bool TRACK>
__device__ __forceinline__ void nn_fill_cell(const NnArgs &a, const int cx, const int cy, const int cz, const int lane, uint32_t &dmax) {
  const nn::Geom &g = a.g;
  const int64_t cell = ((int64_t)cx * g.ncy + cy) * g.ncz + cz;
  const uint32_t *rec = a.lists + cell * nn::kStride;
  nn_cu32 *lp = reinterpret_cast<nn_cu32 *>(reinterpret_cast<uintptr_t>(rec));
  const int cnt = (int)lp[0];
  const int y = lane >> 3, z = lane & 7;
  const uint32_t ayz = (uint32_t)y | ((uint32_t)z << 8);
  uint32_t best[nn::kB];
#pragma unroll
  for (int x = 0; x < nn::kB; ++x) best[x] = 0xFFFFFFFFu;
  for (int i = 0; i < cnt; ++i) {
    const nn_u32x4 e = *reinterpret_cast<nn_cu32x4 *>(lp + 4 + 4 * i);
    uint32_t k = ((uint32_t)__builtin_amdgcn_sdot4((int)ayz, (int)e.x, 0, false) << nn::kSH) + e.y;
    best[0] = min(best[0], k);
#pragma unroll
    for (int x = 1; x < nn::kB; ++x) {
      k += e.z;
      best[x] = min(best[x], k);
    }
  }
  // the voxel in the ARRAY's coordinates (a shard's array lies somewhere inside the region)
  const int X0 = nn::kB * cx - g.fx, Y = nn::kB * cy + y - g.fy, Z = nn::kB * cz + z - g.fz;
  const bool inyz = (unsigned)Y < (unsigned)g.ay && (unsigned)Z < (unsigned)g.az;
  vox_t *out = a.coc + ((int64_t)X0 * g.ay + Y) * g.az + Z;
  const int64_t plane = (int64_t)g.ay * g.az;
#pragma unroll
  for (int x = 0; x < nn::kB; ++x) {
    const uint32_t ww = rec[7 + ((best[x] & 0x1F0u) >> 2)];
    if (inyz && (unsigned)(X0 + x) < (unsigned)g.ax) {
      out[x * plane] = ww;
      if (TRACK) dmax = max(dmax, (best[x] >> nn::kSH) - (uint32_t)nn::kBias + (uint32_t)(x * x + y * y + z * z));
    }
  }
}

template <bool TRACK>
__global__ __launch_bounds__(256) void k_nn_fill(NnArgs a) {
  const nn::Geom &g = a.g;
  if (*a.failed) return;
  // grid: the cells that got a list, (ceil((lz1 - lz0) / 4), ly1 - ly0, lx1 - lx0)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int cz = g.lz0 + (int)blockIdx.x * 4 + wave, cy = g.ly0 + (int)blockIdx.y, cx = g.lx0 + (int)blockIdx.z;
  if (cz >= g.lz1) return;
  uint32_t dmax = 0;
  nn_fill_cell<TRACK>(a, cx, cy, cz, lane, dmax);
  if (TRACK) {
    for (int off = 32; off > 0; off >>= 1) dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, off));
    if (lane == 0 && (unsigned long long)dmax > *a.maxd2) atomicMax(a.maxd2, (unsigned long long)dmax);
  }
}

// ---- the incremental transform: only the cells a change can reach ---------------------------------------------------------------------
// A cell's list depends on the sites inside its search window and on nothing else (nn_core.hpp: build_list), and its record keeps
// the window's reach: a voxel that changed occupancy dirties exactly the cells within whose reach it lies.  k_nn_mark: a work-group
// per changed voxel, its threads over the (2 kKmax + 1)^3 cells around it; k_nn_lists_dirty: a team of four lanes per dirty cell,
// straight from memory (nn::PlainSrc: a few thousand cells do not pay for staging a neighbourhood); k_nn_fill_dirty: a wave per
// dirty cell.  More dirty cells than the list holds fails the transform (the full one serves the update).
constexpr int kMarkQueue = 4096;
__global__ __launch_bounds__(1024) void k_nn_mark(NnArgs a) {
  __shared__ uint32_t s_q[kMarkQueue];
  __shared__ uint32_t s_n;
  __shared__ unsigned long long s_base;
  const nn::Geom &g = a.g;
  const uint32_t total = a.nchg[0] + a.nchg[1];
  constexpr int E = 2 * nn::kKmax + 1, E3 = E * E * E;
  // a lane per (changed voxel, cell of the (2 kKmax + 1)^3 around it): every record read of the launch in flight at once.  The
  // cells found go to a queue in LDS and from there to the list with ONE atomic on the list's cursor per work-group and flush
  // (an atomic per wave, ~5 000 of them on one address, was 20 of this kernel's 24 us)
  const unsigned long long pairs = (unsigned long long)total * E3;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  for (unsigned long long p0 = (unsigned long long)blockIdx.x * blockDim.x; p0 < pairs; p0 += (unsigned long long)gridDim.x * blockDim.x) {  // (uniform)
    const unsigned long long p = p0 + threadIdx.x;
    if (p < pairs) {
      const uint32_t v = (uint32_t)(p / E3);
      const int o = (int)(p - (unsigned long long)v * E3);
      const uint32_t idx = v < a.nchg[0] ? a.chg[0][v] : a.chg[1][v - a.nchg[0]];
      const int z = (int)(idx % (uint32_t)g.nz), y = (int)((idx / (uint32_t)g.nz) % (uint32_t)g.ny), x = (int)(idx / ((uint32_t)g.nz * (uint32_t)g.ny));
      const int dz = o % E - nn::kKmax, dy = (o / E) % E - nn::kKmax, dx = o / (E * E) - nn::kKmax;
      const int cx = (x >> 3) + dx, cy = (y >> 3) + dy, cz = (z >> 3) + dz;
      if ((unsigned)cx < (unsigned)g.ncx && (unsigned)cy < (unsigned)g.ncy && (unsigned)cz < (unsigned)g.ncz) {
        const int64_t c = ((int64_t)cx * g.ncy + cy) * g.ncz + cz;
        const int r = max(max(dx < 0 ? -dx : dx, dy < 0 ? -dy : dy), dz < 0 ? -dz : dz);
        if (r <= (int)(a.lists[c * nn::kStride + 1] & 255u) && a.dirty_flag[c] == 0u && atomicExch(&a.dirty_flag[c], 1u) == 0u)
          s_q[atomicAdd(&s_n, 1u)] = (uint32_t)c;  // (at most kMarkQueue - 1024 queued when an iteration starts)
      }
    }
    __syncthreads();
    const uint32_t n = s_n;
    const bool last = p0 + (unsigned long long)gridDim.x * blockDim.x >= pairs;
    if (n && (last || n > (uint32_t)kMarkQueue - 1024u)) {  // (uniform)
      if (threadIdx.x == 0) s_base = atomicAdd(a.dirty_count, (unsigned long long)n);
      __syncthreads();
      const unsigned long long base = s_base;
      for (uint32_t i = threadIdx.x; i < n; i += blockDim.x)
        if (base + i < a.dirty_cap) a.dirty_list[base + i] = s_q[i];
      __syncthreads();
      if (threadIdx.x == 0) s_n = 0;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_nn_lists_dirty(NnArgs a) {
  __shared__ uint32_t s_slots[16];
  __shared__ uint32_t s_raw[16 * (nn::kRaw + 1)];
  const nn::Geom &g = a.g;
  if (*a.failed) return;
  const unsigned long long nd = *a.dirty_count;
  if (nd > a.dirty_cap) {  // (too much has changed for the dirty list: this transform fails)
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(a.failed, 1ull);
    return;
  }
  const int tid = (int)threadIdx.x, team_i = tid >> 4;
  const nn::Frame fr = nn::frame_of(g);
  const nn::PlainSrcT<false> src{a.ctab, a.sites, g.ncx, g.ncy, g.ncz};
  unsigned bad = 0, entries = 0;
  for (unsigned long long i = blockIdx.x * 16ull + (unsigned long long)team_i; i < nd; i += gridDim.x * 16ull) {
    const uint32_t c = a.dirty_list[i];
    const int cz = (int)(c % (uint32_t)g.ncz), cy = (int)((c / (uint32_t)g.ncz) % (uint32_t)g.ncy), cx = (int)(c / ((uint32_t)g.ncz * (uint32_t)g.ncy));
    WideTeam team{tid & 15, &s_slots[team_i], &s_raw[team_i * (nn::kRaw + 1)]};
    if (team.rank == 0) s_slots[team_i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const int n = nn::build_list(src, team, cx, cy, cz, a.lists + (int64_t)c * nn::kStride, false, nn::kNone, 0xFFFFFFFFu, fr);
    if (team.rank == 0) {
      if (n <= 0) ++bad; else entries += (unsigned)n;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    bad += (unsigned)__shfl_xor((int)bad, off);
    entries += (unsigned)__shfl_xor((int)entries, off);
  }
  if ((tid & 63) == 0) {
    if (bad) atomicAdd(a.failed, (unsigned long long)bad);
    (void)entries;  // (the statistics of an incremental transform: the lists it rebuilt are not added to the total)
  }
}

template <bool TRACK>
__global__ __launch_bounds__(256) void k_nn_fill_dirty(NnArgs a) {
  const nn::Geom &g = a.g;
  if (*a.failed) return;
  const unsigned long long nd = *a.dirty_count;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint32_t dmax = 0;
  for (unsigned long long i = blockIdx.x * 4ull + (unsigned long long)wave; i < nd; i += gridDim.x * 4ull) {
    const uint32_t c = a.dirty_list[i];
    const int cz = (int)(c % (uint32_t)g.ncz), cy = (int)((c / (uint32_t)g.ncz) % (uint32_t)g.ncy), cx = (int)(c / ((uint32_t)g.ncz * (uint32_t)g.ncy));
    nn_fill_cell<TRACK>(a, cx, cy, cz, lane, dmax);
    if (lane == 0) a.dirty_flag[c] = 0u;
  }
  if (TRACK) {
    for (int off = 32; off > 0; off >>= 1) dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, off));
    if (lane == 0 && (unsigned long long)dmax > *a.maxd2) atomicMax(a.maxd2, (unsigned long long)dmax);
  }
}

// ---- the cells a scene leaves without a list, one by one ---------------------------------------------------------------------------------
// A work-group per such cell, a thread per two voxels: the nearest site by brute force -- among EVERY site where the cell found
// nothing within its widest window (a sparse corner of the map), among the sites of its window where it found more survivors
// than a list holds (a cell over a dense cluster).  Exact like the lists (the window holds every site that can win in the cell).
// Runs behind the fill, whose words for these cells (their records are empty) it overwrites.  A handful of cells per update at
// the ends of the density range the cell transform is tried in; a scene that leaves more than fail_cap cells without a list fails
// the transform as before.
__device__ __forceinline__ void nn_brute_cells(const NnArgs &a, uint32_t *s_sites) {
  const nn::Geom &g = a.g;
  const bool TRACK = a.maxd2 != nullptr;
  if (*a.failed) return;
  const uint32_t nf = (uint32_t)min(*a.nfail, (unsigned long long)a.fail_cap);
  const uint32_t total = (uint32_t)min(*a.cursor, (unsigned long long)a.sites_cap);
  const int t = (int)threadIdx.x;
  uint32_t dmax = 0;
  for (uint32_t fi = blockIdx.x; fi < nf; fi += gridDim.x) {
    const uint32_t c = a.fail_list[fi];
    const int cz = (int)(c % (uint32_t)g.ncz), cy = (int)((c / (uint32_t)g.ncz) % (uint32_t)g.ncy), cx = (int)(c / ((uint32_t)g.ncz * (uint32_t)g.ncy));
    const uint32_t info = a.lists[(int64_t)c * nn::kStride + 1];
    const bool dense = ((info >> 8) & 255u) == (uint32_t)nn::kWhyDense;
    const int K = dense ? (int)(info >> 16) : 0;
    // this thread's two voxels of the cell: (x, y, z) and (x + 4, y, z)
    const int vx = nn::kB * cx + (t >> 6), vy = nn::kB * cy + ((t >> 3) & 7), vz = nn::kB * cz + (t & 7);
    uint32_t b0 = 0xFFFFFFFFu, b1 = 0xFFFFFFFFu, w0 = 0, w1 = 0;
    auto range = [&](uint32_t i0, const uint32_t i1) {  // the sites [i0, i1) against the cell's voxels (every thread calls)
      for (; i0 < i1; i0 += 256u) {
        __syncthreads();
        if (i0 + (uint32_t)t < i1) s_sites[t] = a.sites[i0 + (uint32_t)t];
        __syncthreads();
        const int m = (int)min(256u, i1 - i0);
        for (int k = 0; k < m; ++k) {
          const uint32_t w = s_sites[k];
          int sx, sy, sz;
          nn::unpack_site(w, sx, sy, sz);
          const int dy = sy - vy, dz = sz - vz, d0 = sx - vx, d1 = d0 - 4;
          const uint32_t r = (uint32_t)(dy * dy + dz * dz);
          const uint32_t e0 = r + (uint32_t)(d0 * d0), e1 = r + (uint32_t)(d1 * d1);
          if (e0 < b0) b0 = e0, w0 = w;
          if (e1 < b1) b1 = e1, w1 = w;
        }
      }
    };
    if (dense) {
      for (int dx = -K; dx <= K; ++dx)
        for (int dy = -K; dy <= K; ++dy) {
          const int X = cx + dx, Y = cy + dy;
          if ((unsigned)X >= (unsigned)g.ncx || (unsigned)Y >= (unsigned)g.ncy) continue;  // (block-uniform)
          const uint32_t *row = a.ctab + ((int64_t)X * g.ncy + Y) * (g.ncz + 1);
          range(row[max(cz - K, 0)], row[min(cz + K, g.ncz - 1) + 1]);
        }
    } else {
      range(0u, total);
    }
    // the words a voxel stores: global coordinates (the region's origin added), modulo 1024 -- as an entry's W (nn_core.hpp)
    auto word = [&](const uint32_t w) {
      int sx, sy, sz;
      nn::unpack_site(w, sx, sy, sz);
      return (((uint32_t)(sx + g.wx) & 1023u) << 20) | (((uint32_t)(sy + g.wy) & 1023u) << 10) | ((uint32_t)(sz + g.wz) & 1023u);
    };
    const int Y = vy - g.fy, Z = vz - g.fz, X0 = vx - g.fx;
    if ((unsigned)Y < (unsigned)g.ay && (unsigned)Z < (unsigned)g.az) {
      const int64_t plane = (int64_t)g.ay * g.az;
      vox_t *out = a.coc + ((int64_t)X0 * g.ay + Y) * g.az + Z;
      if ((unsigned)X0 < (unsigned)g.ax && b0 != 0xFFFFFFFFu) {
        out[0] = word(w0);
        if (TRACK) dmax = max(dmax, b0);
      }
      if ((unsigned)(X0 + 4) < (unsigned)g.ax && b1 != 0xFFFFFFFFu) {
        out[4 * plane] = word(w1);
        if (TRACK) dmax = max(dmax, b1);
      }
    }
  }
  if (TRACK) {
    for (int off = 32; off > 0; off >>= 1) dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, off));
    if ((t & 63) == 0 && (unsigned long long)dmax > *a.maxd2) atomicMax(a.maxd2, (unsigned long long)dmax);
  }
}


// ---- the transform's last launch: the cells without a list, then one thread reports and cleans up ----------------------------------
// One host synchronisation then ends the update: no copy of the counters, no reset launch ahead of the next transform.  (The
// same work done by the fill's last work-group -- a ticket every work-group takes at its end -- doubled the fill's time: 2048
// returning atomics on one address, all at the same moment.)  The launch has as many work-groups as the LAST transform left cells
// without a list (one, normally: a scene inside the density range leaves none, and the first few that appear are served by that
// one group); the last group to finish closes.  pub == nullptr (a shard's try, a masked transform): no report, the host reads
// the counters itself.
__global__ __launch_bounds__(256) void k_nn_close(NnArgs a) {
  __shared__ uint32_t s_sites[256];
  __shared__ uint32_t s_last;
  if (a.nfail) nn_brute_cells(a, s_sites);
  if (!a.pub) return;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    s_last = atomicAdd(a.ticket, 1ull) == (unsigned long long)gridDim.x - 1ull;
  }
  __syncthreads();
  if (!s_last || threadIdx.x != 0) return;
  __threadfence();
  const unsigned long long failed = *a.failed, entries = *a.entries, md = a.maxd2 ? atomicMax(a.maxd2, 0ull) : 0ull;
  volatile unsigned long long *h = a.pub;
  h[a.pub_failed] = failed, h[a.pub_entries] = entries, h[a.pub_maxd2] = md;
  if (a.dirty_count) h[a.pub_dirty] = *a.dirty_count;
  if (a.nfail) h[a.pub_brute] = min(*a.nfail, (unsigned long long)a.fail_cap), *a.nfail = 0;
  if (a.track_dst && failed == 0) *a.track_dst = a.dirty_flag ? max(*a.track_dst, md) : md;  // (incremental: the cells left alone keep theirs)
  *a.cursor = 0, *a.failed = 0, *a.entries = 0, *a.ticket = 0;
  if (a.dirty_count) *a.dirty_count = 0;
  if (a.maxd2) *a.maxd2 = 0;
  if (failed == 0) a.queues[0] = 0, a.queues[1] = 0;
  __threadfence_system();
  h[a.pub_tag] = a.tag;  // (last: the host trusts the other values once it sees this update's tag)
  __threadfence_system();
}

}  // namespace fiesta
