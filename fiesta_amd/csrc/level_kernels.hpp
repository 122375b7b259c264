// fiesta_amd/csrc/level_kernels.hpp -- the LEVEL engine of UpdateESDF: a globally level-synchronous sparse-frontier sweep
// that works on the voxel words in HBM directly (no tiles, no staging).  Shared by the dense-array map (dense_map.hip) and
// the paged hash-block map (hash_map.hip) through an address-space policy.
//
// The reference's update queue is a FIFO (src/ESDFMap.cpp:339-392), so its entries are processed in LAYERS: layer L + 1 is
// what the processing of layer L enqueued.  The order INSIDE a layer is an accident of the queue history and cannot be
// reproduced by a parallel machine; the layers can.  One level here = one layer:
//   phase A (pull, :349-367)  every frontier voxel looks at its 24 stencil neighbours IN THE FIELD AS THE LEVEL FOUND IT
//                             (nothing but the entries' own frontier tags is written during this phase, by plain stores) and remembers the best obstacle;
//   phase B (:369-391)        a voxel that improved stores its new obstacle and stays in the frontier (re-queued, :371);
//                             one that did not offers ITS obstacle to the 24 neighbours: a compare-and-swap minimum on the
//                             neighbour's word (d^2 is recomputed from the ids, exact int32); a neighbour that improves joins
//                             the next frontier -- once: bit 30 of the word (kAct) is the "already queued" mark, set by
//                             whoever improves the voxel first and cleared when the voxel is processed.
// The next frontier is compacted with a wave ballot + prefix count and ONE atomic per wave.  Two barriers per level.
// For the frontiers of a sensor frame (a few hundred voxels) the whole update is ONE launch of ONE work-group
// (k_level_run: the barriers are __syncthreads, ~2 memory latencies per level); wider levels run on the 32 CUs of one XCD
// behind flag barriers (k_level_grid, below: why one XCD, how the placement is checked, how every wait is bounded); a
// frontier beyond that is handed to the frontier rounds, or -- engine pinned -- goes on as a pair of launches per level over
// any number of work-groups (k_level_pull / k_level_push: the kernel boundary is the barrier).  The launches of an update
// are chained without host round trips: each finds out on the device whether there is anything for it.
//
// Deletes (:292-337).  The reference walks the vanished obstacle's list and re-seeds every member from its first valid
// neighbour; those re-seeded voxels then pull and push like everybody else.  Here the orphans are found by the scan of
// dense_map.hip: k_invalidate / hash_map.hip: k_h_invalidate, reset to "no obstacle" and put into level 0 as PULL-ONLY
// entries: a voxel without an obstacle has nothing to push, its first pull is its re-seed.
// Orphans OUTSIDE the update window (local maps) are the one place where the reference's list order shows: such a voxel is
// re-seeded from its first IN-WINDOW neighbour that is valid at that moment of the list walk -- a live obstacle, or an
// orphan walked earlier -- and one that finds nothing stays at infinity for good (never queued :329, never pushed into
// :378).  The list is push-front in adoption order, i.e. walked from the rim of the dead cell inwards; "walked earlier" is
// modelled as "orphan of another vanished obstacle, or of the same one and not closer to it" (k_level_outside).  An
// out-of-window orphan that passes keeps asking its in-window neighbours once per level (nobody pushes into it) until it
// holds a value, goes on while it improves, and is frozen after that -- as in the reference (:345-373).
//
// oracle/esdf_port.cpp: relax_levels() is the CPU model of exactly this schedule; tests/test_levelsync_model.py judges
// it against the order spread of the verbatim reference without a GPU.
#pragma once
#include "common.hpp"
#include "dense_map.hpp"
#include "hash_map.hpp"
#include "relax_kernels.hpp"

namespace fiesta {

// ---- control block of one map's level engine (device memory; a pinned host copy travels back once per chain) ----------
struct LevelCtl {
  uint32_t n[3];      // frontier lengths: level l reads n[l % 3], appends to n[(l + 1) % 3], clears n[(l + 2) % 3]
  uint32_t nwait[3];  // of those: out-of-window orphans that only wait (a frontier of nothing else is a finished update)
  uint32_t level;     // levels run so far in this update; the current frontier is list[level & 1]
  uint32_t overflow;  // an append did not fit its list: the frontier-round engine has to finish this update
  uint32_t nout;      // out-of-window orphans collected by the delete scan (list `outside`)
  uint32_t writes;    // voxel words replaced (statistics)
  uint32_t maxd2;     // largest squared distance adopted (feeds C_MAXD2, the bound of the delete scan)
  uint32_t work;      // levels that found a non-empty frontier
  uint32_t invalidated;  // orphans the delete scan found (statistics)
  uint32_t ticks;     // time inside k_level_run, 10 ns units (statistics)
  uint32_t phase[4];  // of that, thread 0's view: fetch + pull, barrier, push, append + barriers
  uint32_t items;     // frontier entries processed, summed over the levels (statistics)
  uint32_t peak;      // the largest frontier
  uint32_t trace[48]; // the first levels of the update: entries << 16 | duration in 10 ns units (statistics)
  // k_level_grid (one slot per launch of it within an update):
  uint32_t bar[8];    // arrivals at its barriers (monotonic: barrier k is over when the count reaches k x work-groups)
  uint32_t xccs[8];   // bit x: a participating work-group ran on XCD x (more than one bit: the launch leaves the field alone)
  uint32_t grid_refused;  // ... and says so here
  uint32_t grid_levels;   // levels k_level_grid ran (statistics)
};

struct LevelArgs {
  vox_t *coc;
  uint32_t *list[2];
  uint32_t *res;       // phase A's verdict per frontier entry
  uint32_t *outside;   // out-of-window orphans of the delete scan
  LevelCtl *ctl;
  LevelCtl *ctl_other; // the NEXT update's control block: k_level_run clears it (no memset on the next update's path)
  uint32_t cap;        // entries a list holds
  uint32_t single_cap; // k_level_run leaves a frontier larger than this to the multi-work-group kernels
  uint32_t level;      // which level this launch starts at / is (host-side count; the device checks); kLvAny: ask the device
  uint32_t grid_min;   // k_level_grid leaves a frontier of at most this many entries to k_level_run again
  uint32_t grid_max;   // ... and one of more than this many to the frontier rounds (an update that large is cheaper there)
  uint32_t items_max;  // ... as is an update that has processed more than this many frontier entries altogether
  uint32_t bar;        // k_level_grid: its slot in LevelCtl::bar / xccs
  uint32_t *flags;     // k_level_grid: one word per participating work-group (its latest barrier number), one 128-byte line
  uint32_t bar_base;   // k_level_grid: barrier numbers of this launch start above this (never reused: the flags are never reset)
  uint32_t spin_limit; // k_level_grid: polls a barrier waits before it gives the update up (0: at once -- tests of the recovery)
  unsigned long long *counters;
  int track;           // maintain counters[C_MAXD2]
};

// phase A's verdict, one word per frontier entry
constexpr uint32_t kLvNone = 0xFFFFFFFFu;   // nothing to do (no obstacle yet, inside the window: a push will find it)
constexpr uint32_t kLvWait = 0xFFFFFFFEu;   // no obstacle yet, OUTSIDE the window: ask again next level
constexpr uint32_t kLvAny = 0xFFFFFFFFu;    // LevelArgs::level: "whatever level the control block says"
constexpr uint32_t kLvPush = 0x40000000u;   // | id: did not improve, offers this obstacle to its neighbours
                                            // otherwise: the id it improved to

// ---- address spaces ---------------------------------------------------------------------------------------------------
// A frontier ENTRY names a voxel by its packed coordinates (x << 20 | y << 10 | z: local coordinates of a dense array of at
// most 1024 voxels per axis, window coordinates of the paged map), so decoding one needs no memory access and no division;
// its ADDRESS (index of its word) may need one load (the paged map's directory).  The steps are split so that a lane can
// request all its directory entries, then all its words, each as ONE batch of loads.
// `box` = update window intersected with the array: the voxels the propagation may read and write (VoxInRange :63-72).
struct LvBox {
  int x0, y0, z0;
  unsigned ex, ey, ez;  // extents - 1
  __device__ inline bool has(int x, int y, int z) const {
    return (unsigned)(x - x0) <= ex && (unsigned)(y - y0) <= ey && (unsigned)(z - z0) <= ez;
  }
};
inline LvBox lv_box(const Geom &g) {
  LvBox b;
  const int x0 = std::max(g.wx0, 0), y0 = std::max(g.wy0, 0), z0 = std::max(g.wz0, 0);
  const int x1 = std::min(g.wx1, g.nx - 1), y1 = std::min(g.wy1, g.ny - 1), z1 = std::min(g.wz1, g.nz - 1);
  if (x1 < x0 || y1 < y0 || z1 < z0) {  // an empty window: no voxel is in it
    b.x0 = b.y0 = b.z0 = 1 << 30, b.ex = b.ey = b.ez = 0;
  } else {
    b.x0 = x0, b.y0 = y0, b.z0 = z0, b.ex = (unsigned)(x1 - x0), b.ey = (unsigned)(y1 - y0), b.ez = (unsigned)(z1 - z0);
  }
  return b;
}
__host__ __device__ inline uint32_t lv_pack(int x, int y, int z) { return ((uint32_t)x << 20) | ((uint32_t)y << 10) | (uint32_t)z; }

// Dense array (at most 1024 voxels per axis: larger arrays keep to the other engines): address = linear index; ids are
// plain global coordinates (no wrap).
struct DenseSpace {
  Geom g;
  const uint32_t *occbits;
  LvBox box;
  static constexpr int kWrap = 0;
  __device__ inline void decode(uint32_t e, int &x, int &y, int &z) const { x = (int)(e >> 20) & 1023, y = (int)(e >> 10) & 1023, z = (int)e & 1023; }
  __device__ inline void coords(uint32_t a, int &x, int &y, int &z) const {  // of an ADDRESS
    z = (int)(a % (uint32_t)g.nz);
    const uint32_t r = a / (uint32_t)g.nz;
    y = (int)(r % (uint32_t)g.ny), x = (int)(r / (uint32_t)g.ny);
  }
  __device__ inline bool valid(int x, int y, int z) const { return box.has(x, y, z); }
  __device__ inline int32_t doff(int dx, int dy, int dz) const { return (dx * g.ny + dy) * g.nz + dz; }
  // the own word of voxel (x,y,z): a voxel that made it into a list exists
  __device__ inline int32_t page_self(bool live, int, int, int) const { return live ? 0 : -1; }
  __device__ inline uint32_t addr_self(int32_t, int x, int y, int z) const { return ((uint32_t)x * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nz + (uint32_t)z; }
  // a neighbour's word: step 1 its page (>= 0: it has a word), step 2 its address
  __device__ inline int32_t page(bool ok, int, int, int) const { return ok ? 0 : -1; }
  __device__ inline uint32_t addr(int32_t, uint32_t self, int32_t off, int, int, int) const { return self + (uint32_t)off; }
  __device__ inline bool resident(uint32_t) const { return true; }
  __device__ inline bool occupied(uint32_t, int x, int y, int z) const { return occ_test(occbits, g, x, y, z); }
  __device__ inline uint32_t tile_of(const TileGrid &tg, uint32_t, int x, int y, int z) const { return (uint32_t)tg.tile_of(x, y, z); }
  // is the obstacle that word `id` names (held by voxel x,y,z) still occupied?
  __device__ inline bool alive(int x, int y, int z, vox_t id) const {
    int cx, cy, cz;
    unpack_coc(0, x + g.gx0, y + g.gy0, z + g.gz0, id, cx, cy, cz);
    cx -= g.gx0, cy -= g.gy0, cz -= g.gz0;
    return g.in_grid(cx, cy, cz) && occ_test(occbits, g, cx, cy, cz);
  }
};
// Paged hash-block map: address = place in the page pool, through the directory; ids are map coordinates modulo 1024
// decoded relative to the voxel (wrap).
struct PagedSpace {
  Geom g;
  const uint32_t *occbits;
  const int32_t *dir;        // window tile -> page (-1: none)
  const int32_t *page_tile;  // page -> window tile (-1: parked)
  LvBox box;
  static constexpr int kWrap = 1;
  static constexpr int kWin = HashMap::kWin, kNTY = HashMap::kNTY, kNTZ = HashMap::kNTZ, kPageVox = HashMap::kPageVox;
  __device__ inline void decode(uint32_t e, int &x, int &y, int &z) const { x = (int)(e >> 20) & 1023, y = (int)(e >> 10) & 1023, z = (int)e & 1023; }
  __device__ inline void coords(uint32_t a, int &x, int &y, int &z) const {  // of an ADDRESS (one dependent load)
    const int t = page_tile[a / (uint32_t)kPageVox], off = (int)(a % (uint32_t)kPageVox);
    x = (t / (kNTY * kNTZ)) * 16 + (off >> 9);
    y = ((t / kNTZ) % kNTY) * 16 + ((off >> 5) & 15);
    z = (t % kNTZ) * 32 + (off & 31);
  }
  __device__ inline bool valid(int x, int y, int z) const { return box.has(x, y, z); }
  __device__ inline int32_t doff(int, int, int) const { return 0; }
  __device__ inline int32_t page_self(bool live, int x, int y, int z) const {
    const int32_t p = dir[live ? ((x >> 4) * kNTY + (y >> 4)) * kNTZ + (z >> 5) : 0];  // (the load is unconditional: no branch)
    return live ? p : -1;
  }
  __device__ inline uint32_t addr_self(int32_t p, int x, int y, int z) const {
    return (uint32_t)p * (uint32_t)kPageVox + (uint32_t)(((x & 15) * 16 + (y & 15)) * 32 + (z & 31));
  }
  __device__ inline int32_t page(bool ok, int x, int y, int z) const {
    const int32_t p = dir[ok ? ((x >> 4) * kNTY + (y >> 4)) * kNTZ + (z >> 5) : 0];
    return ok ? p : -1;
  }
  __device__ inline uint32_t addr(int32_t p, uint32_t, int32_t, int x, int y, int z) const { return addr_self(p, x, y, z); }
  __device__ inline bool resident(uint32_t a) const { return page_tile[a / (uint32_t)kPageVox] >= 0; }
  __device__ inline bool occupied(uint32_t a, int, int, int) const { return (occbits[a >> 5] >> (a & 31)) & 1u; }
  __device__ inline uint32_t tile_of(const TileGrid &, uint32_t a, int, int, int) const { return (uint32_t)page_tile[a / (uint32_t)kPageVox]; }
  __device__ inline bool alive(int x, int y, int z, vox_t id) const {
    int dx, dy, dz;
    coc_offset(1, x + g.gx0, y + g.gy0, z + g.gz0, id, dx, dy, dz);
    const int cx = x - dx, cy = y - dy, cz = z - dz;
    if ((unsigned)cx >= (unsigned)kWin || (unsigned)cy >= (unsigned)kWin || (unsigned)cz >= (unsigned)kWin) return false;
    const int32_t p = page(true, cx, cy, cz);
    if (p < 0) return false;  // (its page left the window with it, or never existed)
    const uint32_t ca = addr_self(p, cx, cy, cz);
    return (occbits[ca >> 5] >> (ca & 31)) & 1u;
  }
};

// ---- small device helpers ---------------------------------------------------------------------------------------------
// A field word as the L2 holds it now (k_level_run keeps running across levels: a plain load may be served from the CU's
// L1, which atomics of an earlier phase never refreshed).
template <bool COHERENT>
__device__ inline vox_t lv_load(const vox_t *p) {
  if (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}
// The field's read-modify-writes: ONE = issued by k_level_run (a single work-group, hence a single L2 for the whole
// update; on coarse-grained device memory both forms execute in the L2 -- same instruction, measured the same).
template <bool ONE>
__device__ inline vox_t lv_cas(vox_t *p, vox_t expect, vox_t want) {
  if (ONE) {
    __hip_atomic_compare_exchange_strong(p, &expect, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return expect;  // (the value found)
  }
  return atomicCAS(p, expect, want);
}
template <bool ONE>
__device__ inline void lv_and(vox_t *p, vox_t mask) {
  if (ONE)
    (void)__hip_atomic_fetch_and(p, mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else
    atomicAnd(p, mask);
}
// Appends `value` for every lane whose `pred` holds; ONE atomic per wave.  Returns false if the list was full.
__device__ inline bool lv_append(bool pred, uint32_t value, uint32_t *list, uint32_t *count, uint32_t cap) {
  const unsigned long long m = __ballot(pred);
  if (!m) return true;
  const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(m));
  base = __shfl(base, leader);
  const uint32_t at = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  if (pred && at < cap) list[at] = value;
  return base + (uint32_t)__popcll(m) <= cap;
}
// The same for up to NBITS values per lane: bit k of `mask` set = the lane appends value(k).  ONE atomic per wave (a scan that
// finds 10^5 orphans behind a deleted surface otherwise spends its time on that one hot counter).
template <int NBITS, class Value>
__device__ inline bool lv_append_many(uint32_t mask, Value value, uint32_t *list, uint32_t *count, uint32_t cap) {
  const uint32_t mine = (uint32_t)__popc(mask);
  if (!__ballot(mine != 0u)) return true;
  const int lane = threadIdx.x & 63;
  uint32_t incl = mine;  // inclusive prefix sum over the wave
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
    if (lane >= off) incl += up;
  }
  const uint32_t total = (uint32_t)__shfl((int)incl, 63);
  uint32_t base = 0;
  if (lane == 63) base = atomicAdd(count, total);
  base = (uint32_t)__shfl((int)base, 63);
  uint32_t at = base + incl - mine;
#pragma unroll
  for (int k = 0; k < NBITS; ++k) {
    const bool on = (mask >> k) & 1u;
    if (on && at < cap) list[at] = value(k);
    at += on ? 1u : 0u;
  }
  return base + total <= cap;
}
template <class S>
__device__ inline int32_t lv_d2(const S &sp, int x, int y, int z, vox_t id) {
  return dist2(S::kWrap, x + sp.g.gx0, y + sp.g.gy0, z + sp.g.gz0, id);
}
// what "no obstacle" is worth in a comparison (wrap maps never adopt a candidate at or beyond the reach of an id)
template <class S>
__device__ inline int32_t lv_inf(const S &) {
  return S::kWrap ? kD2Cap : kD2Inf;
}
// has_link (common.hpp) without short circuits: an obstacle, or the stale link of a reset voxel
__device__ inline bool lv_link(vox_t w) { return ((w & kNoCoc) == 0u) | (((w & kAct) == 0u) & ((w & kIdMask) != 0u)); }
template <class S>
__device__ inline int32_t lv_have(const S &sp, int x, int y, int z, vox_t w) {
  return (w & kNoCoc) ? lv_inf(sp) : lv_d2(sp, x, y, z, w & kIdMask);
}

// The 24 stencil offsets (include/parameters.h:54-68, the reference's order).  FOUR LANES SHARE A FRONTIER ENTRY, six
// directions each (lane & 3 = which quarter): the per-direction code exists six times, not twenty-four -- with everything
// unrolled for one lane the kernel was ~90 KB of instructions -- and a level's loads and atomics are spread over four
// times as many lanes.  Directions are per-lane data, so the four quarters run the same instructions.
#define FIESTA_LV_DX(DX, DY, DZ) DX,
#define FIESTA_LV_DY(DX, DY, DZ) DY,
#define FIESTA_LV_DZ(DX, DY, DZ) DZ,
__device__ const signed char kLvDx[24] = {FIESTA_STENCIL24(FIESTA_LV_DX)};
__device__ const signed char kLvDy[24] = {FIESTA_STENCIL24(FIESTA_LV_DY)};
__device__ const signed char kLvDz[24] = {FIESTA_STENCIL24(FIESTA_LV_DZ)};
struct LvDirs {
  int dx[6], dy[6], dz[6];
  int q;           // the lane's quarter: directions 6q .. 6q + 5
  template <class S>
  __device__ inline void init(const S &sp) {
    q = (int)(threadIdx.x & 3u);
    // The lane's six directions, each packed as (dx + 2) | (dy + 2) << 4 | (dz + 2) << 8 in a LITERAL: the quarter selects
    // between four literals per slot (v_cndmask with immediates).  A constexpr table indexed by 6 q + j became a table in
    // memory behind a tree of branches -- several hundred instructions and a chain of loads in front of every launch.
#define FIESTA_LV_PICK(A, B, C_, D) (q == 0 ? (A) : q == 1 ? (B) : q == 2 ? (C_) : (D))
    const int pk[6] = {FIESTA_LV_PICK(545, 529, 561, 544), FIESTA_LV_PICK(547, 563, 531, 548), FIESTA_LV_PICK(530, 274, 786, 514),
                       FIESTA_LV_PICK(562, 818, 306, 578), FIESTA_LV_PICK(290, 289, 291, 34), FIESTA_LV_PICK(802, 803, 801, 1058)};
#undef FIESTA_LV_PICK
#pragma unroll
    for (int j = 0; j < 6; ++j) dx[j] = (pk[j] & 15) - 2, dy[j] = ((pk[j] >> 4) & 15) - 2, dz[j] = (pk[j] >> 8) - 2;
    (void)sp;
  }
};

// A lane's share of one frontier entry's stencil, as it sits in registers between the two phases of a level.
struct LvItem {
  int x, y, z;
  uint32_t self;    // address of the entry's own word
  vox_t w;          // that word when phase A read it (kUnobserved: the entry names no voxel of the map)
  uint32_t an[6];   // addresses of the lane's six neighbours' words (valid where nb != kUnobserved)
  vox_t nb[6];      // their words when phase A read them; kUnobserved: no such voxel / outside the window / never observed
};

// Loads a lane's share of an entry's stencil: the directory entries as one batch, then the words as one batch.  No load
// sits behind a branch (a lane without that neighbour reads word 0 and drops the value): a conditional load costs an
// exec-mask region of five scalar instructions per direction, and the whole level is a few hundred instructions.
template <class S, bool COHERENT>
__device__ inline void lv_fetch(const S &sp, const vox_t *coc, uint32_t e, bool live, const LvDirs &dr, LvItem &it) {
  sp.decode(e, it.x, it.y, it.z);
  int32_t pg[6];
  const int32_t pself = sp.page_self(live, it.x, it.y, it.z);
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int ux = it.x + dr.dx[j], uy = it.y + dr.dy[j], uz = it.z + dr.dz[j];
    pg[j] = sp.page(live & sp.valid(ux, uy, uz), ux, uy, uz);
  }
  it.self = pself >= 0 ? sp.addr_self(pself, it.x, it.y, it.z) : 0u;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const uint32_t at = sp.addr(pg[j], it.self, sp.doff(dr.dx[j], dr.dy[j], dr.dz[j]), it.x + dr.dx[j], it.y + dr.dy[j], it.z + dr.dz[j]);
    it.an[j] = pg[j] >= 0 ? at : 0u;
    const vox_t w = lv_load<COHERENT>(coc + it.an[j]);
    it.nb[j] = pg[j] >= 0 ? w : kUnobserved;
  }
  const vox_t w = lv_load<COHERENT>(coc + it.self);
  it.w = pself >= 0 ? w : kUnobserved;
}

// ---- phase A for one frontier entry (the four lanes of its quad call this together; all four return the verdict) ---------
template <class S, bool ONE>
__device__ inline uint32_t lv_pull(const S &sp, vox_t *coc, const LvItem &it, const LvDirs &dr) {
  const vox_t w = it.w;
  // the entry is being processed: its "queued" mark goes (and with it whatever a reset word still carried)
  // (a plain store: during phase A nobody else writes this word -- a voxel is in the frontier once, the pushes wait behind
  //  the barrier -- and readers of this phase take it with or without the mark; one atomic less per entry on the lines
  //  the whole frontier hammers)
  if (dr.q == 0 && w != kUnobserved) {
    const vox_t plain = (w & kNoCoc) ? kInf : (w & ~kAct);
    if (plain != w) __hip_atomic_store(coc + it.self, plain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  const bool have = !(w & kNoCoc);
  int32_t best = lv_have(sp, it.x, it.y, it.z, w);
  vox_t bid = kNoCoc;
#pragma unroll
  for (int j = 0; j < 6; ++j) {  // (selects, no branches)
    const vox_t wn = it.nb[j];
    const vox_t id = wn & kIdMask;
    const int32_t d = lv_d2(sp, it.x, it.y, it.z, id);
    // a candidate: an obstacle, or the stale link of a voxel the local-map rule reset (:353 tests the id only)
    const bool take = lv_link(wn) & (d < best);
    best = take ? d : best;
    bid = take ? id : bid;
  }
  // the best of the four quarters; the lower quarter wins a tie (= the first in the reference's direction order)
#pragma unroll
  for (int m = 1; m <= 2; m <<= 1) {
    const int32_t ob = __shfl_xor(best, m);
    const vox_t oi = (vox_t)__shfl_xor((int)bid, m);
    const bool other_lower = (dr.q & m) != 0;
    if (ob < best || (ob == best && other_lower && !(oi & kNoCoc))) best = ob, bid = oi;
  }
  if (w == kUnobserved) return kLvNone;  // (not a voxel of the map any more: a page that left; never for dense maps)
  if (!(bid & kNoCoc)) return bid;
  if (have) return kLvPush | (w & kIdMask);
  return sp.valid(it.x, it.y, it.z) ? kLvNone : kLvWait;
}

// compare-and-swap minimum: voxel (address a, coordinates x,y,z) takes obstacle `id` at squared distance d iff that is
// strictly closer than what it holds (:357, :382).  `seen` is a recent value of the word.  Returns 0: no change, 1: changed
// and the voxel was not queued yet (the caller appends it), 2: changed, already queued.
template <class S, bool ONE>
__device__ inline int lv_min(const S &sp, vox_t *coc, uint32_t a, int x, int y, int z, vox_t id, int32_t d, vox_t seen) {
  for (;;) {
    if (seen == kUnobserved) return 0;  // never observed: propagation does not enter (:382, -10000 > tmp is false)
    if (!(d < lv_have(sp, x, y, z, seen))) return 0;
    const vox_t old = lv_cas<ONE>(coc + a, seen, id | kAct);
    if (old == seen) return ((seen & kAct) && !(seen & kNoCoc)) ? 2 : 1;
    seen = old;
  }
}

// ---- phase B for one frontier entry (again the whole quad) ---------------------------------------------------------------
// An entry that improved stores its new obstacle into its own word (quarter 0 does); one that did not offers its obstacle
// to its 24 neighbours, six per lane.  Either way: compare-and-swap minima, ALL issued before the first result is looked
// at (one atomic latency per level), losers retried one by one (rare), and the voxels that enter the next frontier
// appended with wave votes + one atomic per wave.  `put(at, entry)` stores into the next frontier.  Every lane of a wave
// calls this (dead lanes with verdict kLvNone).  Returns the number of words the lane replaced.
template <class S, bool ONE, class Put>
__device__ inline uint32_t lv_push(const S &sp, vox_t *coc, uint32_t e, const LvItem &it, const LvDirs &dr, uint32_t verdict,
                                   uint32_t *n_next, uint32_t *n_wait, uint32_t &maxd2, Put put) {
  const bool waits = verdict == kLvWait && dr.q == 0;
  const bool active = verdict != kLvNone && verdict != kLvWait;
  const bool pushes = active && (verdict & kLvPush) != 0;
  const bool improved = active && !pushes && dr.q == 0;
  const vox_t id = verdict & kIdMask;
  uint32_t tried = 0, queue = 0, wrote = 0;
  vox_t got[6];
  vox_t got_self = 0;
  // the voxel's offset from the obstacle: a neighbour's squared distance to it is d + 2 e.u + |e|^2 (e: the direction)
  int ux, uy, uz;
  coc_offset(S::kWrap, it.x + sp.g.gx0, it.y + sp.g.gy0, it.z + sp.g.gz0, id, ux, uy, uz);
  const int32_t dv = ux * ux + uy * uy + uz * uz;
  // -- issue.  What a word read during phase A looks like now, unless somebody's push of THIS phase got there first: every
  //    frontier entry cleared its own mark in phase A (a reset orphan became plain "no obstacle"), nobody set one.
  auto settled = [](vox_t w) { return (w & kNoCoc) ? ((w & kAct) ? kInf : w) : (w & ~kAct); };
  const vox_t expect = settled(it.w);
  if (improved) got_self = lv_cas<ONE>(coc + it.self, expect, id | kAct);
  auto dnb = [&](const int j) {  // the obstacle's squared distance from neighbour j
    return dv + 2 * (dr.dx[j] * ux + dr.dy[j] * uy + dr.dz[j] * uz) + (dr.dx[j] * dr.dx[j] + dr.dy[j] * dr.dy[j] + dr.dz[j] * dr.dz[j]);
  };
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    got[j] = 0;
    const int32_t dn = dnb(j);
    const int32_t hv = lv_have(sp, it.x + dr.dx[j], it.y + dr.dy[j], it.z + dr.dz[j], it.nb[j]);
    const bool want = pushes & (it.nb[j] != kUnobserved) & (!S::kWrap | (dn < kD2Cap)) & (dn < hv);  // (no short circuits)
    if (want) {
      tried |= 1u << j;
      got[j] = lv_cas<ONE>(coc + it.an[j], settled(it.nb[j]), id | kAct);
    }
  }
  // -- judge
  if (improved) {
    const int r = got_self == expect ? 1 : lv_min<S, ONE>(sp, coc, it.self, it.x, it.y, it.z, id, dv, got_self);
    // (r == 0: a push of this very phase got there first with something at least as close -- it set the mark and queued
    //  the voxel; r == 2: the same, and this store still improved on it)
    if (r) ++wrote, maxd2 = max(maxd2, (uint32_t)dv);
    if (r == 1) queue |= 1u << 6;  // re-queued (:371)
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    if (tried & (1u << j)) {
      const int32_t dn = dnb(j);
      const int r = got[j] == settled(it.nb[j])
                        ? 1
                        : lv_min<S, ONE>(sp, coc, it.an[j], it.x + dr.dx[j], it.y + dr.dy[j], it.z + dr.dz[j], id, dn, got[j]);
      if (r) ++wrote, maxd2 = max(maxd2, (uint32_t)dn);
      if (r == 1) queue |= 1u << j;
    }
  }
  if (waits) queue |= 1u << 6, atomicAdd(n_wait, 1u);
  // -- append: one vote per kind of entry (own voxel, direction 0..5); the votes and their running totals are wave-uniform
  //    (scalar registers), a lane's position is a vote's total so far + the lanes below it in that vote; ONE atomic per wave
  const int lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  unsigned long long vote[7];
  uint32_t total = 0;
#pragma unroll
  for (int b = 0; b < 7; ++b) {
    vote[b] = __ballot((queue >> b) & 1u);
    total += (uint32_t)__popcll(vote[b]);
  }
  if (total) {
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(n_next, total);
    base = (uint32_t)__shfl((int)base, 0);
#pragma unroll
    for (int b = 0; b < 7; ++b) {
      if (queue & (1u << b)) {
        const uint32_t at = base + (uint32_t)__popcll(vote[b] & below);
        if (b == 6)
          put(at, e);
        else
          put(at, lv_pack(it.x + dr.dx[b], it.y + dr.dy[b], it.z + dr.dz[b]));
      }
      base += (uint32_t)__popcll(vote[b]);
    }
  }
  return wrote;
}

// ---- seeding ----------------------------------------------------------------------------------------------------------
// Insert drain (:278-291): a queued voxel that is still occupied becomes its own obstacle and enters level 0.
template <class S>
__global__ void k_level_seed_insert(S sp, LevelArgs a, const uint32_t *ins, int64_t n) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  bool add = false;
  uint32_t e = 0;
  if (i < n) {
    const uint32_t at = ins[i];
    int x = 0, y = 0, z = 0;
    const bool here = sp.resident(at);
    if (here) sp.coords(at, x, y, z);
    if (here && sp.occupied(at, x, y, z)) {  // "Exist after a whole bunch of updates" (:282)
      const vox_t self = pack_coc(x + sp.g.gx0, y + sp.g.gy0, z + sp.g.gz0) | kAct;
      add = atomicExch(a.coc + at, self) != self;  // (a voxel may sit in the queue twice: inserted, deleted, inserted)
      e = lv_pack(x, y, z);
    }
  }
  if (!lv_append(add, e, a.list[0], &a.ctl->n[0], a.cap)) a.ctl->overflow = 1;
}

// Out-of-window orphans of the delete scan (their words are still untouched, the in-window orphans carry kReset | dead id):
// which of them would the reference's list walk have re-seeded?  See the header comment.
template <class S>
__global__ void k_level_outside(S sp, LevelArgs a) {
  const uint32_t n = min(a.ctl->nout, a.cap);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < (n + 63u) / 64u * 64u; i += gridDim.x * blockDim.x) {
    bool valid = false;
    uint32_t e = 0;
    if (i < n) {
      e = a.outside[i];
      int x, y, z;
      sp.decode(e, x, y, z);
      const uint32_t at = sp.addr_self(sp.page_self(true, x, y, z), x, y, z);
      const vox_t dead = a.coc[at] & kIdMask;
      int cx, cy, cz;  // the vanished obstacle, in the coordinates ids are made of
      unpack_coc(S::kWrap, x + sp.g.gx0, y + sp.g.gy0, z + sp.g.gz0, dead, cx, cy, cz);
      const int32_t own = lv_d2(sp, x, y, z, dead);
#pragma unroll
      for (int k = 0; k < 24; ++k) {
        const int ux = x + kLvDx[k], uy = y + kLvDy[k], uz = z + kLvDz[k];
        const int32_t pg = sp.page(!valid && sp.valid(ux, uy, uz), ux, uy, uz);
        if (pg >= 0) {
          const vox_t wn = a.coc[sp.addr(pg, at, sp.doff(kLvDx[k], kLvDy[k], kLvDz[k]), ux, uy, uz)];
          if (wn != kUnobserved) {
            if ((wn & kNoCoc) && (wn & kAct)) {  // an orphan the scan reset: walked before this one?
              const vox_t nd = wn & kIdMask;
              if (nd != dead) {
                valid = true;
              } else {
                const int ex = ux + sp.g.gx0 - cx, ey = uy + sp.g.gy0 - cy, ez = uz + sp.g.gz0 - cz;
                valid = ex * ex + ey * ey + ez * ez >= own;
              }
            } else if (has_link(wn)) {
              valid = sp.alive(ux, uy, uz, wn & kIdMask);
            }
          }
        }
      }
      a.coc[at] = valid ? kReset : kInf;
    }
    if (!lv_append(valid, e, a.list[0], &a.ctl->n[0], a.cap)) a.ctl->overflow = 1;
  }
}

// ---- the list walk of the delete drain, without the lists ---------------------------------------------------------------
// The reference re-seeds the orphans of a vanished obstacle while it walks that obstacle's list (:300-331): every orphan
// takes the obstacle of its FIRST neighbour (stencil order) that is valid at that moment -- and orphans walked earlier are
// valid again.  A list is push-front in adoption order, so it is walked from the voxels farthest from the obstacle to the
// nearest: the dead cell fills from its rim inwards before any propagation starts, and every re-seeded orphan is queued in
// layer 0 with the value it got.  Here (schedule 6 of the CPU model oracle/esdf_port.cpp): the orphans of
// ALL vanished obstacles at once, in shells of decreasing whole-voxel distance from their own obstacle, one shell after the
// other; inside a shell every orphan looks at the field as the shell (or its pass of 256 orphans) found it.  After the
// delete scan "valid" is simply "holds an obstacle": every voxel whose obstacle vanished has been reset.  Orphans that find
// nobody keep "no obstacle" and leave level 0 (a push will find them, :378).  ONE work-group, level 0 of at most kFillCap
// entries; a larger one keeps the orphans as pull-only entries of level 0 (the schedule of rounds 1-3 of this engine).
constexpr uint32_t kFillCap = 8192;
template <class S, int NT>
__global__ __launch_bounds__(NT) void k_level_fill(S sp, LevelArgs a) {
  __shared__ uint8_t s_shell[kFillCap];    // shell of entry i (255: not an orphan to fill)
  __shared__ uint16_t s_order[kFillCap];   // the orphans, sorted by shell (far shells first)
  __shared__ uint8_t s_keep[kFillCap];     // does entry i stay in level 0?
  __shared__ uint32_t s_count[64], s_start[64], s_fill[64], s_n, s_fmax;
  LevelCtl *ctl = a.ctl;
  const int tid = threadIdx.x;
  const uint32_t n = ctl->n[0];
  if (n == 0 || n > kFillCap || n > a.cap || ctl->overflow) return;
  if (tid == 0) s_fmax = 0;
  uint32_t fill_max = 0;  // largest d^2 an orphan adopted here: it may never improve in a level and so never be counted there
  uint32_t *list = a.list[0];
  if (tid < 64) s_count[tid] = 0, s_fill[tid] = 0;
  __syncthreads();
  for (uint32_t i = tid; i < n; i += NT) {
    int x, y, z;
    sp.decode(list[i], x, y, z);
    uint32_t shell = 255;
    const int32_t pg = sp.page_self(sp.valid(x, y, z), x, y, z);
    if (pg >= 0) {
      const vox_t w = a.coc[sp.addr_self(pg, x, y, z)];
      if ((w & kNoCoc) && (w & kAct) && (w & kIdMask) != 0u && w != kUnobserved) {  // kReset | the vanished obstacle's id
        const int32_t d2 = lv_d2(sp, x, y, z, w & kIdMask);
        int r = (int)sqrtf((float)d2);
        r += ((r + 1) * (r + 1) <= d2) ? 1 : 0, r -= (r * r > d2) ? 1 : 0;  // floor(sqrt(d2)), exactly
        shell = (uint32_t)min(r, 63);
        atomicAdd(&s_count[shell], 1u);
      }
    }
    s_shell[i] = (uint8_t)shell;
    s_keep[i] = shell == 255 ? 1 : 0;  // (seeds and waiting orphans outside the window stay; orphans: if they find somebody)
  }
  __syncthreads();
  if (tid == 0) {  // far shells first
    uint32_t at = 0;
    for (int sh = 63; sh >= 0; --sh) s_start[sh] = at, at += s_count[sh];
  }
  __syncthreads();
  for (uint32_t i = tid; i < n; i += NT) {
    const uint32_t sh = s_shell[i];
    if (sh != 255) s_order[s_start[sh] + atomicAdd(&s_fill[sh], 1u)] = (uint16_t)i;
  }
  __syncthreads();
  LvDirs dr;
  dr.init(sp);
  constexpr uint32_t QT = NT / 4;
  const int lane = tid & 63;
  for (int sh = 63; sh >= 0; --sh) {
    const uint32_t cnt = s_count[sh], first = s_start[sh];
    for (uint32_t base = 0; base < cnt; base += QT) {  // (wave-uniform bounds: the barriers below are reached by everybody)
      const uint32_t k = base + ((uint32_t)tid >> 2);
      const bool live = k < cnt;
      const uint32_t idx = live ? s_order[first + k] : 0u;
      LvItem it;
      uint32_t key = 99;   // the lane's first valid direction: 6 q + j
      vox_t got = kNoCoc;
      if (__ballot(live)) {
        lv_fetch<S, true>(sp, a.coc, live ? list[idx] : 0u, live, dr, it);
#pragma unroll
        for (int j = 5; j >= 0; --j) {
          const vox_t wn = it.nb[j];
          bool ok = wn != kUnobserved && !(wn & kNoCoc);
          // (the stale link of a voxel the local-map rule reset names an obstacle that may still stand, :308)
          if (wn != kUnobserved && (wn & kNoCoc) && !(wn & kAct) && (wn & kIdMask) != 0u)
            ok = sp.alive(it.x + dr.dx[j], it.y + dr.dy[j], it.z + dr.dz[j], wn & kIdMask);
          if (ok) key = (uint32_t)(6 * dr.q + j), got = wn & kIdMask;
        }
      }
      // the first valid direction of the quad (the reference's stencil order), and the obstacle behind it
      uint32_t best = key;
      best = min(best, (uint32_t)__shfl_xor((int)best, 1));
      best = min(best, (uint32_t)__shfl_xor((int)best, 2));
      const vox_t id = (vox_t)__shfl((int)got, (lane & ~3) | (int)min(best / 6u, 3u));
      __syncthreads();  // every orphan of this pass has looked
      if (live && dr.q == 0) {
        if (best != 99u) {
          __hip_atomic_store(a.coc + it.self, id | kAct, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          s_keep[idx] = 1;
          fill_max = max(fill_max, (uint32_t)lv_d2(sp, it.x, it.y, it.z, id));
        }
      }
      __syncthreads();  // ... and every re-seeded one is in the field before the next pass looks
    }
  }
  // orphans that found nobody: plain "no obstacle", out of level 0; the others move up
  if (tid == 0) s_n = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n; base += NT) {  // (order inside level 0 does not matter: wave votes, as everywhere)
    const uint32_t i = base + tid;
    const bool in = i < n;
    const uint32_t e = in ? list[i] : 0u;
    const bool keep = in && s_keep[i];
    if (in && !keep) {
      int x, y, z;
      sp.decode(e, x, y, z);
      const int32_t pg = sp.page_self(true, x, y, z);
      if (pg >= 0) __hip_atomic_store(a.coc + sp.addr_self(pg, x, y, z), kInf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();  // (every entry of this stretch has been read before any is overwritten)
    const unsigned long long m = __ballot(keep);
    uint32_t at = 0;
    if (lane == 0 && m) at = atomicAdd(&s_n, (uint32_t)__popcll(m));
    at = (uint32_t)__shfl((int)at, 0) + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (keep) list[at] = e;  // (at < base + NT: a kept entry never lands on one that is still to be read -- see the barrier above)
  }
  if (fill_max) atomicMax(&s_fmax, fill_max);
  __syncthreads();
  if (tid == 0) {
    ctl->n[0] = s_n;
    // (the bounded delete scan reaches ceil(sqrt(C_MAXD2)) + 1 voxels; a filled orphan that only pushes from
    //  here on must be inside that bound, or a later delete of its obstacle leaves it with a stale link)
    if (s_fmax > ctl->maxd2) ctl->maxd2 = s_fmax;
    if (a.track && s_fmax) atomicMax(&a.counters[C_MAXD2], (unsigned long long)s_fmax);
  }
}

// ---- the levels -------------------------------------------------------------------------------------------------------
// ONE work-group runs levels until the frontier is empty (or only waits), outgrows `single_cap`, or a list overflows.
// The frontier lives in LDS (entries beyond the LDS lists' size go to the global list at the same position: the launch
// that takes over finds them there), phase A's verdicts too; a level of at most NT entries keeps every entry's stencil in
// registers from phase A to phase B -- what a level costs then is one batch of loads and one batch of atomics.
template <class S, int NT, int CAP>
__global__ __launch_bounds__(NT) void k_level_run(S sp, LevelArgs a) {
  __shared__ uint32_t s_list[2][CAP];
  __shared__ vox_t s_nb[14][NT];     // a level of more than NT / 4 entries: the two entries of every quad between the phases
  __shared__ uint32_t s_an[14][NT];  // (per entry: the lane's six neighbour words / addresses, then the own word / address)
  __shared__ uint32_t s_next, s_nextwait, s_wrote, s_maxd2, s_trace_n;
  LevelCtl *ctl = a.ctl;
  const int tid = threadIdx.x;
  const unsigned long long t_in = wall_clock64();
  if (tid == 0) {  // (what the host used to do with two small launches of its own)
    *a.ctl_other = LevelCtl{};
    a.counters[C_INSERT] = 0, a.counters[C_DELETE] = 0;  // both queues are drained (the seeds ran before this launch)
  }
  // (the host names the level this launch starts at: the counters and the list are then requested together, one memory
  //  latency in front of the first level instead of three)
  uint32_t level = a.level;
  if (level == kLvAny) level = ctl->level;  // (a launch behind k_level_grid: the host does not know how far that got)
  const uint32_t first = (uint32_t)tid < (uint32_t)CAP ? a.list[level & 1u][tid] : 0u;  // (a list holds far more than CAP entries)
  uint32_t n = ctl->n[level % 3u], nwait = ctl->nwait[level % 3u], work = 0;
  const uint32_t found_overflow = ctl->overflow;
  if (found_overflow) return;  // (the update was given up ahead of this launch: the host picks up the pieces)
  bool fits = true;
  if (tid == 0) s_wrote = 0, s_maxd2 = 0;
  bool in_lds = n <= (uint32_t)CAP;  // the current frontier is (also) in s_list
  static_assert(CAP <= NT, "one entry per thread loads the list");
  if (in_lds && (uint32_t)tid < n) s_list[level & 1u][tid] = first;
  __syncthreads();
  uint32_t wrote = 0, maxd2 = 0;
  // (an update is over after at most a few thousand levels -- the longest chain of voxels the grid holds; the bound on the
  //  loop only keeps a damaged field from hanging the device: the rounds take over)
  LvDirs dr;
  dr.init(sp);
  constexpr uint32_t QT = NT / 4;  // entries a pass of the work-group covers
  while (fits && n != 0 && n != nwait && n <= a.single_cap && n <= (uint32_t)CAP && work < 65536u) {
    const uint32_t *in = s_list[level & 1u];
    uint32_t *out = s_list[(level + 1u) & 1u];
    uint32_t *gout = a.list[(level + 1u) & 1u];
    if (tid == 0) s_next = 0, s_nextwait = 0;
    auto put = [&](uint32_t at, uint32_t e) {
      if (at < (uint32_t)CAP)
        out[at] = e;
      else if (at < a.cap)
        gout[at] = e;
    };
    if (tid == 0) ctl->items += n, ctl->peak = max(ctl->peak, n), s_trace_n = n;
    const unsigned long long p0 = wall_clock64();
    unsigned long long p1 = p0, p2 = p0, p3 = p0;
    if (n <= QT) {
      LvItem it;
      const uint32_t i = (uint32_t)tid >> 2;
      const bool live = i < n;
      const uint32_t e = live ? in[i] : 0u;
      uint32_t verdict = kLvNone;
      if (__ballot(live)) {  // (a wave without an entry goes straight to the barrier: at a handful of entries most do)
        lv_fetch<S, true>(sp, a.coc, e, live, dr, it);
        verdict = lv_pull<S, true>(sp, a.coc, it, dr);
      }
      if (verdict == 0x12345678u) ++wrote;  // (keeps the verdict -- and the loads behind it -- ahead of the clock read)
      p1 = wall_clock64();
      __syncthreads();  // every pull has read the field; s_next is reset
      p2 = wall_clock64();
      if (__ballot(live && verdict != kLvNone)) wrote += lv_push<S, true>(sp, a.coc, e, it, dr, live ? verdict : kLvNone, &s_next, &s_nextwait, maxd2, put);
      p3 = wall_clock64();
    } else {
      // QT < n <= CAP = 2 QT: every quad has TWO entries.  Both stencils are requested together (one memory latency for the
      // level's pulls, as above); the first stays in registers across the barrier, the second waits in LDS.
      static_assert(CAP == 2 * (int)QT, "two entries per quad");
      const uint32_t i0 = (uint32_t)tid >> 2, i1 = i0 + QT;
      const bool live1 = i1 < n;
      const uint32_t e0 = in[i0], e1 = live1 ? in[i1] : 0u;
      uint32_t v0, v1;
      {
        LvItem it0, it1;
        lv_fetch<S, true>(sp, a.coc, e0, true, dr, it0);
        lv_fetch<S, true>(sp, a.coc, e1, live1, dr, it1);
        auto park = [&](const LvItem &it, const int k) {
#pragma unroll
          for (int j = 0; j < 6; ++j) s_nb[7 * k + j][tid] = it.nb[j], s_an[7 * k + j][tid] = it.an[j];
          s_nb[7 * k + 6][tid] = it.w, s_an[7 * k + 6][tid] = it.self;
        };
        v0 = lv_pull<S, true>(sp, a.coc, it0, dr);
        park(it0, 0);
        v1 = lv_pull<S, true>(sp, a.coc, it1, dr);
        park(it1, 1);
      }
      __syncthreads();  // every pull has read the field
      auto unpark = [&](LvItem &it, const uint32_t e, const int k) {
        sp.decode(e, it.x, it.y, it.z);
#pragma unroll
        for (int j = 0; j < 6; ++j) it.nb[j] = s_nb[7 * k + j][tid], it.an[j] = s_an[7 * k + j][tid];
        it.w = s_nb[7 * k + 6][tid], it.self = s_an[7 * k + 6][tid];
      };
      {
        LvItem it;
        unpark(it, e0, 0);
        wrote += lv_push<S, true>(sp, a.coc, e0, it, dr, v0, &s_next, &s_nextwait, maxd2, put);
      }
      if (__ballot(live1 && v1 != kLvNone)) {
        LvItem it;
        unpark(it, e1, 1);
        wrote += lv_push<S, true>(sp, a.coc, e1, it, dr, live1 ? v1 : kLvNone, &s_next, &s_nextwait, maxd2, put);
      }
    }
    __syncthreads();
    n = s_next, nwait = s_nextwait;
    if (n > a.cap) fits = false;
    ++level, ++work;
    __syncthreads();  // (s_next is reset at the top of the next level)
    if (tid == 0) {
      const unsigned long long p4 = wall_clock64();
      ctl->phase[0] += (uint32_t)(p1 - p0), ctl->phase[1] += (uint32_t)(p2 - p1), ctl->phase[2] += (uint32_t)(p3 - p2), ctl->phase[3] += (uint32_t)(p4 - p3);
      if (level <= 48u) ctl->trace[level - 1u] = (s_trace_n << 16) | (uint32_t)min((unsigned long long)0xFFFFu, p4 - p0);
    }
  }
  // hand the state back: the next launch (or the host) goes on from here
  if (in_lds && work && n <= a.cap)
    for (uint32_t i = tid; i < min(n, (uint32_t)CAP); i += NT) a.list[level & 1u][i] = s_list[level & 1u][i];
  atomicAdd(&s_wrote, wrote);
  atomicMax(&s_maxd2, maxd2);
  __syncthreads();
  if (tid == 0) {
    ctl->level = level;
    ctl->n[level % 3u] = min(n, a.cap), ctl->nwait[level % 3u] = nwait;
    ctl->n[(level + 1u) % 3u] = 0, ctl->nwait[(level + 1u) % 3u] = 0;
    ctl->n[(level + 2u) % 3u] = 0, ctl->nwait[(level + 2u) % 3u] = 0;
    if (!found_overflow && (!fits || work >= 65536u)) ctl->overflow = 1;
    ctl->writes += s_wrote;
    ctl->work += work;
    if (s_maxd2 > ctl->maxd2) ctl->maxd2 = s_maxd2;
    if (a.track && s_maxd2) atomicMax(&a.counters[C_MAXD2], (unsigned long long)s_maxd2);
    ctl->ticks += (uint32_t)(wall_clock64() - t_in);
  }
}

// ---- wide levels: one launch, many work-groups, barriers between them --------------------------------------------------
// k_level_run is one work-group on one CU: a level of n entries costs it ~2.5 us + n x 40 ns (every lane's neighbour word
// is a memory request of its own, and they all leave through one CU).  Here the same two phases run on up to 32 CUs,
// with a barrier among the work-groups where k_level_run has __syncthreads.  What makes that
// affordable is that all participating work-groups share ONE L2: the XCDs' L2s are not coherent with each other, so a
// barrier across XCDs would need an agent-scope release + acquire on every CU per phase (~3.5 us each, twice a level);
// inside one XCD the field is coherent through the L2 for the loads that bypass the L1 (lv_load<true>) and the
// read-modify-writes the phases already use.  Placement is not promised by HIP, so it is CHECKED: the launch has 8 x G
// work-groups, those with blockIdx % 8 == 0 take part (observed: they share an XCD), each records the XCD it runs on
// (HW_REG_XCC_ID) before the first barrier -- through read-modify-writes, which are coherent anywhere -- and if more than
// one XCD shows up every work-group leaves before touching the field: k_level_run (launched behind this kernel anyway)
// goes on alone and the host stops asking for the grid.  A different placement changes the speed, never the result.
// Every wait is bounded: a barrier that is not complete after `spin_limit` polls sets overflow = 2, all work-groups
// leave, and the host rebuilds the frontier-round engine's state from the tags in the field (LevelEngine::kAbort).
struct LvGridSync {
  uint32_t *count;     // LevelCtl::bar[slot]: arrivals at the FIRST barrier (read-modify-writes: coherent on any placement)
  uint32_t *overflow;  // LevelCtl::overflow
  uint32_t *flags;     // all later barriers: work-group g stores the barrier's number into flags[g], wave 0 of everybody reads
  uint32_t base;       //   the 32 words with one load per poll.  Plain stores stay in the XCD's L2 and loads that bypass the L1
  uint32_t g, groups, limit, done;  // are served from it: ~0.7 us a barrier where a counter of atomics costs ~2.5 (an atomic
  uint32_t *s_ok;      //   drops its line from the L2, so every poll behind one goes to memory).  Same-XCD only -- checked first.
  __device__ inline bool give_up() {
    (void)__hip_atomic_fetch_max(overflow, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
  }
  // Every memory operation of the calling work-group is complete when its arrival is published (__syncthreads waits for
  // each wave's outstanding loads, stores and atomics).  Both forms return false if the update is being given up.
  __device__ inline bool first() {
    __syncthreads();
    if (threadIdx.x == 0) {
      (void)__hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      uint32_t polls = 0;
      bool ok = true;
      while (__hip_atomic_fetch_add(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < groups) {
        if (polls++ >= limit || __hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
          ok = give_up();
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      *s_ok = ok ? 1u : 0u;
    }
    __syncthreads();
    return *s_ok != 0u;
  }
  __device__ inline bool operator()() {
    __syncthreads();
    const uint32_t target = base + ++done;
    if (threadIdx.x < 64u) {
      const uint32_t lane = threadIdx.x;
      if (lane == 0) __hip_atomic_store(flags + g, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      uint32_t polls = 0;
      bool ok = true;
      for (;;) {
        // lanes 0 .. groups-1: the flags; lane 32: has somebody given up?
        const uint32_t *p = lane < groups ? flags + lane : overflow;
        const uint32_t f = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__any(lane == 32u && f != 0u) || polls++ >= limit) {
          ok = give_up();
          break;
        }
        if (__all(lane >= groups || (int32_t)(f - target) >= 0)) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (lane == 0) *s_ok = ok ? 1u : 0u;
    }
    __syncthreads();
    return *s_ok != 0u;
  }
};

template <class S, int NT>
__global__ __launch_bounds__(NT) void k_level_grid(S sp, LevelArgs a) {
  if (blockIdx.x & 7u) return;
  __shared__ uint32_t s_ok, s_wrote, s_maxd2;
  LevelCtl *ctl = a.ctl;
  const int tid = threadIdx.x;
  const uint32_t g = blockIdx.x >> 3, G = gridDim.x >> 3;
  // the state the previous launch left (every work-group reads the same words: they all take the same way)
  uint32_t level = ctl->level;
  if (ctl->overflow || ctl->grid_refused) return;
  uint32_t n = ctl->n[level % 3u], nwait = ctl->nwait[level % 3u];
  if (n == 0 || n == nwait || n <= a.grid_min || n > a.grid_max) return;
  uint32_t items = ctl->items;  // (entries processed so far: only work-group 0 writes it, at the end of a level)
  if (items > a.items_max) return;
  const unsigned long long t_in = wall_clock64();
  LvGridSync sync{&ctl->bar[a.bar], &ctl->overflow, a.flags, a.bar_base, g, G, a.spin_limit, 0u, &s_ok};
  if (tid == 0) {
    s_wrote = 0, s_maxd2 = 0;
    const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;  // hwreg(HW_REG_XCC_ID, 0, 4)
    (void)__hip_atomic_fetch_or(&ctl->xccs[a.bar], 1u << xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (!sync.first()) return;
  {  // (a read-modify-write reads the word where the other XCDs' read-modify-writes went)
    const uint32_t seen = __hip_atomic_fetch_or(&ctl->xccs[a.bar], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__popc(seen) != 1) {
      if (g == 0 && tid == 0) ctl->grid_refused = 1;
      return;
    }
  }
  LvDirs dr;
  dr.init(sp);
  constexpr uint32_t QT = NT / 4;  // entries a pass of one work-group covers
  uint32_t wrote = 0, maxd2 = 0, work = 0;
  bool fits = true;
  for (;;) {
    const uint32_t *in = a.list[level & 1u];
    uint32_t *out = a.list[(level + 1u) & 1u];
    uint32_t *n_next = &ctl->n[(level + 1u) % 3u], *w_next = &ctl->nwait[(level + 1u) % 3u];
    auto put = [&](uint32_t at, uint32_t e) {
      if (at < a.cap) out[at] = e;
    };
    auto clear_slot = [&]() {  // the counters read last a level ago and appended to next level (written where the appends will go)
      if (g == 0 && tid == 0) {
        __hip_atomic_store(&ctl->n[(level + 2u) % 3u], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ctl->nwait[(level + 2u) % 3u], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    };
    const unsigned long long p0 = wall_clock64();
    unsigned long long p1 = p0, p2 = p0, p3 = p0;
    // this work-group's share: whole waves (16 entries), the same for everybody
    const uint32_t per = ((n + G - 1u) / G + 15u) & ~15u;
    const uint32_t lo = min(n, g * per), hi = min(n, lo + per);
    bool ok;
    if (per <= QT) {  // one pass: the stencil stays in registers between the phases
      LvItem it;
      const uint32_t i = lo + ((uint32_t)tid >> 2);
      const bool live = i < hi;
      const uint32_t e = live ? lv_load<true>(in + i) : 0u;
      uint32_t verdict = kLvNone;
      if (__ballot(live)) {
        lv_fetch<S, true>(sp, a.coc, e, live, dr, it);
        verdict = lv_pull<S, false>(sp, a.coc, it, dr);
      }
      if (verdict == 0x12345678u) ++wrote;  // (keeps the verdict -- and the loads behind it -- ahead of the clock read)
      p1 = wall_clock64();
      ok = sync();  // every pull has read the field
      p2 = wall_clock64();
      if (ok) {
        clear_slot();
        if (__ballot(live && verdict != kLvNone)) wrote += lv_push<S, false>(sp, a.coc, e, it, dr, live ? verdict : kLvNone, n_next, w_next, maxd2, put);
      }
      p3 = wall_clock64();
    } else {  // several passes: the verdicts wait in memory, the stencil is fetched again (as k_level_pull / k_level_push do)
      for (uint32_t i = lo + ((uint32_t)tid >> 2); i < lo + per; i += QT) {
        LvItem it;
        const bool live = i < hi;
        if (!__ballot(live)) continue;
        lv_fetch<S, true>(sp, a.coc, live ? lv_load<true>(in + i) : 0u, live, dr, it);
        const uint32_t verdict = lv_pull<S, false>(sp, a.coc, it, dr);
        if (live && dr.q == 0) a.res[i] = verdict;
      }
      ok = sync();
      if (ok) {
        clear_slot();
        for (uint32_t i = lo + ((uint32_t)tid >> 2); i < lo + per; i += QT) {
          const bool live = i < hi;
          const uint32_t verdict = live ? lv_load<true>(a.res + i) : kLvNone;
          if (!__ballot(verdict != kLvNone)) continue;
          LvItem it;
          const uint32_t e = live ? lv_load<true>(in + i) : 0u;
          lv_fetch<S, true>(sp, a.coc, e, live && verdict != kLvNone && verdict != kLvWait, dr, it);
          wrote += lv_push<S, false>(sp, a.coc, e, it, dr, verdict, n_next, w_next, maxd2, put);
        }
      }
    }
    if (ok) ok = sync();  // every push has landed, every append is counted
    if (!ok) {
      fits = false;
      break;
    }
    const uint32_t n_was = n;
    n = __hip_atomic_load(n_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    nwait = __hip_atomic_load(w_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ++level, ++work;
    items += n_was;
    if (g == 0 && tid == 0) {  // the control block follows level by level: a launch that is given up leaves a consistent count
      ctl->level = level;
      ctl->items += n_was, ctl->peak = max(ctl->peak, n_was);
      const unsigned long long p4 = wall_clock64();  // (work-group 0's view, like k_level_run's: pull, barrier, push, barrier + counts)
      ctl->phase[0] += (uint32_t)(p1 - p0), ctl->phase[1] += (uint32_t)(p2 - p1), ctl->phase[2] += (uint32_t)(p3 - p2), ctl->phase[3] += (uint32_t)(p4 - p3);
      if (level <= 48u) ctl->trace[level - 1u] = (min(n_was, 0xFFFFu) << 16) | (uint32_t)min((unsigned long long)0xFFFFu, wall_clock64() - p0);
    }
    if (n > a.cap) {  // the list is full: the frontier-round engine has to finish (all work-groups read the same count)
      if (g == 0 && tid == 0) ctl->n[level % 3u] = a.cap, (void)__hip_atomic_fetch_max(&ctl->overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
    if (n == 0 || n == nwait || n <= a.grid_min || n > a.grid_max || items > a.items_max || work >= 65536u) break;
  }
  (void)fits;
  if (wrote) atomicAdd(&s_wrote, wrote);
  if (maxd2) atomicMax(&s_maxd2, maxd2);
  __syncthreads();
  if (tid == 0) {
    if (s_wrote) atomicAdd(&ctl->writes, s_wrote);
    if (s_maxd2) {
      atomicMax(&ctl->maxd2, s_maxd2);
      if (a.track) atomicMax(&a.counters[C_MAXD2], (unsigned long long)s_maxd2);
    }
    if (g == 0) {
      ctl->work += work, ctl->grid_levels += work;
      ctl->ticks += (uint32_t)(wall_clock64() - t_in);
    }
  }
}

// A level as two launches over any number of work-groups.  `a.level` is the level the host believes this launch to be; the
// device's own count decides (a launch that finds the update finished, or at another level, does nothing).
template <class S>
__global__ __launch_bounds__(256) void k_level_pull(S sp, LevelArgs a) {
  LevelCtl *ctl = a.ctl;
  const uint32_t level = ctl->level;
  if (level != a.level || ctl->overflow) return;
  const uint32_t n = ctl->n[level % 3u];
  if (n == 0 || n == ctl->nwait[level % 3u]) return;
  const uint32_t *in = a.list[level & 1u];
  if (blockIdx.x == 0 && threadIdx.x == 0) ctl->n[(level + 2u) % 3u] = 0, ctl->nwait[(level + 2u) % 3u] = 0;
  LvDirs dr;
  dr.init(sp);
  const uint32_t n_up = (n + 15u) / 16u * 16u;
  for (uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 2; i < n_up; i += (gridDim.x * blockDim.x) >> 2) {
    LvItem it;
    const bool live = i < n;
    lv_fetch<S, false>(sp, a.coc, live ? in[i] : 0u, live, dr, it);
    const uint32_t verdict = lv_pull<S, false>(sp, a.coc, it, dr);
    if (live && dr.q == 0) a.res[i] = verdict;
  }
}
template <class S>
__global__ __launch_bounds__(256) void k_level_push(S sp, LevelArgs a) {
  LevelCtl *ctl = a.ctl;
  const uint32_t level = ctl->level;
  if (level != a.level || ctl->overflow) return;
  const uint32_t n = ctl->n[level % 3u];
  if (n == 0 || n == ctl->nwait[level % 3u]) return;
  const uint32_t *in = a.list[level & 1u];
  uint32_t *out = a.list[(level + 1u) & 1u];
  uint32_t wrote = 0, maxd2 = 0;
  auto put = [&](uint32_t at, uint32_t e) {
    if (at < a.cap) out[at] = e;
  };
  LvDirs dr;
  dr.init(sp);
  const uint32_t n_up = (n + 15u) / 16u * 16u;
  for (uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 2; i < n_up; i += (gridDim.x * blockDim.x) >> 2) {
    const bool live = i < n;
    const uint32_t verdict = live ? a.res[i] : kLvNone;
    if (!__ballot(verdict != kLvNone)) continue;
    LvItem it;
    const uint32_t e = live ? in[i] : 0u;
    lv_fetch<S, false>(sp, a.coc, e, live && verdict != kLvNone && verdict != kLvWait, dr, it);
    wrote += lv_push<S, false>(sp, a.coc, e, it, dr, verdict, &ctl->n[(level + 1u) % 3u], &ctl->nwait[(level + 1u) % 3u], maxd2, put);
  }
  __shared__ uint32_t s_wrote, s_maxd2;
  if (threadIdx.x == 0) s_wrote = 0, s_maxd2 = 0;
  __syncthreads();
  if (wrote) atomicAdd(&s_wrote, wrote);
  if (maxd2) atomicMax(&s_maxd2, maxd2);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_wrote) atomicAdd(&ctl->writes, s_wrote);
    if (s_maxd2) {
      atomicMax(&ctl->maxd2, s_maxd2);
      if (a.track) atomicMax(&a.counters[C_MAXD2], (unsigned long long)s_maxd2);
    }
  }
}
// closes a level of the two-launch form (ONE thread): the counters move on
namespace {  // (this header is included by two translation units)
__global__ void k_level_next(LevelArgs a) {
  LevelCtl *ctl = a.ctl;
  const uint32_t level = ctl->level;
  if (level != a.level || ctl->overflow) return;
  const uint32_t n = ctl->n[level % 3u];
  if (n == 0 || n == ctl->nwait[level % 3u]) return;
  if (ctl->n[(level + 1u) % 3u] > a.cap) ctl->n[(level + 1u) % 3u] = a.cap, ctl->overflow = 1;
  ctl->level = level + 1u;
  ctl->work += 1u;
  ctl->items += n, ctl->peak = max(ctl->peak, n);
}
}  // namespace

// The update did not fit the level engine's lists: the frontier-round engine finishes it.  It starts from frontier TAGS
// in the field and a list of active tiles, so every tagged word gets its tile activated (and a reset word loses the dead
// id it still carried).  A tagged voxel that holds an obstacle may still owe its pull (see k_level_list_to_tiles): it takes
// it here, from the neighbours that are NOT tagged -- those will never push; the tagged ones push in the rounds.
template <class S>
__global__ void k_level_to_tiles(S sp, LevelArgs a, int64_t nvox, TileGrid tg, uint32_t *flag, uint32_t *list, unsigned long long *count) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvox; i += (int64_t)gridDim.x * blockDim.x) {
    const vox_t w = a.coc[i];
    if (w == kUnobserved || !(w & kAct)) continue;
    if (!sp.resident((uint32_t)i)) continue;
    if (w & kNoCoc) a.coc[i] = kReset;
    int x, y, z;
    sp.coords((uint32_t)i, x, y, z);
    if (!(w & kNoCoc)) {
      int32_t best = lv_d2(sp, x, y, z, w & kIdMask);
      vox_t bid = kNoCoc;
      for (int k = 0; k < 24; ++k) {
        const int ux = x + kLvDx[k], uy = y + kLvDy[k], uz = z + kLvDz[k];
        const int32_t pg = sp.page(sp.valid(ux, uy, uz), ux, uy, uz);
        if (pg < 0) continue;
        const vox_t wn = a.coc[sp.addr(pg, (uint32_t)i, sp.doff(kLvDx[k], kLvDy[k], kLvDz[k]), ux, uy, uz)];
        if (wn == kUnobserved || (wn & kAct) || !has_link(wn)) continue;
        const int32_t d = lv_d2(sp, x, y, z, wn & kIdMask);
        if (d < best) best = d, bid = wn & kIdMask;
      }
      if (!(bid & kNoCoc)) (void)lv_min<S, false>(sp, a.coc, (uint32_t)i, x, y, z, bid, best, a.coc[i]);
    }
    const uint32_t t = sp.tile_of(tg, (uint32_t)i, x, y, z);
    if (flag[t] == 0u) activate_tile(t, flag, list, count);
  }
}

// A level outgrew what one work-group should carry: the frontier-round engine goes on from the CURRENT frontier.  Its
// entries wear the tag the rounds start from (the "queued" mark) -- but the rounds only let a voxel PULL that has no
// obstacle when its tile is staged, and a frontier entry here may have just received its first obstacle from a push and
// still owe its pull (:349-367).  So the hand-over is: phase A of this level (k_level_pull: the verdicts), then this
// kernel: an entry that improved stores its obstacle; every entry gets its tag back and its tile activated.  An orphan
// outside the window that still waits gets its reset word back: the rounds' own pass for such voxels (k_reseed_outside)
// finds it.
template <class S>
__global__ void k_level_list_to_tiles(S sp, LevelArgs a, TileGrid tg, uint32_t *flag, uint32_t *list, unsigned long long *count) {
  const LevelCtl *ctl = a.ctl;
  const uint32_t level = ctl->level, n = min(ctl->n[level % 3u], a.cap);
  const uint32_t *in = a.list[level & 1u];
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    int x, y, z;
    sp.decode(in[i], x, y, z);
    const int32_t pg = sp.page_self(true, x, y, z);
    if (pg < 0) continue;
    const uint32_t at = sp.addr_self(pg, x, y, z);
    const uint32_t verdict = a.res[i];
    if (verdict == kLvWait) {
      a.coc[at] = kReset;
      continue;  // (outside the window: the rounds never stage it)
    }
    if (verdict != kLvNone && !(verdict & kLvPush)) {
      const int32_t dv = lv_d2(sp, x, y, z, verdict & kIdMask);
      (void)lv_min<S, false>(sp, a.coc, at, x, y, z, verdict & kIdMask, dv, a.coc[at]);
      if (a.track) atomicMax(&a.counters[C_MAXD2], (unsigned long long)dv);  // (the rounds that take over only see later writes)
    }
    if (!(a.coc[at] & kNoCoc)) atomicOr(a.coc + at, kAct);
    if (!sp.valid(x, y, z)) continue;
    const uint32_t t = sp.tile_of(tg, at, x, y, z);
    if (flag[t] == 0u) activate_tile(t, flag, list, count);
  }
}

// ---- host side: buffers and the launch sequence, shared by both map classes ------------------------------------------
struct LevelEngine {
  DevBuf<uint32_t> list[2], res, outside, flags;
  uint32_t grid_launches = 0;
  LevelCtl *ctl2 = nullptr;   // device: two blocks, alternating between updates (the idle one is cleared by k_level_run)
  LevelCtl *ctl = nullptr;    // the current update's
  LevelCtl *h_ctl = nullptr;  // pinned
  uint32_t serial = 0;
  uint32_t cap = 0;
  static constexpr uint32_t kSingleCap = 512;  // frontier one work-group keeps to itself: two entries per quad of lanes
  static constexpr uint32_t kInsertCap = 512;  // inserts of an update `auto` still gives to this engine.  (r05 tried 2048 so that config 4's
                                               // 648-insert frames would keep the FIFO layers: every one of them -- ~90 k voxel writes each -- outgrew
                                               // kItemsMax and was handed to the rounds anyway, 0.35 ms per update instead of 0.25.)
  static constexpr uint32_t kGridEnter = 192;  // ... and what it keeps while k_level_grid stands behind it (7-10 us a level there)
  static constexpr uint32_t kGridMin = 64;     // frontier k_level_grid gives back to the one work-group
  static constexpr uint32_t kGridMax = 4096;   // frontier beyond which an update goes to the frontier rounds (pinned to this engine:
                                               // on as pairs of launches over all CUs, which beat one XCD from ~10 k entries)
  static constexpr uint32_t kItemsMax = 16384; // frontier entries altogether beyond which an update goes to the frontier rounds (unless pinned)
  static constexpr uint32_t kTiny = 8;         // inserts + deletes of an update that is launched without the grid behind it
  enum Outcome { kDone = 0, kOverflow = 1, kHandOver = 2, kAbort = 3 };
  static constexpr int kNT = 1024;
  uint32_t grid_enter = kGridEnter, grid_min = kGridMin;  // (members so that a tuning build can sweep them)
  int grid_groups = 32;          // work-groups of k_level_grid that take part (one XCD has 32 CUs); 0: never launch it
  uint32_t spin_limit = 1u << 18;  // polls one of its barriers waits (~0.2 s) before the update is given up
  bool grid_ok = true;           // until a launch found its work-groups on more than one XCD
  ~LevelEngine() {
    if (ctl2) (void)hipFree(ctl2);
    if (h_ctl) (void)hipHostFree(h_ctl);
  }
  void ensure(uint32_t want, hipStream_t s) {
    if (!ctl2) {
      FIESTA_HIP_CHECK(hipMalloc((void **)&ctl2, 2 * sizeof(LevelCtl)));
      FIESTA_HIP_CHECK(hipMemsetAsync(ctl2, 0, 2 * sizeof(LevelCtl), s));
      FIESTA_HIP_CHECK(hipHostMalloc((void **)&h_ctl, sizeof(LevelCtl)));
      ctl = ctl2;
      flags.ensure_exact(64, s);
      FIESTA_HIP_CHECK(hipMemsetAsync(flags.p, 0, 64 * sizeof(uint32_t), s));
    }
    if (want <= cap) return;
    for (auto &l : list) l.ensure_exact(want, s);
    res.ensure_exact(want, s);
    outside.ensure_exact(want, s);
    cap = want;
  }
  LevelArgs args(vox_t *coc, unsigned long long *counters, bool track) const {
    LevelArgs a;
    a.coc = coc;
    a.list[0] = list[0].p, a.list[1] = list[1].p;
    a.res = res.p;
    a.outside = outside.p;
    a.ctl = ctl;
    a.ctl_other = ctl2 + ((ctl - ctl2) ^ 1);
    a.cap = cap;
    a.single_cap = kSingleCap;
    a.level = 0;
    a.grid_min = grid_min;
    a.grid_max = kGridMax;
    a.items_max = 0xFFFFFFFFu;
    a.bar = 0;
    a.flags = flags.p;
    a.bar_base = 0;
    a.spin_limit = spin_limit;
    a.counters = counters;
    a.track = track ? 1 : 0;
    return a;
  }
  // a new update: the block the previous update's k_level_run cleared
  void begin() { ctl = ctl2 + (++serial & 1u); }
  // Runs the levels.  One chain of launches = k_level_run (narrow levels) -> k_level_grid (wide ones, one XCD's CUs) ->
  // k_level_run (the tail), each of which finds out on the device whether there is anything for it; one host round trip per
  // chain, and a sensor frame's update is one chain.  `tiny`: a handful of seeds -- the first chain is k_level_run alone.
  // kDone: the update is finished.  kOverflow: a list overflowed, the caller rebuilds the rounds' state by a scan
  // (k_level_to_tiles).  kAbort: k_level_grid gave up at a barrier; the caller does both of the following.  kHandOver (only
  // if !wide, and only without the grid): a level outgrew the one work-group; the current frontier is in the list, the
  // caller passes it to the rounds (k_level_list_to_tiles).  wide, without the grid: such levels go on as pairs of launches.
  // `done` is recorded behind the last kernel of every chain: when run() returns it marks the end of the levels' device work.
  template <class S>
  Outcome run(const S &sp, LevelArgs a, hipStream_t s, hipEvent_t done, bool wide, bool tiny, int64_t *launches) {
    uint32_t slot = 0;
    bool first = true;
    a.items_max = wide ? 0xFFFFFFFFu : kItemsMax;
    for (;;) {
      const bool grid = grid_ok && grid_groups > 0 && slot < 8u && !(first && tiny);
      a.level = first ? 0u : kLvAny;
      a.single_cap = grid ? grid_enter : kSingleCap;
      hipLaunchKernelGGL((k_level_run<S, kNT, (int)kSingleCap>), dim3(1), dim3(kNT), 0, s, sp, a);
      ++*launches;
      if (grid) {
        a.level = kLvAny, a.bar = slot++;
        a.bar_base = (++grid_launches) << 18;  // (2^18 barriers a launch: 65536 levels is the bound of one update)
        hipLaunchKernelGGL((k_level_grid<S, kNT>), dim3(8 * grid_groups), dim3(kNT), 0, s, sp, a);
        a.single_cap = kSingleCap;
        hipLaunchKernelGGL((k_level_run<S, kNT, (int)kSingleCap>), dim3(1), dim3(kNT), 0, s, sp, a);
        *launches += 2;
      }
      first = false;
      FIESTA_HIP_CHECK(hipGetLastError());
      FIESTA_HIP_CHECK(hipEventRecord(done, s));
      FIESTA_HIP_CHECK(hipMemcpyAsync(h_ctl, ctl, sizeof(LevelCtl), hipMemcpyDeviceToHost, s));
      FIESTA_HIP_CHECK(hipStreamSynchronize(s));
      for (;;) {
        if (h_ctl->overflow == 2u) return kAbort;
        if (h_ctl->overflow) return kOverflow;
        if (h_ctl->grid_refused) grid_ok = false;
        const uint32_t l = h_ctl->level, n = h_ctl->n[l % 3u];
        if (n == 0 || n == h_ctl->nwait[l % 3u]) return kDone;
        a.level = l;
        if (h_ctl->items > a.items_max && n > grid_min) return kHandOver;  // (a large update after all: cheaper on the frontier rounds)
        if (n <= kSingleCap || (grid_ok && grid_groups > 0 && slot < 8u && n <= a.grid_max)) break;  // another chain
        if (!wide) return kHandOver;
        // a chain of wide levels; work-groups in proportion to the frontier this chain starts with
        const int blocks = (int)std::min<uint32_t>((n + 63u) / 64u * 2u, 16384u), chain = 8;  // (64 entries per work-group pass)
        for (int k = 0; k < chain; ++k) {
          a.level = l + (uint32_t)k;
          hipLaunchKernelGGL((k_level_pull<S>), dim3(blocks), dim3(256), 0, s, sp, a);
          hipLaunchKernelGGL((k_level_push<S>), dim3(blocks), dim3(256), 0, s, sp, a);
          hipLaunchKernelGGL(k_level_next, dim3(1), dim3(1), 0, s, a);
          *launches += 3;
        }
        FIESTA_HIP_CHECK(hipGetLastError());
        FIESTA_HIP_CHECK(hipEventRecord(done, s));
        FIESTA_HIP_CHECK(hipMemcpyAsync(h_ctl, ctl, sizeof(LevelCtl), hipMemcpyDeviceToHost, s));
        FIESTA_HIP_CHECK(hipStreamSynchronize(s));
      }
    }
  }
};

}  // namespace fiesta
