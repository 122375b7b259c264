// fiesta_amd/csrc/nn_core.hpp -- the CELL transform: UpdateESDF's fixed point on a fully observed map (the exact Euclidean
// feature transform of the occupied set) computed the way a SPARSE obstacle set wants it: every 8^3 cell
// of voxels gets the short list of obstacles that can be the nearest one of any of its voxels, and its 512 voxels take
// the minimum over that list.  No pass over the grid reads anything but the 1-bit occupancy map; the only per-voxel
// traffic is the 4-byte store of the result (src/ESDFMap.cpp:273-398 is what this replaces, as the envelope passes of
// ft_core.hpp do; which of the two transforms serves an update is dense_map.hip's choice).
//
// This header is the arithmetic both sides share: the gfx950 kernels (nn_kernels.hpp) and the host model the CPU tests
// drive (tests/cpp/nn_model.cpp against brute force / scipy).  Plain pointers, no HIP types.
//
//   cell          8 x 8 x 8 voxels, origin O = 8 (cx, cy, cz); voxel v = O + (x, y, z), 0 <= x, y, z <= 7
//   site          an occupied voxel s; p = s - O its offset from the cell origin
//   competitor t  the site nearest to the cell centre found in the 5^3 cells around the cell (any site would do: it
//                 only has to be SOME obstacle; the nearest one prunes best)
//   keep rule     s is dropped iff t is at least as near as s for EVERY voxel of the cell:
//                     |p - v|^2 >= |q - v|^2  for all v   <=>   |p|^2 - |q|^2 >= 14 * sum_a max(0, p_a - q_a)     (q = t - O)
//                 (the half-space test of the bisector of s and t against the cell's box; ties go to t, which carries
//                 the same distance -- ids are tie-equivalent).  Whatever survives is a superset of the
//                 cell's true winners.
//   search window every site that survives lies within R = |t - c| + 2 h of the centre c = O + 3.5 (h = the box's half
//                 diagonal, 3.5 sqrt 3): rows of cells whose nearest point is farther are never read.  In DOUBLED
//                 coordinates (e = 2 p - 7: integers) that is |e|^2 <= rad2.
//   key           for voxel v and list entry i:  ((|p|^2 - 2 v.p + 147) << 9) | i << 4  -- |v - s|^2 minus the voxel's own
//                 |v|^2 (the same for every entry), biased to stay non-negative (|v|^2 <= 147); the minimum over the
//                 list names the nearest site, smallest list index on ties.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FIESTA_NN_HD __host__ __device__
#else
#define FIESTA_NN_HD
#include <math.h>
#endif
// Small-integer products and the square roots of the window bounds.  On the device: 24-bit multiplies (v_mul_i32_i24 is full
// rate, v_mul_lo_u32 a quarter: every operand here is below 2^12) and the bare v_sqrt_f32 (1 ulp; the bounds carry margins
// of 0.01 and more, and only have to err on the wide side).  On the host: the plain operators.
#if defined(__HIP_DEVICE_COMPILE__)
#define FIESTA_NN_MUL(a, b) __mul24((a), (b))
#define FIESTA_NN_SQRT(x) __builtin_amdgcn_sqrtf(x)
#else
#define FIESTA_NN_MUL(a, b) ((a) * (b))
#define FIESTA_NN_SQRT(x) sqrtf(x)
#endif

namespace fiesta {
namespace nn {

constexpr int kB = 8;               // cell edge
constexpr int kCap = 30;            // entries a cell's list holds; a cell that needs more fails the transform
constexpr int kSH = 9;              // a key's distance part starts at this bit; bits 4..8: the list index (so key & 0x1F0 is the
                                    // byte offset of the winner's entry), bits 0..3: zero
constexpr int kBias = 3 * 7 * 7;    // |v|^2 of the voxel farthest from the cell origin
constexpr int kKfirst = 1;           // cells around the cell the competitor is first looked for in (any site will do: the nearest prunes best)
constexpr int kKmax = 7;            // the search window reaches this many cells (p in [-56, 63]: -2 p stays inside a signed byte)
constexpr int kStride = 128;        // dwords of a cell's record: [0] the number of entries, [1] the reach of the search window the list was
                                    // built with, in cells (what an incremental update marks dirty cells by), [4 + 4 i ..] entry i = (b, K, m, W)
constexpr int kRaw = 32;            // candidates a team's scratch holds between the sweep and the record
constexpr int kThin = 16;           // lists longer than this are thinned pairwise
constexpr uint32_t kPadK = 0xFFFFFFFFu;  // K of a padding entry (b = m = 0): never the minimum
constexpr int kWhySparse = 1, kWhyDense = 2, kWhyOpen = 3;  // why a cell got no list (record dword 1, bits 8..15; bits 16..: the window's reach)
constexpr int kNone = 0x7FFFFFFF;

// The REGION the transform works on: the occupancy bits it reads, in its own coordinates (sites, cells).  An unsharded map:
// the array itself.  A shard: its array grown by a margin towards the neighbour shards, cut out of the replica of the global
// bitmap on global multiples of 8 -- the array then lies somewhere inside the region (f*: the region coordinates of its
// voxel (0, 0, 0); a*: its extents), only the cells that hold a voxel of it get a list (l*), and a face of the region
// beyond which the global grid goes on is OPEN: a cell whose search window touches it cannot be served.
struct Geom {
  int nx, ny, nz;     // voxels of the region
  int ncx, ncy, ncz;  // cells = ceil(n / 8)
  int wx, wy, wz;     // region -> the coordinates a voxel word stores (the region's origin in the global grid)
  int open;           // open faces: 1 x-lo, 2 x-hi, 4 y-lo, 8 y-hi, 16 z-lo, 32 z-hi
  int lx0, lx1, ly0, ly1, lz0, lz1;  // cells that get a list (half-open ranges)
  int fx, fy, fz;     // region coordinates of the array's voxel (0, 0, 0)
  int ax, ay, az;     // the array's extents
  FIESTA_NN_HD inline bool big() const { return nx > 1024 || ny > 1024 || nz > 1024; }  // sites modulo 1024 (site_offset)
};
// a region that IS the array
FIESTA_NN_HD inline Geom whole_geom(int nx, int ny, int nz) {
  Geom g{};
  g.nx = nx, g.ny = ny, g.nz = nz;
  g.ncx = (nx + kB - 1) / kB, g.ncy = (ny + kB - 1) / kB, g.ncz = (nz + kB - 1) / kB;
  g.lx1 = g.ncx, g.ly1 = g.ncy, g.lz1 = g.ncz;
  g.ax = nx, g.ay = ny, g.az = nz;
  return g;
}
// The region of an array at l0[] (extents ln[]) of a global grid G[], grown by mc voxels towards the rest of the grid: cut
// on global multiples of 8, clipped to the grid; a face that does not reach the grid's is open.  rlo[]: its origin in the
// grid.  false: more than kRegionMax voxels along an axis.  (A region of more than 1024 voxels stores its sites modulo 1024:
// Geom::big; the words of its entries are global coordinates modulo 1024 either way.)
constexpr int kRegionMax = 1536;  // voxels of a region along an axis (sites modulo 1024: k_nn_cells carries 192 cells of a row)
FIESTA_NN_HD inline bool region_geom(const int *G, const int *l0, const int *ln, int mc, Geom &g, int *rlo) {
  int n[3], f[3], c0[3], c1[3], open = 0;
  for (int k = 0; k < 3; ++k) {
    int lo = l0[k] - mc;
    lo = (lo < 0 ? 0 : lo) & ~((k == 2 ? 32 : kB) - 1);  // (z: whole 32-bit words of the bitmap's rows, so that wide loads stay dword-aligned)
    int hi = (l0[k] + ln[k] + mc + kB - 1) & ~(kB - 1);  // (exclusive)
    if (hi > G[k]) hi = G[k];
    rlo[k] = lo, n[k] = hi - lo, f[k] = l0[k] - lo;
    if (lo > 0) open |= 1 << (2 * k);
    if (hi < G[k]) open |= 2 << (2 * k);
    c0[k] = f[k] / kB, c1[k] = (f[k] + ln[k] - 1) / kB + 1;
    if (n[k] > kRegionMax) return false;
  }
  g = whole_geom(n[0], n[1], n[2]);
  g.wx = rlo[0], g.wy = rlo[1], g.wz = rlo[2];
  g.open = open;
  g.lx0 = c0[0], g.lx1 = c1[0], g.ly0 = c0[1], g.ly1 = c1[1], g.lz0 = c0[2], g.lz1 = c1[2];
  g.fx = f[0], g.fy = f[1], g.fz = f[2];
  g.ax = ln[0], g.ay = ln[1], g.az = ln[2];
  return true;
}
// what build_list needs of it
struct Frame {
  int wx = 0, wy = 0, wz = 0, open = 0, ncx = 0, ncy = 0, ncz = 0;
};
FIESTA_NN_HD inline Frame frame_of(const Geom &g) { return Frame{g.wx, g.wy, g.wz, g.open, g.ncx, g.ncy, g.ncz}; }
// does the window of +-K cells around cell (cx, cy, cz) touch an open face?
FIESTA_NN_HD inline bool leaves_region(const Frame &f, int cx, int cy, int cz, int K) {
  return ((f.open & 1) && cx - K < 0) || ((f.open & 2) && cx + K >= f.ncx) || ((f.open & 4) && cy - K < 0) ||
         ((f.open & 8) && cy + K >= f.ncy) || ((f.open & 16) && cz - K < 0) || ((f.open & 32) && cz + K >= f.ncz);
}

// entry of a kept site (offsets p from the cell origin, each within [-8 kKmax, 8 kKmax + 7]), 16 bytes:
//   b  bytes (-2 py, -2 pz, 0, 0): what v_dot4_i32_i8 multiplies with the lane's (y, z, 0, 0)
//   K  ((|p|^2 + kBias) << kSH) | index << 4
//   m  (-2 px) << kSH: the step of a key from one x-slab of the cell to the next
//   W  the site's packed coordinates: the word a voxel stores (common.hpp: pack_coc of a plain-id map)
FIESTA_NN_HD inline uint32_t entry_b(int py, int pz) { return ((uint32_t)(-2 * py) & 255u) | (((uint32_t)(-2 * pz) & 255u) << 8); }
FIESTA_NN_HD inline uint32_t entry_k(int px, int py, int pz, int idx) {
  return ((uint32_t)(px * px + py * py + pz * pz + kBias) << kSH) | ((uint32_t)idx << 4);
}
FIESTA_NN_HD inline uint32_t entry_m(int px) { return (uint32_t)(-2 * px * (1 << kSH)); }
// the key of voxel (x, y, z) of the cell against an entry, as the kernel computes it
FIESTA_NN_HD inline uint32_t key_of(uint32_t b, uint32_t K, uint32_t m, int x, int y, int z) {
  const int m2y = (int)(int8_t)(b & 255u), m2z = (int)(int8_t)((b >> 8) & 255u);
  const int t = y * m2y + z * m2z;                           // -2 (y py + z pz)
  return (uint32_t)(t * (1 << kSH)) + K + (uint32_t)x * m;  // (two's complement: the sum is the non-negative key)
}

FIESTA_NN_HD inline void unpack_site(uint32_t w, int &x, int &y, int &z) {
  x = (int)((w >> 20) & 1023u), y = (int)((w >> 10) & 1023u), z = (int)(w & 1023u);
}
// A site's offset from a cell origin (ox, oy, oz).  A region of more than 1024 voxels along an axis (a config-5 shard: 1024
// owned + ghost layers + margin) stores its sites MODULO 1024 -- a site is only ever looked at from a cell within a search
// window of it (< 512 voxels), so the offset is the one residue in [-512, 512), exactly as the wrap maps' voxel words are
// decoded (common.hpp: coc_offset).
template <bool WRAP>
FIESTA_NN_HD inline void site_offset(uint32_t w, int ox, int oy, int oz, int &px, int &py, int &pz) {
  unpack_site(w, px, py, pz);
  px -= ox, py -= oy, pz -= oz;
  if (WRAP) px = ((px + 512) & 1023) - 512, py = ((py + 512) & 1023) - 512, pz = ((pz + 512) & 1023) - 512;
}

// |2 (s - c)|^2 of a site at offset p: the doubled offset from the cell centre is 2 p - 7 per axis
FIESTA_NN_HD inline int e2_of(int px, int py, int pz) {
  const int ex = 2 * px - 7, ey = 2 * py - 7, ez = 2 * pz - 7;
  return FIESTA_NN_MUL(ex, ex) + FIESTA_NN_MUL(ey, ey) + FIESTA_NN_MUL(ez, ez);
}
// (2 R)^2 rounded up: 2 R = |e_t| + 4 h, 4 h = 14 sqrt 3 = 24.2487
FIESTA_NN_HD inline int rad2_of(int e2) {
  const float r = FIESTA_NN_SQRT((float)e2) + 24.27f;
  return (int)(r * r) + 1;
}
// cells the window reaches along an axis: a site of the ball has 16 |d| - 7 <= 2 R
FIESTA_NN_HD inline int reach_of(int rad2) { return (int)((FIESTA_NN_SQRT((float)rad2) + 7.02f) * 0.0625f); }
// doubled gap between the centre and a cell d cells away along one axis
FIESTA_NN_HD inline int gap_of(int d) { return d ? 16 * (d < 0 ? -d : d) - 7 : 0; }

// is the site at offset p dominated by the one at q over the whole cell (q2 = |q|^2)?  Ties go to q.
FIESTA_NN_HD inline bool dominated(int px, int py, int pz, int qx, int qy, int qz, int q2) {
  const int ax = px - qx, ay = py - qy, az = pz - qz;
  return FIESTA_NN_MUL(px, px) + FIESTA_NN_MUL(py, py) + FIESTA_NN_MUL(pz, pz) - q2 >=
         FIESTA_NN_MUL(14, (ax > 0 ? ax : 0) + (ay > 0 ? ay : 0) + (az > 0 ? az : 0));
}
// a candidate between sweep and record: three signed bytes
FIESTA_NN_HD inline uint32_t pack_p(int px, int py, int pz) { return ((uint32_t)px & 255u) | (((uint32_t)py & 255u) << 8) | (((uint32_t)pz & 255u) << 16); }
FIESTA_NN_HD inline void unpack_p(uint32_t v, int &px, int &py, int &pz) {
  px = (int)(int8_t)(v & 255u), py = (int)(int8_t)((v >> 8) & 255u), pz = (int)(int8_t)((v >> 16) & 255u);
}

// Where the lists come from: the first-site table and the site array, through an accessor.  bounds(): the index range of
// the sites of cells z0..z1 of cell row (X, Y) -- empty for a row outside the map, z clamped into the row; site(): the
// packed site behind an index of such a range.  PlainSrc reads the two arrays as they lie in memory (the host model; the
// kernel's path for the rare cell whose window leaves the staged neighbourhood); k_nn_lists stages its work-group's
// neighbourhood in LDS and reads that (nn_kernels.hpp: StagedSrc, reach = kStageK = 4 cells).
template <bool WRAP>
struct PlainSrcT {
  static constexpr bool wrap = WRAP;     // sites are stored modulo 1024 (site_offset)
  static constexpr int reach = 1 << 20;  // cells from the asking cell this source can serve
  const uint32_t *ctab, *sites;
  int ncx, ncy, ncz;
  FIESTA_NN_HD inline void bounds(int X, int Y, int z0, int z1, uint32_t &i0, uint32_t &i1) const {
    i0 = i1 = 0;
    if ((unsigned)X >= (unsigned)ncx || (unsigned)Y >= (unsigned)ncy) return;
    const uint32_t *row = ctab + ((int64_t)X * ncy + Y) * (ncz + 1);
    i0 = row[z0 < 0 ? 0 : z0], i1 = row[(z1 > ncz - 1 ? ncz - 1 : z1) + 1];
  }
  FIESTA_NN_HD inline uint32_t site(uint32_t i) const { return sites[i]; }
};
typedef PlainSrcT<false> PlainSrc;

// Who builds a list: ONE lane on the host model, a TEAM of four adjacent lanes in k_nn_lists (the rows of the search window
// dealt out among them: four times the waves in flight for the same work -- the sweeps are chains of dependent reads).
//   lanes, rank   size of the team, this lane's place in it
//   nearest()     the team's minimum of (e2, w) pairs, smaller e2 first, then smaller w (called by every lane, outside loops)
//   restart()     forget the slots handed out (every lane, outside loops)
//   slot()        the next free slot (any lane, inside loops)
//   count()       slots handed out (called by every lane once all have finished their loops)
//   count_now()   slots handed out so far, as this lane sees them (inside loops; entries below it may still be on their way)
//   put(), get()  the team's scratch of kRaw words: the candidates between the sweep and the record
struct Solo {
  static constexpr int lanes = 1;
  int rank = 0, n = 0;
  uint32_t raw[kRaw];
  FIESTA_NN_HD inline void nearest(int &, uint32_t &) const {}
  FIESTA_NN_HD inline void restart() { n = 0; }
  FIESTA_NN_HD inline int slot() { return n++; }
  FIESTA_NN_HD inline int count() const { return n; }
  FIESTA_NN_HD inline int count_now() const { return n; }
  FIESTA_NN_HD inline void put(int k, uint32_t v) { raw[k] = v; }
  FIESTA_NN_HD inline uint32_t get(int k) const { return raw[k]; }
};

// the rows (dx, dy) of a (2 K + 1)^2 window in row-major order, every Team::lanes-th of them for this lane
struct RowWalk {
  int dx, dy, K, step;
  FIESTA_NN_HD inline RowWalk(int K_, int rank, int lanes) : dx(-K_), dy(-K_ + rank), K(K_), step(lanes) { wrap(); }
  FIESTA_NN_HD inline void wrap() {
    while (dy > K) dy -= 2 * K + 1, ++dx;
  }
  FIESTA_NN_HD inline bool done() const { return dx > K; }
  FIESTA_NN_HD inline void next() {
    dy += step;
    wrap();
  }
};

// nearest site to the centre of cell (cx, cy, cz) among this lane's rows of the cells within +-K: doubled squared
// distance and word (ties: the smaller word, so that a team agrees whatever the split)
template <class Src, class Team>
FIESTA_NN_HD inline void scan_nearest(const Src &src, const Team &team, int cx, int cy, int cz, int K, int &best_e2, uint32_t &best_w) {
  const int ox = kB * cx, oy = kB * cy, oz = kB * cz;
  for (RowWalk rw(K, team.rank, Team::lanes); !rw.done(); rw.next()) {
    uint32_t i, i1;
    src.bounds(cx + rw.dx, cy + rw.dy, cz - K, cz + K, i, i1);
    for (; i < i1; ++i) {
      const uint32_t w = src.site(i);
      int px, py, pz;
      site_offset<Src::wrap>(w, ox, oy, oz, px, py, pz);
      const int e2 = e2_of(px, py, pz);
      if (e2 < best_e2 || (e2 == best_e2 && w < best_w)) best_e2 = e2, best_w = w;
    }
  }
}

// The competitor of cell (cx, cy, cz): the nearest site to its centre among the 3^3 cells around it, or the first ring beyond
// that holds one (up to the source's reach).  te2 stays kNone if there is none.
template <class Src, class Team>
FIESTA_NN_HD inline void first_competitor(const Src &src, Team &team, int cx, int cy, int cz, int kfirst, int &te2, uint32_t &tw) {
  constexpr int kmax = Src::reach < kKmax ? Src::reach : kKmax;
  te2 = kNone, tw = 0xFFFFFFFFu;
  for (int K = kfirst; K <= kmax && te2 == kNone; ++K) {
    scan_nearest(src, team, cx, cy, cz, K, te2, tw);
    team.nearest(te2, tw);
  }
}
// cells the search window of a cell with competitor (te2, tw) reaches (what the sweep costs: k_nn_lists sorts its cells by it)
template <bool WRAP = false>
FIESTA_NN_HD inline int window_reach(int te2, uint32_t tw, int cx, int cy, int cz) {
  if (te2 == kNone) return kKmax + 1;
  int qx, qy, qz;
  site_offset<WRAP>(tw, kB * cx, kB * cy, kB * cz, qx, qy, qz);
  const int fx = qx > 7 - qx ? qx : 7 - qx, fy = qy > 7 - qy ? qy : 7 - qy, fz = qz > 7 - qz ? qz : 7 - qz;
  const int Kc = ((int)(FIESTA_NN_SQRT((float)(FIESTA_NN_MUL(fx, fx) + FIESTA_NN_MUL(fy, fy) + FIESTA_NN_MUL(fz, fz))) + 0.02f) + 7) >> 3;
  const int Kb = reach_of(rad2_of(te2));
  return Kb < Kc ? Kb : Kc;
}

// The record of cell (cx, cy, cz) into out[kStride].  Returns the number of entries; 0 when the cell cannot be served: no
// site within the search window's reach, a competitor so far away that the window would exceed kKmax cells, more than
// kCap survivors; -1 (nothing written) when the window would leave what this source can serve (Src::reach cells): the
// caller asks again with a source that reaches farther.  (Every lane of a team returns the same value; lane 0 writes the
// count.)  have: the competitor (te2_in, tw_in) was found already (first_competitor with kKfirst).  fr: a shard's region
// (open faces fail the cells whose window touches them; the entries' words are in global coordinates).
template <class Src, class Team>
FIESTA_NN_HD inline int build_list(const Src &src, Team &team, int cx, int cy, int cz, uint32_t *out, bool have = false, int te2_in = kNone,
                                   uint32_t tw_in = 0xFFFFFFFFu, const Frame fr = Frame{}) {
  constexpr int kmax = Src::reach < kKmax ? Src::reach : kKmax;
  const int ox = kB * cx, oy = kB * cy, oz = kB * cz;
  int raw = 0;
  int reach = kKmax;  // (a cell that cannot be served: anything that changes within the widest window may help it)
  int why = kWhySparse;  // should the cell end without a list: nothing within reach / a shard's open face / too many survivors
  // Second try: more candidates than the scratch holds are collected again against the NEAREST site of the 5^3 cells (the
  // first competitor came from the 3^3 cells and may be a poor one).
  for (int kfirst = kKfirst; kfirst <= 2; kfirst = 2 + (raw <= kRaw)) {
    int te2 = te2_in;
    uint32_t tw = tw_in;
    if (!have || kfirst != kKfirst) first_competitor(src, team, cx, cy, cz, kfirst, te2, tw);
    raw = 0;
    if (te2 == kNone) {
      if (kmax < kKmax) return -1;
      break;
    }
    int qx, qy, qz;
    site_offset<Src::wrap>(tw, ox, oy, oz, qx, qy, qz);
    const int q2 = FIESTA_NN_MUL(qx, qx) + FIESTA_NN_MUL(qy, qy) + FIESTA_NN_MUL(qz, qz);
    const int rad2 = rad2_of(te2);
    // Two bounds on where a winner can lie, both from the competitor t: the ball (|s - c| <= |t - c| + 2 h), and a cube -- a
    // site s that is nearest to some voxel v of the cell has |s_a - v_a| <= |s - v| <= |t - v| <= M on every axis, M the
    // distance from t to the cell's farthest corner, so p_a lies in [-M, 7 + M]: (M + 7) / 8 cells either way.  The cube is
    // the tighter one along the axes (M <= |t - c| + h), the ball cuts its corners: the window is their intersection.
    const int Kw = window_reach<Src::wrap>(te2, tw, cx, cy, cz);
    if (Kw > kmax) {
      if (kfirst < 2) {  // (a poor competitor widens the window: look for the nearest one before giving up)
        raw = kRaw + 1;
        continue;
      }
      if (kmax < kKmax) return -1;
      raw = 0;
      break;
    }
    if (fr.open && leaves_region(fr, cx, cy, cz, Kw)) {  // a shard: obstacles beyond the region could win in this cell
      if (kfirst < 2) {
        raw = kRaw + 1;
        continue;
      }
      raw = 0;
      why = kWhyOpen;
      break;
    }
    reach = Kw;
    why = kWhyDense;
    team.restart();
    for (RowWalk rw(Kw, team.rank, Team::lanes); !rw.done(); rw.next()) {
      const int gx = gap_of(rw.dx), gy = gap_of(rw.dy);
      const int rem = rad2 - FIESTA_NN_MUL(gx, gx) - FIESTA_NN_MUL(gy, gy);
      if (rem < 0) continue;  // the whole row of cells lies outside the ball
      int m = (int)((FIESTA_NN_SQRT((float)rem) + 7.02f) * 0.0625f);  // cells along z the ball still touches
      m = m > Kw ? Kw : m;
      uint32_t i, i1;
      src.bounds(cx + rw.dx, cy + rw.dy, cz - m, cz + m, i, i1);
      for (; i < i1; ++i) {
        const uint32_t w = src.site(i);
        int px, py, pz;
        site_offset<Src::wrap>(w, ox, oy, oz, px, py, pz);
        bool keep = w == tw || !dominated(px, py, pz, qx, qy, qz, q2);
        if (keep && kfirst >= 2 && w != tw) {
          // the second try also asks the candidates already collected (dominance is transitive: whoever drops s here, or
          // whoever later drops that one, stands in for s) -- a list that overflowed against t alone comes out thinned
          const int have_n = team.count_now();
          for (int j = 0; j < have_n && j < kRaw && keep; ++j) {
            int ux, uy, uz;
            unpack_p(team.get(j), ux, uy, uz);
            keep = !dominated(px, py, pz, ux, uy, uz, FIESTA_NN_MUL(ux, ux) + FIESTA_NN_MUL(uy, uy) + FIESTA_NN_MUL(uz, uz)) ||
                   (ux == px && uy == py && uz == pz);
          }
        }
        if (keep) {
          const int k = team.slot();
          if (k < kRaw) team.put(k, pack_p(px, py, pz));
        }
      }
    }
    raw = team.count();
    if (kfirst >= 2) break;
  }
  int n = 0;
  if (raw > 0 && raw <= kRaw) {
    // A long list is thinned pairwise: a candidate any OTHER candidate dominates goes (dominance is transitive, so testing
    // against dropped ones is as good) -- 25 -> 13 entries at most on config 2's scene; short lists (the mean is 6) skip it.
    if (raw > kThin) {
      for (int k = team.rank; k < raw; k += Team::lanes) {
        const uint32_t pk = team.get(k);
        int px, py, pz;
        unpack_p(pk, px, py, pz);
        bool dead = false;
        for (int j = 0; j < raw && !dead; ++j) {
          int qx, qy, qz;
          unpack_p(team.get(j) & 0xFFFFFFu, qx, qy, qz);
          dead = j != k && dominated(px, py, pz, qx, qy, qz, qx * qx + qy * qy + qz * qz);
        }
        if (dead) team.put(k, pk | 0x80000000u);  // (the low 24 bits stay: others still test against it)
      }
    }
    team.restart();
    for (int k = team.rank; k < raw; k += Team::lanes) {
      const uint32_t pk = team.get(k);
      if (pk & 0x80000000u) continue;
      int px, py, pz;
      unpack_p(pk, px, py, pz);
      const int i = team.slot();
      if (i < kCap) {
        uint32_t *e = out + 4 + 4 * i;
        e[0] = entry_b(py, pz), e[1] = entry_k(px, py, pz, i), e[2] = entry_m(px);
        // (the word a voxel stores: global coordinates, modulo 1024 on grids beyond that -- common.hpp: pack_coc)
        e[3] = (((uint32_t)(px + ox + fr.wx) & 1023u) << 20) | (((uint32_t)(py + oy + fr.wy) & 1023u) << 10) | ((uint32_t)(pz + oz + fr.wz) & 1023u);
      }
    }
    n = team.count();
  }
  if (n > kCap) n = 0;
  if (team.rank == 0) {
    out[0] = (uint32_t)n;
    // the window's reach; a cell without a list: the widest reach (whatever changes around it may help), and why -- k_nn_brute
    // serves the few cells a scene leaves without a list one by one (sparse: against every site; dense: against its window)
    out[1] = n ? (uint32_t)reach : ((uint32_t)kKmax | ((uint32_t)why << 8) | ((uint32_t)reach << 16));
    if (n & 1) {  // (the kernel takes two entries per step)
      uint32_t *e = out + 4 + 4 * n;
      e[0] = 0u, e[1] = kPadK, e[2] = 0u, e[3] = 0u;
    }
  }
  return n;
}

}  // namespace nn
}  // namespace fiesta
