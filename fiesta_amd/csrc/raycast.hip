// fiesta_amd/csrc/raycast.hip -- per-ray HIP kernels that turn one sensor frame into the occupancy delta.
//
// Replaces (reference = HKUST-Aerial-Robotics/FIESTA):
//   Raycast(start,end,min,max,&out)          src/raycast.cpp:56-158   -> dda_walk (device, one CASTING ray per lane)
//   Fiesta::RaycastProcess / Multithread     include/Fiesta.h:194-303 -> k_ray_ends / k_ray_cast_list / k_ray_walk /
//                                                                       k_ray_codes / k_ray_resolve_w / k_ray_apply_w
//   pinhole part of Fiesta::DepthConversion  include/Fiesta.h:341-351 -> k_depth_points
//
// The reference processes the cloud sequentially and de-duplicates per frame with two stamp arrays:
//   set_occ_  : only the FIRST point (cloud order) that falls into an end-point voxel casts a ray;
//   set_free_ : a ray walks its voxels far -> near, counts a "miss" in each, and STOPS at the first voxel
//               that an EARLIER ray of the frame already stamped (that voxel is still counted once more).
// Which voxels a ray touches therefore depends on all earlier rays. Both rules are
// reproduced exactly, not approximated, by making "earlier" explicit:
//   * end points: atomicMin of the ray index per end-point voxel; the minimum index wins;
//   * free space: let F[v] be the smallest index of a ray whose (truncated) walk contains v. Ray i is
//     truncated at the first voxel with F[v] < i. F and the truncations are a fixed point of each other and
//     the fixed point is unique (ray 0 is never truncated, ray 1 only depends on ray 0, ...), so iterating
//     "truncate with the previous F, rebuild F with atomicMin" from "nobody is truncated" converges to
//     exactly the sequential result; the loop stops when no ray's truncation changed (a dozen or two rounds on a
//     depth image, each a replay of the casting rays' stored walks, a wave per ray).
// Counters are applied once, by k_ray_apply_w, along the final truncated walks.
// All ray arithmetic is f64 in the reference's operation order and this file is compiled with
// -ffp-contract=off, so voxel decisions (floor, comparisons with min/max ray length) match bit for bit.
#include <cmath>
#include <cstring>

#include "dense_map.hpp"
#include "hash_map.hpp"

namespace fiesta {

constexpr uint32_t kCodeSkip = 0xFFFFFFFFu;      // no-op entry: beyond max range or outside the map
constexpr uint32_t kCodeMinBreak = 0xFFFFFFFEu;  // closer than min_ray_length: the walk ends here
constexpr uint32_t kCodeNoCount = 0x40000000u;   // inside the map but outside the update window
constexpr uint32_t kCodeIdxMask = 0x3FFFFFFFu;
constexpr int kMaxRayVoxels = 1500;              // src/raycast.cpp:127-130

struct RayArgs {
  double T[16];
  double o[3];
  double minr, maxr;
  double lc[3], rc[3];  // l_cornor / r_cornor in metres
  int dedup;
  int inverse;  // SIGNED_NEEDED companion map (include/Fiesta.h:216-218,249-251): end points count as free, crossings as occupied
};

__device__ inline int sgn_i(int v) { return v == 0 ? 0 : (v < 0 ? -1 : 1); }            // signum (:6-8)
__device__ inline double wrap1(double v) { return fmod(fmod(v, 1.0) + 1.0, 1.0); }      // mod (:10-12)
__device__ inline double first_crossing(double s, double ds) {                          // intbound (:14-23)
  if (ds < 0) {
    s = -s;
    ds = -ds;
  }
  return (1 - wrap1(s)) / ds;
}

// Amanatides-Woo traversal with the reference's arithmetic (src/raycast.cpp:56-158). `emit(x,y,z,k)` is
// called for every voxel the reference pushes; returns the count, or -1 if it would exceed 1500 voxels.
template <typename Emit>
__device__ inline int dda_walk(const double *a, const double *b, const double *lo, const double *hi, Emit emit) {
  int c[3] = {(int)floor(a[0]), (int)floor(a[1]), (int)floor(a[2])};
  const int e[3] = {(int)floor(b[0]), (int)floor(b[1]), (int)floor(b[2])};
  const double r0 = b[0] - a[0], r1 = b[1] - a[1], r2 = b[2] - a[2];
  const double reach2 = r0 * r0 + r1 * r1 + r2 * r2;
  double tmax[3], tstep[3];
  int step[3];
  for (int i = 0; i < 3; ++i) {
    const double delta = e[i] - c[i];  // NB: integer voxel delta, not the true ray direction (:89-91)
    step[i] = sgn_i((int)delta);
    tmax[i] = first_crossing(a[i], delta);
    tstep[i] = ((double)step[i]) / delta;
  }
  if (step[0] == 0 && step[1] == 0 && step[2] == 0) return 0;
  int count = 0;
  for (int guard = 0; guard < 8192; ++guard) {
    if (c[0] >= lo[0] && c[0] < hi[0] && c[1] >= lo[1] && c[1] < hi[1] && c[2] >= lo[2] && c[2] < hi[2]) {
      emit(c[0], c[1], c[2], count);
      ++count;
      const double q0 = c[0] - a[0], q1 = c[1] - a[1], q2 = c[2] - a[2];
      if (q0 * q0 + q1 * q1 + q2 * q2 > reach2) return count;
      if (count > kMaxRayVoxels) return -1;
    }
    if (c[0] == e[0] && c[1] == e[1] && c[2] == e[2]) break;
    int ax;  // strict '<' tie rules (:139-157)
    if (tmax[0] < tmax[1])
      ax = (tmax[0] < tmax[2]) ? 0 : 2;
    else
      ax = (tmax[1] < tmax[2]) ? 1 : 2;
    c[ax] += step[ax];
    tmax[ax] += tstep[ax];
  }
  return count;
}

__device__ inline bool ray_pos_in_map(const Geom &g, const double *p) {
  return !(p[0] < g.lo[0] || p[1] < g.lo[1] || p[2] < g.lo[2] || p[0] > g.hi[0] || p[1] > g.hi[1] || p[2] > g.hi[2]);
}

__device__ inline void count_observation(int64_t idx, int occ, unsigned long long *cnt, uint32_t *touched,
                                         unsigned long long *counters) {
  const unsigned long long old = atomicAdd(&cnt[idx], ((unsigned long long)(uint32_t)occ << 32) | 1ull);
  wave_append((uint32_t)old == 0, (uint32_t)idx, touched, &counters[C_TOUCHED]);
}

// ---- paged (hash-block) maps: window coordinates (map voxel - window origin g.gx0..), pool address through the directory ----
namespace paged {
constexpr int kWin = HashMap::kWin, kNTY = HashMap::kNTY, kNTZ = HashMap::kNTZ, kPageVox = HashMap::kPageVox;
__device__ inline int tile_id(int x, int y, int z) { return ((x >> 4) * kNTY + (y >> 4)) * kNTZ + (z >> 5); }
__device__ inline bool in_win(int x, int y, int z) {
  return (unsigned)x < (unsigned)kWin && (unsigned)y < (unsigned)kWin && (unsigned)z < (unsigned)kWin;
}
__device__ inline int64_t vaddr(const int32_t *dir, int x, int y, int z) {
  const int32_t p = dir[tile_id(x, y, z)];
  return p < 0 ? -1 : (int64_t)p * kPageVox + (((x & 15) * 16 + (y & 15)) * 32 + (z & 31));
}
}  // namespace paged

// The end point of cloud point i as RaycastProcess computes it (include/Fiesta.h:202-218): transformed, clipped to
// max_ray_length (then counted FREE). false: the point is skipped (NaN, or closer than min_ray_length).
__device__ inline bool ray_end_point(const RayArgs &ra, const float *pts, int64_t i, double *q, int &occ) {
  const double px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
  if (isnan(px) || isnan(py) || isnan(pz)) return false;  // include/Fiesta.h:202
  double h[4];
  for (int r = 0; r < 4; ++r) h[r] = ra.T[4 * r] * px + ra.T[4 * r + 1] * py + ra.T[4 * r + 2] * pz + ra.T[4 * r + 3] * 1.0;
  q[0] = h[0] / h[3], q[1] = h[1] / h[3], q[2] = h[2] / h[3];  // :204-205
  const double d0 = q[0] - ra.o[0], d1 = q[1] - ra.o[1], d2 = q[2] - ra.o[2];
  const double len = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
  if (len < ra.minr) return false;  // :209
  occ = 1;
  if (len > ra.maxr) {  // clip to max range and mark the clipped end point FREE (:211-213)
    for (int k = 0; k < 3; ++k) q[k] = (q[k] - ra.o[k]) / len * ra.maxr + ra.o[k];
    occ = 0;
  }
  if (ra.inverse) occ = 0;  // inv_esdf_map_->SetOccupancy(point, 0) whatever the end point is (:216-218)
  return true;
}

// The free-space visit of walk voxel (vx,vy,vz) as RaycastProcess would do it (include/Fiesta.h:240-248), dense map:
// what the replay needs to know about it.
__device__ inline uint32_t ray_visit_code(const Geom &g, const RayArgs &ra, int vx, int vy, int vz) {
  const double c[3] = {(vx + 0.5) * g.res, (vy + 0.5) * g.res, (vz + 0.5) * g.res};
  const double e0 = c[0] - ra.o[0], e1 = c[1] - ra.o[1], e2 = c[2] - ra.o[2];
  const double l2 = sqrt(e0 * e0 + e1 * e1 + e2 * e2);
  if (l2 < ra.minr) return kCodeMinBreak;
  if (l2 > ra.maxr || !ray_pos_in_map(g, c)) return kCodeSkip;
  const int x = (int)floor((c[0] - g.org[0]) / g.res) - g.gx0, y = (int)floor((c[1] - g.org[1]) / g.res) - g.gy0,
            z = (int)floor((c[2] - g.org[2]) / g.res) - g.gz0;
  if (!g.in_grid(x, y, z)) return kCodeSkip;
  return (uint32_t)g.idx(x, y, z) | ((g.in_window(x, y, z) && g.owned(x, y, z)) ? 0u : kCodeNoCount);
}

// ... on a paged map: packed window coordinates for now (k_ray_translate_codes), the tile marked for allocation
__device__ inline uint32_t ray_visit_code_paged(const Geom &g, const RayArgs &ra, int vx, int vy, int vz, uint32_t *need) {
  const double c[3] = {(vx + 0.5) * g.res, (vy + 0.5) * g.res, (vz + 0.5) * g.res};
  const double e0 = c[0] - ra.o[0], e1 = c[1] - ra.o[1], e2 = c[2] - ra.o[2];
  const double l2 = sqrt(e0 * e0 + e1 * e1 + e2 * e2);
  if (l2 < ra.minr) return kCodeMinBreak;
  if (l2 > ra.maxr) return kCodeSkip;
  const int x = (int)floor((c[0] - g.org[0]) / g.res) - g.gx0, y = (int)floor((c[1] - g.org[1]) / g.res) - g.gy0,
            z = (int)floor((c[2] - g.org[2]) / g.res) - g.gz0;
  if (!paged::in_win(x, y, z)) return kCodeSkip;
  need[paged::tile_id(x, y, z)] = 1u;
  return pack_coc(x, y, z) | (g.in_window(x, y, z) ? 0u : kCodeNoCount);
}

// ---- dense maps: end points for every point of the cloud, walks only for the rays that cast -------------------------
// A 640x480 depth image holds ~300 k points but only a few thousand distinct end-point voxels, and only the first point
// of each casts a ray (set_occ_, include/Fiesta.h:221-232).  So the frame is split: k_ray_ends (one lane per point:
// count the end point, stamp set_occ_), k_ray_cast_list (the winners, compacted), k_ray_walk (one lane per CASTING
// ray: the traversal, stored ray-major), then the de-duplication rounds and the counting with one WAVE per casting
// ray -- a lane per walk entry, so that a round costs two memory latencies instead of one per voxel of the walk.
// flags: bit0 valid point, bit1 casts (winner of its end-point voxel)
__global__ void k_ray_ends(Geom g, RayArgs ra, const float *pts, int64_t n, int32_t *end_idx, uint8_t *flags,
                           uint32_t *stamp_occ, uint32_t tagged, unsigned long long *cnt, uint32_t *touched,
                           unsigned long long *counters) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  double q[3];
  int occ = 0;
  const bool valid = i < n && ray_end_point(ra, pts, i, q, occ);
  // SetOccupancy(point, occ) (src/ESDFMap.cpp:401-437); every valid point counts its end point
  int eidx = -1;
  bool counts = false;
  if (valid && ray_pos_in_map(g, q)) {
    const int x = (int)floor((q[0] - g.org[0]) / g.res) - g.gx0, y = (int)floor((q[1] - g.org[1]) / g.res) - g.gy0,
              z = (int)floor((q[2] - g.org[2]) / g.res) - g.gz0;
    if (g.in_grid(x, y, z)) {
      eidx = (int)g.idx(x, y, z);
      counts = g.in_window(x, y, z) && g.owned(x, y, z);
    }
  }
  if (i < n) {
    flags[i] = valid ? 1 : 0;
    end_idx[i] = eidx;
  }
  // Neighbouring pixels end in the same voxel a dozen at a time: one atomic per RUN of equal end points inside the wave
  // (the counter word takes the run's observations and hits at once; the run's first lane carries its smallest index)
  const int lane = threadIdx.x & 63;
  const int prev = __shfl_up(eidx, 1);
  const unsigned long long heads = __ballot(lane == 0 || eidx != prev), hits = __ballot(occ == 1);
  if (eidx >= 0 && ((heads >> lane) & 1ull)) {
    const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1));
    const int len = above ? __ffsll((long long)above) : 64 - lane;
    const unsigned long long run = (len == 64 ? ~0ull : ((1ull << len) - 1ull)) << lane;
    if (counts) {
      const unsigned long long old = atomicAdd(&cnt[eidx], ((unsigned long long)__popcll(hits & run) << 32) | (unsigned long long)len);
      wave_append((uint32_t)old == 0, (uint32_t)eidx, touched, &counters[C_TOUCHED]);
    }
    if (ra.dedup) atomicMin(&stamp_occ[eidx], tagged | (uint32_t)i);  // set_occ_ (:221-232)
  }
}

// the casting rays, compacted (any order: every later step identifies a ray by its cloud index)
__global__ __launch_bounds__(1024) void k_ray_cast_list(int64_t n, int dedup, const int32_t *end_idx, uint8_t *flags,
                                                        const uint32_t *stamp_occ, uint32_t tagged, int32_t *cast, int *n_cast) {
  __shared__ int wcount[16], wbase[16];
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  bool casts = false;
  if (i < n) {
    const uint8_t f = flags[i];
    const int32_t e = end_idx[i];
    casts = (f & 1) && (!dedup || e < 0 || stamp_occ[e] == (tagged | (uint32_t)i));
    if (casts) flags[i] = f | 2;
  }
  const unsigned long long m = __ballot(casts);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wcount[wave] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {  // one atomic per work-group
    int tot = 0;
    for (int w = 0; w < 16; ++w) wbase[w] = tot, tot += wcount[w];
    const int base = tot ? atomicAdd(n_cast, tot) : 0;
    for (int w = 0; w < 16; ++w) wbase[w] += base;
  }
  __syncthreads();
  if (casts) cast[wbase[wave] + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)i;
}

// Raycast(origin/res, point/res, l_cornor/res, r_cornor/res) (include/Fiesta.h:233-237) for casting ray c, one lane per
// ray.  The traversal is serial but light; what the visit of each voxel needs (a square root and three divisions in
// f64) is not, and does not depend on the step before.  So this pass only records the PATH: the first voxel of the walk
// (walk0) and, per further voxel, which axis stepped and in which direction -- the voxels a ray emits are consecutive
// steps, the clipping box being convex -- and k_ray_codes turns the path into visit codes with a lane per voxel.
constexpr uint32_t kStepNone = 6;
__global__ void k_ray_walk(Geom g, RayArgs ra, const float *pts, const int32_t *cast, const int *n_cast, int stride,
                           uint32_t *entries, int32_t *walk0, int32_t *m_count, int32_t *last_k, int *err) {
  const int nc = *n_cast;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += gridDim.x * blockDim.x) {
    double q[3], a[3], b[3], lo[3], hi[3];
    int occ;
    (void)ray_end_point(ra, pts, cast[c], q, occ);
    for (int k = 0; k < 3; ++k) {
      a[k] = ra.o[k] / g.res;
      b[k] = q[k] / g.res;
      lo[k] = ra.lc[k] / g.res;
      hi[k] = ra.rc[k] / g.res;
    }
    uint32_t *row = entries + (int64_t)c * stride;
    bool overflow = false;
    int px = 0, py = 0, pz = 0;
    int m = dda_walk(a, b, lo, hi, [&](int vx, int vy, int vz, int k) {
      if (k >= stride) {
        overflow = true;
        return;
      }
      if (k == 0) {
        walk0[3 * c] = vx, walk0[3 * c + 1] = vy, walk0[3 * c + 2] = vz;
        row[0] = kStepNone;
      } else {
        row[k] = vx != px ? (vx < px ? 1u : 0u) : (vy != py ? (vy < py ? 3u : 2u) : (vz < pz ? 5u : 4u));
      }
      px = vx, py = vy, pz = vz;
    });
    if (m < 0 || overflow) {  // (the reference throws; so does the host once the frame is through)
      atomicExch(err, 1);
      m = 0;
    }
    m_count[c] = m;
    last_k[c] = m;  // "nothing visited yet"
  }
}

// path -> visit codes, in place: one wave per casting ray, a lane per voxel of the walk (the voxel = walk0 + the steps
// so far: an inclusive wave scan of the three axes packed in one word, every lane adding 1 to each field so that
// no field ever borrows)
template <bool PAGED>
__global__ void k_ray_codes(Geom g, RayArgs ra, const int *n_cast, int stride, uint32_t *entries, const int32_t *walk0,
                            const int32_t *m_count, uint32_t *need) {
  const int lane = threadIdx.x & 63, nwaves = (gridDim.x * blockDim.x) >> 6;
  const int nc = *n_cast;
  for (int c = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; c < nc; c += nwaves) {
    const int m = m_count[c];
    uint32_t *row = entries + (int64_t)c * stride;
    int cx = walk0[3 * c], cy = walk0[3 * c + 1], cz = walk0[3 * c + 2];
    for (int base = 0; base < m; base += 64) {
      const int k = base + lane;
      const uint32_t st = k < m ? row[k] : kStepNone;
      uint32_t v = 0x00100401u;  // +1 in each 10-bit field
      if (st < kStepNone) v += (st & 1u) ? -(1u << (10 * (st >> 1))) : (1u << (10 * (st >> 1)));
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(v, off);
        if (lane >= off) v += o;
      }
      const int dx = (int)(v & 1023u) - (lane + 1), dy = (int)((v >> 10) & 1023u) - (lane + 1), dz = (int)((v >> 20) & 1023u) - (lane + 1);
      if (k < m) row[k] = PAGED ? ray_visit_code_paged(g, ra, cx + dx, cy + dy, cz + dz, need) : ray_visit_code(g, ra, cx + dx, cy + dy, cz + dz);
      const uint32_t tot = __shfl(v, 63);
      cx += (int)(tot & 1023u) - 64, cy += (int)((tot >> 10) & 1023u) - 64, cz += (int)((tot >> 20) & 1023u) - 64;
    }
  }
}

// One fixed-point round, one wave per casting ray: truncate the ray with the previous round's first-stamper array and
// rebuild the array for the next round.  Rounds are launched in batches without a host round trip: round `it` (1-based)
// reports into changed[it]; a round that finds its predecessor unchanged (the fixed point) does nothing, and so do all
// later rounds of the batch (their changed[] entries stay 0).  Lane l of a chunk looks at walk entry
// top - l; the first lane that meets the min-length break or an earlier ray's stamp ends the walk.
__global__ void k_ray_resolve_w(int stride, const uint32_t *entries, const int32_t *cast, const int *n_cast,
                                const int32_t *m_count, int32_t *last_k, int have_prev, int stamp, const uint32_t *fprev,
                                uint32_t tag_prev, uint32_t *fnext, uint32_t tag_next, int ibits, int *changed, int it) {
  if (it >= 3 && changed[it - 1] == 0) return;  // round 1 always "changes" (from nothing visited to the full walk)
  const int lane = threadIdx.x & 63, nwaves = (gridDim.x * blockDim.x) >> 6;
  const int nc = *n_cast;
  const uint32_t imask = (1u << ibits) - 1u;
  for (int c = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; c < nc; c += nwaves) {
    const uint32_t i = (uint32_t)cast[c];
    const int m = m_count[c];
    const uint32_t *row = entries + (int64_t)c * stride;
    int lk = m;  // lowest visited entry
    for (int top = m - 2; top >= 0; top -= 64) {
      const int k = top - lane;
      const uint32_t code = k >= 0 ? row[k] : kCodeSkip;
      const bool mb = code == kCodeMinBreak;  // include/Fiesta.h:243-244
      const bool noop = mb || code == kCodeSkip;  // :245-246 and the "-10000" case of :253
      const uint32_t idx = code & kCodeIdxMask;
      bool stop = false;
      if (have_prev && !noop) {
        const uint32_t v = fprev[idx];
        stop = (v >> ibits) == tag_prev && (v & imask) < i;  // set_free_[idx] == tt (:265-269)
      }
      const unsigned long long bm = __ballot(mb), bs = __ballot(stop);
      const int first_mb = bm ? __ffsll((long long)bm) - 1 : 64, first_stop = bs ? __ffsll((long long)bs) - 1 : 64;
      // visited lanes: everything before the break; up to and including the stamped voxel
      const int nvis = min(min(first_mb, first_stop + 1), min(64, top + 1));
      if (stamp && lane < nvis && !noop) atomicMin(&fnext[idx], (tag_next << ibits) | i);
      if (nvis > 0) lk = top - (nvis - 1);
      if (first_mb < 64 || first_stop < 64) break;
    }
    if (lane == 0 && lk != last_k[c]) {
      last_k[c] = lk;
      changed[it] = 1;
    }
  }
}

// SetOccupancy(tmp, 0) along the final walks (include/Fiesta.h:248; inverse map: 1, :250), one wave per casting ray
__global__ void k_ray_apply_w(int stride, const uint32_t *entries, const int *n_cast, const int32_t *m_count,
                              const int32_t *last_k, unsigned long long *cnt, uint32_t *touched, unsigned long long *counters,
                              int free_occ) {
  const int lane = threadIdx.x & 63, nwaves = (gridDim.x * blockDim.x) >> 6;
  const int nc = *n_cast;
  for (int c = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; c < nc; c += nwaves) {
    const int m = m_count[c], lk = last_k[c];
    const uint32_t *row = entries + (int64_t)c * stride;
    for (int top = m - 2; top >= lk; top -= 64) {
      const int k = top - lane;
      const uint32_t code = k >= lk ? row[k] : kCodeSkip;
      const bool counts = code != kCodeSkip && !(code & kCodeNoCount);
      unsigned long long old = 1;
      if (counts) old = atomicAdd(&cnt[code & kCodeIdxMask], ((unsigned long long)(uint32_t)free_occ << 32) | 1ull);
      wave_append(counts && (uint32_t)old == 0, code & kCodeIdxMask, touched, &counters[C_TOUCHED]);
    }
  }
}

// ---- paged (hash-block) maps: the same organisation, in two steps around the allocation of pages ----------------------
// A voxel's slot in the page pool is only known once its page exists.  End points: k_ray_ends_paged stores packed WINDOW
// coordinates (end_idx: coords | occ << 30) and marks the tiles it touches (the reference's Vox2Idx allocates on every
// SetOccupancy, in or out of the update window, src/ESDFMap.cpp:418-421,732-765); after the pages have been allocated
// k_ray_translate_ends turns coordinates into slots, counts the observations and stamps set_occ_.  Walks: the path pass
// is the dense map's; k_ray_codes<true> leaves window coordinates in the entries and marks the tiles of the CASTING
// rays' walks (the reference never touches a voxel of a ray it does not cast); k_ray_translate_codes makes them slots.
__global__ void k_ray_ends_paged(Geom g, RayArgs ra, const float *pts, int64_t n, int32_t *end_idx, uint8_t *flags, uint32_t *need) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  double q[3];
  int occ = 0;
  const bool valid = ray_end_point(ra, pts, i, q, occ);
  int eidx = -1;
  if (valid) {  // PosInMap is always true for the hash build (src/ESDFMap.cpp:46-48): the virtual window is the map
    const int x = (int)floor((q[0] - g.org[0]) / g.res) - g.gx0, y = (int)floor((q[1] - g.org[1]) / g.res) - g.gy0,
              z = (int)floor((q[2] - g.org[2]) / g.res) - g.gz0;
    if (paged::in_win(x, y, z)) {
      eidx = (int)(pack_coc(x, y, z) | ((uint32_t)occ << 30));
      need[paged::tile_id(x, y, z)] = 1u;
    }
  }
  flags[i] = valid ? 1 : 0;
  end_idx[i] = eidx;
}

// (one counter update and one stamp per run of equal end points inside the wave, as in k_ray_ends)
__global__ void k_ray_translate_ends(Geom g, const int32_t *dir, int64_t n, int dedup, int32_t *end_idx, uint32_t *stamp_occ,
                                     uint32_t tagged, unsigned long long *cnt, uint32_t *touched, unsigned long long *counters) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int32_t e = i < n ? end_idx[i] : -1;
  int64_t addr = -1;
  bool counts = false;
  if (e >= 0) {
    int x, y, z;
    unpack_coc((vox_t)e & kCodeIdxMask, x, y, z);
    addr = paged::vaddr(dir, x, y, z);
    counts = g.in_window(x, y, z);
    end_idx[i] = (int32_t)addr;
  }
  const int lane = threadIdx.x & 63;
  const int prev = __shfl_up(e, 1);
  const unsigned long long heads = __ballot(lane == 0 || e != prev);
  if (e >= 0 && addr >= 0 && ((heads >> lane) & 1ull)) {
    const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1));
    const int len = above ? __ffsll((long long)above) : 64 - lane;
    if (counts) {
      const unsigned long long hits = ((e >> 30) & 1) ? (unsigned long long)len : 0ull;
      const unsigned long long old = atomicAdd(&cnt[addr], (hits << 32) | (unsigned long long)len);
      wave_append((uint32_t)old == 0, (uint32_t)addr, touched, &counters[C_TOUCHED]);
    }
    if (dedup) atomicMin(&stamp_occ[addr], tagged | (uint32_t)i);  // set_occ_ (include/Fiesta.h:221-232)
  }
}

// window coordinates -> pool slots, in place (one wave per casting ray)
__global__ void k_ray_translate_codes(const int32_t *dir, const int *n_cast, int stride, uint32_t *entries, const int32_t *m_count) {
  const int lane = threadIdx.x & 63, nwaves = (gridDim.x * blockDim.x) >> 6;
  const int nc = *n_cast;
  for (int c = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; c < nc; c += nwaves) {
    const int m = m_count[c];
    uint32_t *row = entries + (int64_t)c * stride;
    for (int k = lane; k < m; k += 64) {
      const uint32_t code = row[k];
      if (code == kCodeSkip || code == kCodeMinBreak) continue;
      int x, y, z;
      unpack_coc(code & kCodeIdxMask, x, y, z);
      row[k] = (uint32_t)paged::vaddr(dir, x, y, z) | (code & kCodeNoCount);
    }
  }
}

// Pinhole back-projection (include/Fiesta.h:341-351): uint16 millimetres -> float sensor-frame points.
__global__ void k_depth_points(const uint16_t *depth, int rows, int cols, double fx, double fy, double cx, double cy,
                               float *pts) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows * cols) return;
  const int v = i / cols, u = i % cols;
  const double d = depth[i] / 1000.0;  // k_depth_scaling_factor (:328)
  pts[3 * i] = (float)((u - cx) * d / fx);
  pts[3 * i + 1] = (float)((v - cy) * d / fy);
  pts[3 * i + 2] = (float)d;
}

// ... with the temporal consistency filter of DepthConversion (include/Fiesta.h:352-379): a pixel survives iff it lies
// inside the margin, its depth within [min, max], and its re-projection into the PREVIOUS depth image (rel =
// last_transform_^-1 * transform_) lands inside that image and agrees with the depth stored there within the tolerance.
// The reference builds a shorter cloud; here a rejected pixel becomes a NaN point, which RaycastProcess skips
// (include/Fiesta.h:202) -- the surviving points keep their cloud order, which is what the per-frame de-duplication
// depends on.  f64 in the reference's operation order.
struct DepthFilterArgs {
  double tol, dmax, dmin;
  int margin;
  double rel[16];
};
__global__ void k_depth_points_filtered(const uint16_t *depth, const uint16_t *last, int rows, int cols, double fx, double fy,
                                        double cx, double cy, DepthFilterArgs f, float *pts, unsigned long long *n_valid) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  bool keep = false;
  if (i < (int64_t)rows * cols) {
    const int v = i / cols, u = i % cols;
    const double d = depth[i] / 1000.0;
    const float px = (float)((u - cx) * d / fx), py = (float)((v - cy) * d / fy), pz = (float)d;
    if (last && v >= f.margin && v < rows - f.margin && u >= f.margin && u < cols - f.margin && !(d > f.dmax || d < f.dmin)) {
      double h[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) h[r] = f.rel[4 * r] * px + f.rel[4 * r + 1] * py + f.rel[4 * r + 2] * pz + f.rel[4 * r + 3] * 1.0;
      const double qx = h[0] / h[3], qy = h[1] / h[3], qz = h[2] / h[3];
      const double uu = qx * fx / qz + cx, vv = qy * fy / qz + cy;
      if (uu >= 0 && uu < cols && vv >= 0 && vv < rows)
        keep = fabs(last[(int64_t)(int)vv * cols + (int)uu] / 1000.0 - qz) < f.tol;
    }
    const float nanf_ = __int_as_float(0x7FC00000);
    pts[3 * i] = keep ? px : nanf_;
    pts[3 * i + 1] = keep ? py : nanf_;
    pts[3 * i + 2] = keep ? pz : nanf_;
  }
  const unsigned long long m = __ballot(keep);
  if (m && (threadIdx.x & 63) == 0 && n_valid) atomicAdd(n_valid, (unsigned long long)__popcll(m));
}

__global__ void k_raycast_one(const double *io, double *out, int cap, int *n_out) {
  int cnt = dda_walk(io, io + 3, io + 6, io + 9, [&](int x, int y, int z, int k) {
    if (k < cap) {
      out[3 * k] = x;
      out[3 * k + 1] = y;
      out[3 * k + 2] = z;
    }
  });
  *n_out = cnt;
}

// =====================================================================================================
// host side, shared by the dense-array and the paged map
struct RayState {
  DevBuf<uint32_t> entries;
  DevBuf<int32_t> end_idx, m_count, last_k;
  DevBuf<int32_t> walk0;  // dense maps: first voxel of each casting ray's walk
  DevBuf<int32_t> cast;  // dense maps: cloud indices of the casting rays (m_count, last_k, entries rows are per casting ray)
  DevBuf<uint8_t> flags;
  DevBuf<float> points;
  DevBuf<uint16_t> depth, last_depth;  // the image being converted; the previous one (temporal depth filter)
  int last_rows = 0, last_cols = 0;    // 0: no previous image (start of a run: image_cnt_ == 1)
  unsigned long long *d_valid = nullptr;
  uint32_t *stamp_occ = nullptr, *fa = nullptr, *fb = nullptr;
  size_t stamp_words = 0;
  int ibits = 0;
  uint32_t tag = 0;
  static constexpr int kMaxRounds = 240, kFlagInts = kMaxRounds + 8;
  int *d_flags = nullptr;  // [0] casting rays of the frame (dense maps), [1] error, [2 + it] "round it changed something"
  int *h_flags = nullptr;
  int64_t last_iterations = 0;
  RayState() {
    FIESTA_HIP_CHECK(hipMalloc((void **)&d_flags, kFlagInts * sizeof(int)));
    FIESTA_HIP_CHECK(hipHostMalloc((void **)&h_flags, kFlagInts * sizeof(int)));
    FIESTA_HIP_CHECK(hipMalloc((void **)&d_valid, sizeof(unsigned long long)));
  }
  ~RayState() {
    if (stamp_occ) (void)hipFree(stamp_occ);
    if (fa) (void)hipFree(fa);
    if (fb) (void)hipFree(fb);
    if (d_flags) (void)hipFree(d_flags);
    if (h_flags) (void)hipHostFree(h_flags);
    if (d_valid) (void)hipFree(d_valid);
  }
};
struct DenseMap::RaycastState : RayState {};
struct HashMap::RaycastState : RayState {};

void DenseMap::free_raycast_state() {
  delete rc_;
  rc_ = nullptr;
}
void HashMap::free_raycast_state() {
  delete rc_;
  rc_ = nullptr;
}

static inline int rgrid(int64_t n) { return (int)std::max<int64_t>(1, (n + 255) / 256); }

// per-ray buffers of one frame; returns the row stride of `entries`
static int ray_frame_buffers(RayState &rc, double res, int64_t n, const fiesta_hip_raycast_params *p, hipStream_t stream) {
  if (n >= (1ll << 26)) throw Error(FIESTA_HIP_ERR_INVALID, "more than 2^26 points in one frame");
  if (!(p->max_ray_length > 0) || !(p->min_ray_length >= 0)) throw Error(FIESTA_HIP_ERR_INVALID, "bad ray length window");
  // a ray of length <= max_ray_length crosses at most |dx|+|dy|+|dz|+1 voxels
  const int per_axis = (int)std::ceil(p->max_ray_length / res) + 2;
  const int stride = std::min(kMaxRayVoxels + 1, 3 * per_axis + 2);
  rc.entries.ensure((size_t)stride * n, stream);
  rc.end_idx.ensure(n, stream);
  rc.m_count.ensure(n, stream);
  rc.last_k.ensure(n, stream);
  rc.flags.ensure(n, stream);
  return stride;
}

// tagged per-frame stamps (set_occ_/set_free_): value = tag << ibits | ray index, newer tags are smaller. `words` = the
// number of voxel slots (grid size, or the capacity of the page pool). A frame uses one tag for the end points and
// one per fixed-point round; the three arrays are cleared only when the tag space is about to run out (every few
// hundred frames), when the ray-index width changes, or when they had to grow.
static int ray_stamps(RayState &rc, size_t words, int64_t n, hipStream_t stream) {
  int ibits = 20;
  while ((1ll << ibits) < n) ++ibits;
  bool fresh = false;
  if (rc.stamp_words < words) {
    FIESTA_HIP_CHECK(hipStreamSynchronize(stream));
    for (uint32_t **q : {&rc.stamp_occ, &rc.fa, &rc.fb}) {
      if (*q) (void)hipFree(*q);
      *q = nullptr;
      FIESTA_HIP_CHECK(hipMalloc((void **)q, words * sizeof(uint32_t)));
    }
    rc.stamp_words = words;
    fresh = true;
  }
  if (fresh || ibits != rc.ibits || rc.tag < (uint32_t)RayState::kMaxRounds + 16u) {
    for (uint32_t *q : {rc.stamp_occ, rc.fa, rc.fb}) FIESTA_HIP_CHECK(hipMemsetAsync(q, 0xFF, rc.stamp_words * sizeof(uint32_t), stream));
    rc.ibits = ibits;
    rc.tag = (0xFFFFFFFFu >> ibits) - 1u;
  }
  return ibits;
}

static RayArgs ray_args(const double *T, const double *origin, const fiesta_hip_raycast_params *p) {
  RayArgs ra;
  memcpy(ra.T, T, sizeof(ra.T));
  for (int k = 0; k < 3; ++k) {
    ra.o[k] = origin[k];
    ra.lc[k] = p->l_cornor[k];
    ra.rc[k] = p->r_cornor[k];
  }
  ra.minr = p->min_ray_length;
  ra.maxr = p->max_ray_length;
  ra.dedup = p->dedup ? 1 : 0;
  ra.inverse = p->inverse ? 1 : 0;
  return ra;
}

// The casting rays of the frame (compacted) and their walks, as visit codes (paged maps: still window coordinates, tiles
// marked in `need`).  Nothing here waits for the host.
static void ray_cast_walks(RayState &rc, const Geom &g, const RayArgs &ra, const float *dpts, int64_t n, int stride, int ibits,
                           uint32_t tag_occ, hipStream_t stream, uint32_t *need) {
  rc.cast.ensure(n, stream);
  int *n_cast = rc.d_flags;  // [0]: casting rays of this frame (zeroed with the flags)
  hipLaunchKernelGGL(k_ray_cast_list, dim3((int)((n + 1023) / 1024)), dim3(1024), 0, stream, n, ra.dedup, (const int32_t *)rc.end_idx.p, rc.flags.p,
                     (const uint32_t *)rc.stamp_occ, ra.dedup ? (tag_occ << ibits) : 0u, rc.cast.p, n_cast);
  FIESTA_HIP_CHECK(hipGetLastError());
  // (grids are sized for the cloud, not for the casting rays -- their number stays on the device; surplus waves leave)
  const int walk_blocks = (int)std::min<int64_t>((n + 63) / 64, 4096);
  const int wave_blocks = (int)std::min<int64_t>((n + 3) / 4, 2048);  // 4 waves per block
  rc.walk0.ensure(3 * n, stream);
  hipLaunchKernelGGL(k_ray_walk, dim3(walk_blocks), dim3(64), 0, stream, g, ra, dpts, (const int32_t *)rc.cast.p, (const int *)n_cast,
                     stride, rc.entries.p, rc.walk0.p, rc.m_count.p, rc.last_k.p, rc.d_flags + 1);
  FIESTA_HIP_CHECK(hipGetLastError());
  if (need)
    hipLaunchKernelGGL(k_ray_codes<true>, dim3(wave_blocks), dim3(256), 0, stream, g, ra, (const int *)n_cast, stride, rc.entries.p,
                       (const int32_t *)rc.walk0.p, (const int32_t *)rc.m_count.p, need);
  else
    hipLaunchKernelGGL(k_ray_codes<false>, dim3(wave_blocks), dim3(256), 0, stream, g, ra, (const int *)n_cast, stride, rc.entries.p,
                       (const int32_t *)rc.walk0.p, (const int32_t *)rc.m_count.p, (uint32_t *)nullptr);
  FIESTA_HIP_CHECK(hipGetLastError());
}

// The free-space fixed point with one wave per casting ray, then the counters.  Entries index cnt[], the stamp arrays and
// the touched list (dense: voxel index; paged: pool slot).  The batches of the fixed point are the only host round trips.
static void ray_cast_rounds(RayState &rc, const RayArgs &ra, int64_t n, int stride, int ibits, unsigned long long *cnt,
                            uint32_t *touched, unsigned long long *counters, hipStream_t stream) {
  int *n_cast = rc.d_flags;
  const int wave_blocks = (int)std::min<int64_t>((n + 3) / 4, 2048);  // 4 waves per block
  auto resolve = [&](int have_prev, int stamp, const uint32_t *fprev, uint32_t tag_prev, uint32_t *fnext, uint32_t tag_next, int it) {
    hipLaunchKernelGGL(k_ray_resolve_w, dim3(wave_blocks), dim3(256), 0, stream, stride, (const uint32_t *)rc.entries.p,
                       (const int32_t *)rc.cast.p, (const int *)n_cast, (const int32_t *)rc.m_count.p, rc.last_k.p, have_prev, stamp,
                       fprev, tag_prev, fnext, tag_next, ibits, rc.d_flags + 2, it);
    FIESTA_HIP_CHECK(hipGetLastError());
  };
  int64_t iters = 0;
  if (!ra.dedup) {
    resolve(0, 0, nullptr, 0u, nullptr, 0u, 1);
  } else {
    uint32_t *fprev = rc.fa, *fnext = rc.fb;
    uint32_t tag_prev = 0;
    // rounds per host round trip: as many as the previous frame needed plus one (consecutive frames of a sensor need
    // about the same number: one round trip per frame), then a few at a time
    constexpr int kBatch = 4;
    int batch = (int)std::min<int64_t>(std::max<int64_t>(rc.last_iterations + 1, kBatch), 64);
    bool done = false;
    int64_t base = 0;  // rounds whose flag slots were recycled (the fixed point needs at most n rounds: ray i depends
                       // only on rays before it; a valid frame never fails here, however long its dependency chains)
    while (!done) {
      const int64_t first = iters + 1;
      // (a frame starts with at least kMaxRounds + 16 free tags, ray_stamps; one whose dependency chains need more rounds
      //  than tags are left must fail loudly: a wrapped tag would collide with older stamps -- ADVICE r3)
      if (rc.tag <= (uint32_t)batch)
        throw Error(FIESTA_HIP_ERR_STATE, "raycast de-dup: the frame's dependency chains exhausted the stamp tags (split the cloud into smaller frames)");
      for (int b = 0; b < batch; ++b) {
        const uint32_t tag_next = rc.tag--;
        ++iters;
        resolve(iters > 1 ? 1 : 0, 1, fprev, tag_prev, fnext, tag_next, (int)(iters - base));
        std::swap(fprev, fnext);
        tag_prev = tag_next;
      }
      FIESTA_HIP_CHECK(hipMemcpyAsync(rc.h_flags, rc.d_flags, (2 + (iters - base) + 1) * sizeof(int), hipMemcpyDeviceToHost, stream));
      FIESTA_HIP_CHECK(hipStreamSynchronize(stream));
      if (rc.h_flags[1]) break;
      for (int64_t it = std::max<int64_t>(first, 2); it <= iters; ++it)  // the first round (after round 1) that changed nothing
        if (!rc.h_flags[2 + (it - base)]) {
          iters = it;
          done = true;
          break;
        }
      batch = kBatch;
      if (!done && (iters - base) + kBatch > RayState::kMaxRounds) {  // recycle the per-round flag slots
        FIESTA_HIP_CHECK(hipMemsetAsync(rc.d_flags + 2, 0, (RayState::kFlagInts - 2) * sizeof(int), stream));
        base = iters;
      }
      if (!done && iters > n + 64) throw Error(FIESTA_HIP_ERR_STATE, "raycast de-dup: more rounds than rays (internal error)");
    }
  }
  rc.last_iterations = iters;
  hipLaunchKernelGGL(k_ray_apply_w, dim3(wave_blocks), dim3(256), 0, stream, stride, (const uint32_t *)rc.entries.p, (const int *)n_cast,
                     (const int32_t *)rc.m_count.p, (const int32_t *)rc.last_k.p, cnt, touched, counters, ra.inverse ? 1 : 0);
  FIESTA_HIP_CHECK(hipGetLastError());
  FIESTA_HIP_CHECK(hipMemcpyAsync(rc.h_flags, rc.d_flags, 2 * sizeof(int), hipMemcpyDeviceToHost, stream));
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream));
  if (rc.h_flags[1])  // the reference throws std::out_of_range("Too many RaycasMultithread voxels")
    throw Error(FIESTA_HIP_ERR_INVALID, "Too many raycast voxels (a ray crosses more than 1500 voxels)");
}

static const float *ray_points(RayState &rc, const float *points, int64_t n, bool dev, hipStream_t stream) {
  if (dev) return points;
  rc.points.ensure(3 * n, stream);
  FIESTA_HIP_CHECK(hipMemcpyAsync(rc.points.p, points, 3 * n * sizeof(float), hipMemcpyHostToDevice, stream));
  return rc.points.p;
}

static const float *ray_depth_points(RayState &rc, const uint16_t *depth, int rows, int cols, double fx, double fy, double cx,
                                     double cy, hipStream_t stream, const fiesta_hip_depth_filter *f = nullptr,
                                     int64_t *n_valid = nullptr) {
  const int64_t n = (int64_t)rows * cols;
  rc.depth.ensure(n, stream);
  rc.points.ensure(3 * n, stream);
  FIESTA_HIP_CHECK(hipMemcpyAsync(rc.depth.p, depth, n * sizeof(uint16_t), hipMemcpyHostToDevice, stream));
  if (!f) {
    hipLaunchKernelGGL(k_depth_points, dim3(rgrid(n)), dim3(256), 0, stream, (const uint16_t *)rc.depth.p, rows, cols, fx, fy, cx,
                       cy, rc.points.p);
    FIESTA_HIP_CHECK(hipGetLastError());
    if (n_valid) *n_valid = n;
    return rc.points.p;
  }
  // DepthConversion with use_depth_filter_ (include/Fiesta.h:352-379): the previous image stays on the device
  if (f->reset || rc.last_rows != rows || rc.last_cols != cols) rc.last_rows = rc.last_cols = 0;
  DepthFilterArgs a;
  a.tol = f->tolerance, a.dmax = f->max_dist, a.dmin = f->min_dist, a.margin = f->margin;
  memcpy(a.rel, f->rel_transform, sizeof(a.rel));
  FIESTA_HIP_CHECK(hipMemsetAsync(rc.d_valid, 0, sizeof(unsigned long long), stream));
  hipLaunchKernelGGL(k_depth_points_filtered, dim3(rgrid(n)), dim3(256), 0, stream, (const uint16_t *)rc.depth.p,
                     rc.last_rows ? (const uint16_t *)rc.last_depth.p : (const uint16_t *)nullptr, rows, cols, fx, fy, cx, cy, a,
                     rc.points.p, rc.d_valid);
  FIESTA_HIP_CHECK(hipGetLastError());
  if (n_valid) {
    unsigned long long h = 0;
    FIESTA_HIP_CHECK(hipMemcpyAsync(&h, rc.d_valid, sizeof(h), hipMemcpyDeviceToHost, stream));
    FIESTA_HIP_CHECK(hipStreamSynchronize(stream));
    *n_valid = (int64_t)h;
  }
  std::swap(rc.depth.p, rc.last_depth.p);  // current_img becomes last_img (:322-323)
  std::swap(rc.depth.cap, rc.last_depth.cap);
  rc.last_rows = rows, rc.last_cols = cols;
  return rc.points.p;
}

void DenseMap::raycast_frame(const float *points, int64_t n, const double *T, const double *origin,
                             const fiesta_hip_raycast_params *p, bool dev) {
  use_device();
  if (n <= 0) return;
  if (!rc_) rc_ = new RaycastState;
  RaycastState &rc = *rc_;
  const Geom &g = g_;
  enable_distance_tracking();  // depth frames = many small deltas: bound the delete scans from now on
  const int stride = ray_frame_buffers(rc, g.res, n, p, stream_);
  const float *dpts = ray_points(rc, points, n, dev, stream_);
  const RayArgs ra = ray_args(T, origin, p);
  const int ibits = ra.dedup ? ray_stamps(rc, (size_t)g.n, n, stream_) : 20;
  // worst case every ray touches `stride` new voxels
  ensure_touched_capacity(std::min<int64_t>(g.n, n * (int64_t)(stride + 1)));
  FIESTA_HIP_CHECK(hipMemsetAsync(rc.d_flags, 0, RayState::kFlagInts * sizeof(int), stream_));
  const uint32_t tag_occ = ra.dedup ? rc.tag-- : 0;
  hipLaunchKernelGGL(k_ray_ends, dim3(rgrid(n)), dim3(256), 0, stream_, g, ra, dpts, n, rc.end_idx.p, rc.flags.p, rc.stamp_occ,
                     ra.dedup ? (tag_occ << ibits) : 0u, cnt_, touched_.p, counters_);
  FIESTA_HIP_CHECK(hipGetLastError());
  ray_cast_walks(rc, g, ra, dpts, n, stride, ibits, tag_occ, stream_, nullptr);
  ray_cast_rounds(rc, ra, n, stride, ibits, cnt_, touched_.p, counters_, stream_);
}

void DenseMap::raycast_depth(const uint16_t *depth, int rows, int cols, double fx, double fy, double cx, double cy,
                             const double *T, const double *origin, const fiesta_hip_raycast_params *p,
                             const fiesta_hip_depth_filter *f) {
  use_device();
  if (!rc_) rc_ = new RaycastState;
  const float *pts = ray_depth_points(*rc_, depth, rows, cols, fx, fy, cx, cy, stream_, f);
  raycast_frame(pts, (int64_t)rows * cols, T, origin, p, true);
}
int64_t DenseMap::depth_conversion(const uint16_t *depth, int rows, int cols, double fx, double fy, double cx, double cy,
                                   const fiesta_hip_depth_filter *f, float *points_out) {
  use_device();
  if (!rc_) rc_ = new RaycastState;
  int64_t n_valid = 0;
  const float *pts = ray_depth_points(*rc_, depth, rows, cols, fx, fy, cx, cy, stream_, f, &n_valid);
  FIESTA_HIP_CHECK(hipMemcpyAsync(points_out, pts, (size_t)rows * cols * 3 * sizeof(float), hipMemcpyDeviceToHost, stream_));
  FIESTA_HIP_CHECK(hipStreamSynchronize(stream_));
  return n_valid;
}

// Paged map: the same frame in three steps -- walk the rays in window coordinates and mark the tiles they touch,
// allocate the missing pages, translate coordinates to pool slots (+ count / stamp the end points) -- then the common
// rounds. The stamp arrays cover the pool's capacity and are re-created when the pool grows.
void HashMap::raycast_frame(const float *points, int64_t n, const double *T, const double *origin,
                            const fiesta_hip_raycast_params *p, bool dev) {
  use_device();
  if (n <= 0) return;
  if (!rc_) rc_ = new RaycastState;
  RaycastState &rc = *rc_;
  {  // everything a frame observes lies within max_ray_length of the sensor: keep that ball inside the window
    const double reach = std::min(p->max_ray_length + 2 * g_.res, (kWin / 2 - 16) * g_.res);
    int64_t lo[3], hi[3];
    bool sane = reach > 0;
    for (int k = 0; k < 3; ++k) {
      const double a = std::floor((origin[k] - reach - g_.org[k]) / g_.res), b = std::floor((origin[k] + reach - g_.org[k]) / g_.res);
      sane &= std::fabs(a) < 1e9 && std::fabs(b) < 1e9;
      lo[k] = (int64_t)a, hi[k] = (int64_t)b;
    }
    if (sane) ensure_window(lo, hi);
  }
  const int stride = ray_frame_buffers(rc, g_.res, n, p, stream_);
  const float *dpts = ray_points(rc, points, n, dev, stream_);
  const RayArgs ra = ray_args(T, origin, p);
  FIESTA_HIP_CHECK(hipMemsetAsync(rc.d_flags, 0, RayState::kFlagInts * sizeof(int), stream_));
  hipLaunchKernelGGL(k_ray_ends_paged, dim3(rgrid(n)), dim3(256), 0, stream_, g_, ra, dpts, n, rc.end_idx.p, rc.flags.p, need_);
  FIESTA_HIP_CHECK(hipGetLastError());
  allocate_marked();  // the end points' pages
  auto slots_fit = [&]() {
    if ((int64_t)cap_pages_ * kPageVox > (1ll << 30))
      throw Error(FIESTA_HIP_ERR_INVALID, "ray cast: page pool larger than 2^30 voxels (walk entries hold 30-bit slots)");
  };
  slots_fit();
  int ibits = ra.dedup ? ray_stamps(rc, (size_t)cap_pages_ * kPageVox, n, stream_) : 20;
  touched_upper_ = std::min<int64_t>(npages_ * kPageVox, touched_upper_ + n);
  touched_.ensure((size_t)touched_upper_, stream_, touched_.cap);
  const uint32_t tag_occ = ra.dedup ? rc.tag-- : 0;
  hipLaunchKernelGGL(k_ray_translate_ends, dim3(rgrid(n)), dim3(256), 0, stream_, g_, (const int32_t *)dir_, n, ra.dedup, rc.end_idx.p,
                     rc.stamp_occ, ra.dedup ? (tag_occ << ibits) : 0u, cnt_.p, touched_.p, counters_);
  FIESTA_HIP_CHECK(hipGetLastError());
  ray_cast_walks(rc, g_, ra, dpts, n, stride, ibits, tag_occ, stream_, need_);
  allocate_marked();  // the pages the casting rays cross
  slots_fit();
  // (a pool that had to grow takes the stamp arrays with it: the end points' stamps have done their work by now)
  if (ra.dedup) ibits = ray_stamps(rc, (size_t)cap_pages_ * kPageVox, n, stream_);
  touched_upper_ = std::min<int64_t>(npages_ * kPageVox, touched_upper_ + n * (int64_t)stride);
  touched_.ensure((size_t)touched_upper_, stream_, touched_.cap);
  hipLaunchKernelGGL(k_ray_translate_codes, dim3((int)std::min<int64_t>((n + 3) / 4, 2048)), dim3(256), 0, stream_, (const int32_t *)dir_,
                     (const int *)rc.d_flags, stride, rc.entries.p, (const int32_t *)rc.m_count.p);
  FIESTA_HIP_CHECK(hipGetLastError());
  ray_cast_rounds(rc, ra, n, stride, ibits, cnt_.p, touched_.p, counters_, stream_);
}

void HashMap::raycast_depth(const uint16_t *depth, int rows, int cols, double fx, double fy, double cx, double cy,
                            const double *T, const double *origin, const fiesta_hip_raycast_params *p,
                            const fiesta_hip_depth_filter *f) {
  use_device();
  if (!rc_) rc_ = new RaycastState;
  const float *pts = ray_depth_points(*rc_, depth, rows, cols, fx, fy, cx, cy, stream_, f);
  raycast_frame(pts, (int64_t)rows * cols, T, origin, p, true);
}

void raycast_single(const double *start, const double *end, const double *minv, const double *maxv, double *out,
                    int32_t cap, int32_t *n_out, int32_t device) {
  FIESTA_HIP_CHECK(hipSetDevice(device));
  double io[12];
  memcpy(io, start, 24);
  memcpy(io + 3, end, 24);
  memcpy(io + 6, minv, 24);
  memcpy(io + 9, maxv, 24);
  double *d_io = nullptr, *d_out = nullptr;
  int *d_n = nullptr;
  const int c = std::max(1, (int)cap);
  FIESTA_HIP_CHECK(hipMalloc((void **)&d_io, sizeof(io)));
  FIESTA_HIP_CHECK(hipMalloc((void **)&d_out, (size_t)c * 3 * sizeof(double)));
  FIESTA_HIP_CHECK(hipMalloc((void **)&d_n, sizeof(int)));
  FIESTA_HIP_CHECK(hipMemcpy(d_io, io, sizeof(io), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_raycast_one, dim3(1), dim3(1), 0, 0, (const double *)d_io, d_out, (int)cap, d_n);
  int cnt = 0;
  FIESTA_HIP_CHECK(hipMemcpy(&cnt, d_n, sizeof(int), hipMemcpyDeviceToHost));
  if (cnt > 0 && cap > 0)
    FIESTA_HIP_CHECK(hipMemcpy(out, d_out, (size_t)std::min(cnt, (int)cap) * 3 * sizeof(double), hipMemcpyDeviceToHost));
  (void)hipFree(d_io);
  (void)hipFree(d_out);
  (void)hipFree(d_n);
  if (cnt < 0) throw Error(FIESTA_HIP_ERR_INVALID, "Too many raycast voxels (more than 1500)");
  *n_out = cnt;
}

}  // namespace fiesta
