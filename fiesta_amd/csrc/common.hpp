// fiesta_amd/csrc/common.hpp -- shared host/device definitions of the gfx950 ESDF engine.
//
// Voxel state in HBM is ONE 32-bit word per voxel: the packed GLOBAL coordinates of the closest
// obstacle (the reference's closest_obstacle_, include/ESDFMap.h:90). The reference's
// distance_buffer_ (f64) is not stored: Dist() (src/ESDFMap.cpp:122-124) is a pure function of
// (voxel, closest obstacle), so d^2 is recomputed in registers and a query returns
// sqrt((double)d2) * resolution, which is bit-identical to the reference's value. The per-obstacle
// doubly-linked lists (head_/prev_/next_) do not exist here at all -- see DESIGN.md.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdexcept>
#include <string>

namespace fiesta {

// ---- voxel word encoding -------------------------------------------------------------------------
//   bit 31 = 1 : no closest obstacle
//        0xFFFFFFFF  never observed   (reference: distance_buffer_ == -10000)
//        0x80000000  observed, no obstacle reached (reference: +10000, closest_obstacle_ undefined)
//        0xC0000000  as above, and freshly invalidated by a delete (transient seed of the frontier)
//   bit 31 = 0 : bits 29..0 = x<<20 | y<<10 | z of the closest obstacle (global voxel coordinates)
//   bit 30     : ACT, "this voxel belongs to the frontier" -- a transient tag that is only ever set in
//                HBM by the seeding kernels and consumed by the first relaxation round; inside LDS it is
//                the per-sweep frontier bit.
typedef uint32_t vox_t;
constexpr vox_t kUnobserved = 0xFFFFFFFFu;
constexpr vox_t kNoCoc = 0x80000000u;  // bit: word carries no obstacle
constexpr vox_t kInf = 0x80000000u;
constexpr vox_t kAct = 0x40000000u;
constexpr vox_t kReset = kInf | kAct;
constexpr int32_t kD2Inf = 0x7FFFFFFF;
// A "no obstacle" word may still carry a STALE LINK in its low 30 bits (kAct clear): the local-map reset of UpdateOccupancy
// (src/ESDFMap.cpp:256-259) sets the distance to infinity but leaves closest_obstacle_ -- and the voxel's membership in
// that obstacle's list -- alone, so the voxel is re-seeded from its neighbours when that obstacle is deleted later
// (:308-321).  Everything that asks "is there an obstacle" keeps testing kNoCoc; only the delete scan follows the link.
// (An id of all zeros cannot be told from plain kInf: a link to map voxel (0,0,0) mod 1024 is dropped.)
constexpr vox_t kIdMask = 0x3FFFFFFFu;
__host__ __device__ inline bool has_link(vox_t w) { return !(w & kNoCoc) || (!(w & kAct) && (w & kIdMask) != 0u); }
__host__ __device__ inline vox_t stale_link(vox_t w) { return (w & kNoCoc) ? w : (kNoCoc | (w & kIdMask)); }
constexpr int kCoordBits = 10;
constexpr int kMaxDim = 1 << kCoordBits;

// Ids are GLOBAL voxel coordinates modulo 1024 per axis.  A grid of at most 1024 voxels per axis stores the plain
// coordinate.  A larger (sharded) grid -- BASELINE config 5: 2048^3 -- decodes an id RELATIVE TO THE VOXEL THAT HOLDS IT
// ("wrap"): the obstacle is the one point congruent to the id within (-512, 512) voxels of the voxel on every axis.
// That needs no wider word and no second encoding; its price is a reach of 512 voxels: on such grids a candidate with
// d^2 >= 2^18 is never adopted (kD2Cap), i.e. a voxel farther than 51.2 m (at 0.1 m) from every obstacle reads "no
// obstacle".  The reference cannot hold such a grid at all (int indices, 48 B/voxel: 412 GB at 2048^3).
__host__ __device__ inline vox_t pack_coc(int x, int y, int z) {
  return ((vox_t)(x & 1023) << 20) | ((vox_t)(y & 1023) << 10) | (vox_t)(z & 1023);
}
constexpr int32_t kD2Cap = 1 << 18;  // wrap maps: squared reach of an id
// offset of voxel (x,y,z) from the obstacle its word c names: voxel minus obstacle, per axis
__host__ __device__ inline void coc_offset(int wrap, int x, int y, int z, vox_t c, int &dx, int &dy, int &dz) {
  dx = x - (int)((c >> 20) & 1023), dy = y - (int)((c >> 10) & 1023), dz = z - (int)(c & 1023);
  if (wrap) dx = ((dx + 512) & 1023) - 512, dy = ((dy + 512) & 1023) - 512, dz = ((dz + 512) & 1023) - 512;
}
// the obstacle itself, in global coordinates
__host__ __device__ inline void unpack_coc(int wrap, int x, int y, int z, vox_t c, int &cx, int &cy, int &cz) {
  int dx, dy, dz;
  coc_offset(wrap, x, y, z, c, dx, dy, dz);
  cx = x - dx, cy = y - dy, cz = z - dz;
}
// squared voxel distance; exact in int32
__host__ __device__ inline int32_t dist2(int wrap, int x, int y, int z, vox_t c) {
  int dx, dy, dz;
  coc_offset(wrap, x, y, z, c, dx, dy, dz);
  return dx * dx + dy * dy + dz * dz;
}
// plain forms for coordinate systems that never exceed 1024 per axis (the hash-block map's window, the ray codes)
__host__ __device__ inline void unpack_coc(vox_t c, int &x, int &y, int &z) {
  x = (c >> 20) & 1023;
  y = (c >> 10) & 1023;
  z = c & 1023;
}
__host__ __device__ inline int32_t dist2(int x, int y, int z, vox_t c) { return dist2(0, x, y, z, c); }

// Colour of a distance in GetSliceMarker (src/ESDFMap.cpp:584-636, 673): hue h (taken modulo 1) around the colour circle
// at full saturation and value; the six sectors only permute (1, 1 - f', 0).
__host__ __device__ inline void rainbow_rgba(double h, float *rgba) {
  h -= floor(h);
  h *= 6;
  const int sector = (int)floor(h);
  double f = h - sector;
  if (!(sector & 1)) f = 1 - f;
  const double lvl[3] = {1.0, 1.0 - f, 0.0};
  // which level goes to r, g, b in sector 0..5 (6 wraps to 0): two bits each
  const unsigned perm[6] = {0u | 1u << 2 | 2u << 4, 1u | 0u << 2 | 2u << 4, 2u | 0u << 2 | 1u << 4,
                            2u | 1u << 2 | 0u << 4, 1u | 2u << 2 | 0u << 4, 0u | 2u << 2 | 1u << 4};
  const unsigned pm = perm[sector % 6];
  rgba[0] = (float)lvl[pm & 3], rgba[1] = (float)lvl[(pm >> 2) & 3], rgba[2] = (float)lvl[(pm >> 4) & 3], rgba[3] = 1.0f;
}

// The 24-direction stencil (include/parameters.h:54-68): 6 faces, 12 edges, 6 two-step faces.
// Order is the reference's; it is irrelevant for the fixed point.
// the same stencil as two halves of 12 (for kernels that batch the neighbour reads but are short of registers)
#define FIESTA_STENCIL12A(X)                                                                            \
  X(-1, 0, 0) X(1, 0, 0) X(0, -1, 0) X(0, 1, 0) X(0, 0, -1) X(0, 0, 1) X(-1, -1, 0) X(1, 1, 0)          \
  X(0, -1, -1) X(0, 1, 1) X(-1, 0, -1) X(1, 0, 1)
#define FIESTA_STENCIL12B(X)                                                                            \
  X(-1, 1, 0) X(1, -1, 0) X(0, -1, 1) X(0, 1, -1) X(1, 0, -1) X(-1, 0, 1) X(-2, 0, 0) X(2, 0, 0)        \
  X(0, -2, 0) X(0, 2, 0) X(0, 0, -2) X(0, 0, 2)
// ... and as four groups of 6
#define FIESTA_STENCIL6A(X) X(-1, 0, 0) X(1, 0, 0) X(0, -1, 0) X(0, 1, 0) X(0, 0, -1) X(0, 0, 1)
#define FIESTA_STENCIL6B(X) X(-1, -1, 0) X(1, 1, 0) X(0, -1, -1) X(0, 1, 1) X(-1, 0, -1) X(1, 0, 1)
#define FIESTA_STENCIL6C(X) X(-1, 1, 0) X(1, -1, 0) X(0, -1, 1) X(0, 1, -1) X(1, 0, -1) X(-1, 0, 1)
#define FIESTA_STENCIL6D(X) X(-2, 0, 0) X(2, 0, 0) X(0, -2, 0) X(0, 2, 0) X(0, 0, -2) X(0, 0, 2)
// (as initialiser lists)
#define FIESTA_DIR3(DX, DY, DZ) {DX, DY, DZ},
#define FIESTA_STENCIL24(X)                                                                             \
  X(-1, 0, 0) X(1, 0, 0) X(0, -1, 0) X(0, 1, 0) X(0, 0, -1) X(0, 0, 1) X(-1, -1, 0) X(1, 1, 0)          \
  X(0, -1, -1) X(0, 1, 1) X(-1, 0, -1) X(1, 0, 1) X(-1, 1, 0) X(1, -1, 0) X(0, -1, 1) X(0, 1, -1)       \
  X(1, 0, -1) X(-1, 0, 1) X(-2, 0, 0) X(2, 0, 0) X(0, -2, 0) X(0, 2, 0) X(0, 0, -2) X(0, 0, 2)

// Appends `value` to list[] for every ACTIVE lane whose `pred` holds, with ONE atomicAdd per wave (hundreds of
// thousands of appends per batch would otherwise serialise on the counter). Safe in divergent code: the ballot only
// sees the lanes that execute the call. Work-groups are 1-D with a multiple of 64 threads everywhere in this engine.
__device__ inline void wave_append(bool pred, uint32_t value, uint32_t *list, unsigned long long *counter) {
  const unsigned long long m = __ballot(pred);
  if (!m) return;
  const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
  uint32_t base = 0;
  if (lane == leader) base = (uint32_t)atomicAdd(counter, (unsigned long long)__popcll(m));
  base = __shfl(base, leader);
  if (pred) list[base + __popcll(m & ((1ull << lane) - 1ull))] = value;
}

// ---- geometry of one (shard of a) dense grid -------------------------------------------------------
struct Geom {
  int nx, ny, nz;     // local array extent (owned box + ghost layers when sharded)
  int nzw;            // 32-bit words per z-row in the bitmaps = ceil(nz/32)
  int64_t n;          // nx*ny*nz
  int gx0, gy0, gz0;  // global coordinates of local voxel (0,0,0)
  int ox0, oy0, oz0, ox1, oy1, oz1;  // owned box, local coords, inclusive (== whole array if unsharded)
  int wx0, wy0, wz0, wx1, wy1, wz1;  // update window (VoxInRange), local coords, inclusive
  int px0, py0, pz0, px1, py1, pz1;  // previous window (last_min_vec_/last_max_vec_)
  int wrap;              // 1: some GLOBAL extent exceeds 1024: ids are decoded relative to their voxel (see pack_coc)
  int sharded;           // 1: this array is one shard (owned box + 2-voxel ghost layers) of a GX x GY x GZ grid
  int GX, GY, GZ, GZW;   // global grid and words per z-row of the replicated global occupancy bitmap
  double org[3], res, res_inv;
  double lo[3], hi[3];  // min_range_/max_range_
  __host__ __device__ inline int64_t idx(int x, int y, int z) const { return ((int64_t)x * ny + y) * nz + z; }
  __host__ __device__ inline int64_t bitword(int x, int y, int z) const {
    return ((int64_t)x * ny + y) * nzw + (z >> 5);
  }
  __host__ __device__ inline bool in_grid(int x, int y, int z) const {
    return (unsigned)x < (unsigned)nx && (unsigned)y < (unsigned)ny && (unsigned)z < (unsigned)nz;
  }
  __host__ __device__ inline bool in_window(int x, int y, int z) const {
    return x >= wx0 && x <= wx1 && y >= wy0 && y <= wy1 && z >= wz0 && z <= wz1;
  }
  __host__ __device__ inline bool in_prev_window(int x, int y, int z) const {
    return x >= px0 && x <= px1 && y >= py0 && y <= py1 && z >= pz0 && z <= pz1;
  }
  __host__ __device__ inline int64_t gbitword(int gx, int gy, int gz) const {
    return ((int64_t)gx * GY + gy) * GZW + (gz >> 5);
  }
  __host__ __device__ inline bool owned(int x, int y, int z) const {
    return x >= ox0 && x <= ox1 && y >= oy0 && y <= oy1 && z >= oz0 && z <= oz1;
  }
};

struct ProbParams {
  double l_hit, l_miss, l_min, l_max, l_occ;
};

// ---- errors -------------------------------------------------------------------------------------------
struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
#define FIESTA_HIP_CHECK(expr)                                                                          \
  do {                                                                                                  \
    hipError_t _e = (expr);                                                                             \
    if (_e != hipSuccess)                                                                               \
      throw ::fiesta::Error(_e == hipErrorOutOfMemory ? 3 : 2, std::string(#expr) + ": " +              \
                                                                   hipGetErrorString(_e));              \
  } while (0)

// The library carries gfx950 code objects only: on any other HIP device a launch would fail without a code object and
// leave the map uninitialised, so creation refuses it up front.
inline void require_gfx950(int device) {
  hipDeviceProp_t p;
  FIESTA_HIP_CHECK(hipGetDeviceProperties(&p, device));
  if (std::string(p.gcnArchName).compare(0, 6, "gfx950") != 0)
    throw Error(2, std::string("device is ") + p.gcnArchName + ", this engine is built for gfx950 (MI355X) only");
}

}  // namespace fiesta
