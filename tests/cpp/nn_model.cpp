// tests/cpp/nn_model.cpp -- TEST INFRASTRUCTURE: host model of the cell transform's kernels (fiesta_amd/csrc/nn_kernels.hpp).
//
// Compiles fiesta_amd/csrc/nn_core.hpp -- the list rule, the search window, the keys -- with g++ and drives it the way the
// three kernels do: sites ordered by cell row out of the occupancy bitmap (k_nn_cells), one list per cell (k_nn_lists),
// every voxel's minimum key over its cell's list (k_nn_fill).  tests/test_nn_model.py checks the result against scipy's
// exact transform on a machine without a GPU.  Not part of the product: libfiesta_hip.so never links this file.
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../fiesta_amd/csrc/nn_core.hpp"

using namespace fiesta::nn;

extern "C" {
// occ: [nx][ny][nz] bytes (0 / 1).  out: [nx][ny][nz] words (the packed site; 0x80000000 where the cell had no list).
// stats: [0] cells without a list (the kernels would hand the update to the envelope passes), [1] list entries in total,
// [2] longest list, [3] sites.  Returns 0.
static int run_region(const uint8_t *occG, const int *G, const Geom &g, const int *rlo, uint32_t *out, int64_t *stats);

int nn_model_run(const uint8_t *occ, int nx, int ny, int nz, uint32_t *out, int64_t *stats) {
  const int G[3] = {nx, ny, nz}, zero[3] = {0, 0, 0};
  return run_region(occ, G, whole_geom(nx, ny, nz), zero, out, stats);
}

// A SHARD: the array at l0[] (extents ln[]) of the global grid G[], its region grown by mc voxels (nn_core.hpp: region_geom).
// occ: the GLOBAL occupancy [GX][GY][GZ]; out: the array's words [ln0][ln1][ln2] (global coordinates of the nearest site).
// Returns 3 if the region does not fit the site packing.
int nn_model_run_shard(const uint8_t *occ, const int *G, const int *l0, const int *ln, int mc, uint32_t *out, int64_t *stats) {
  Geom g;
  int rlo[3];
  if (!region_geom(G, l0, ln, mc, g, rlo)) return 3;
  return run_region(occ, G, g, rlo, out, stats);
}
}

static int run_region(const uint8_t *occG, const int *G, const Geom &g, const int *rlo, uint32_t *out, int64_t *stats) {
  const int nx = g.nx, ny = g.ny, nz = g.nz;
  auto occ_at = [&](int x, int y, int z) { return occG[((int64_t)(x + rlo[0]) * G[1] + (y + rlo[1])) * G[2] + (z + rlo[2])]; };
  const int64_t nrows = (int64_t)g.ncx * g.ncy;
  std::vector<uint32_t> ctab((size_t)nrows * (g.ncz + 1));
  std::vector<uint32_t> sites;
  // k_nn_cells: per cell row (cx, cy), cells in z order, inside a cell x-major then y then z
  for (int cx = 0; cx < g.ncx; ++cx)
    for (int cy = 0; cy < g.ncy; ++cy) {
      uint32_t *row = ctab.data() + ((int64_t)cx * g.ncy + cy) * (g.ncz + 1);
      for (int cz = 0; cz < g.ncz; ++cz) {
        row[cz] = (uint32_t)sites.size();
        for (int x = kB * cx; x < kB * cx + kB && x < nx; ++x)
          for (int y = kB * cy; y < kB * cy + kB && y < ny; ++y)
            for (int z = kB * cz; z < kB * cz + kB && z < nz; ++z)
              if (occ_at(x, y, z)) sites.push_back((((uint32_t)x & 1023u) << 20) | (((uint32_t)y & 1023u) << 10) | ((uint32_t)z & 1023u));
      }
      row[g.ncz] = (uint32_t)sites.size();
    }
  int64_t failed = 0, entries = 0, longest = 0;
  std::vector<uint32_t> list(kStride);
  for (int cx = g.lx0; cx < g.lx1; ++cx)
    for (int cy = g.ly0; cy < g.ly1; ++cy)
      for (int cz = g.lz0; cz < g.lz1; ++cz) {
        Solo solo;
        int n;
        if (g.big()) {  // (sites modulo 1024: the kernel's WRAP instance)
          const PlainSrcT<true> src{ctab.data(), sites.data(), g.ncx, g.ncy, g.ncz};
          n = build_list(src, solo, cx, cy, cz, list.data(), false, kNone, 0xFFFFFFFFu, frame_of(g));
        } else {
          const PlainSrc src{ctab.data(), sites.data(), g.ncx, g.ncy, g.ncz};
          n = build_list(src, solo, cx, cy, cz, list.data(), false, kNone, 0xFFFFFFFFu, frame_of(g));
        }
        if ((int)list[0] != n) return 2;
        if (n == 0) ++failed;
        entries += n;
        if (n > longest) longest = n;
        const int npad = (n + 1) & ~1;
        // k_nn_fill: lane (y, z), eight x-slabs
        for (int x = 0; x < kB; ++x)
          for (int y = 0; y < kB; ++y)
            for (int z = 0; z < kB; ++z) {
              const int X = kB * cx + x - g.fx, Y = kB * cy + y - g.fy, Z = kB * cz + z - g.fz;  // (the array's coordinates)
              if ((unsigned)X >= (unsigned)g.ax || (unsigned)Y >= (unsigned)g.ay || (unsigned)Z >= (unsigned)g.az) continue;
              uint32_t best = 0xFFFFFFFFu;
              for (int i = 0; i < npad; ++i) {
                const uint32_t k = key_of(list[4 + 4 * i], list[5 + 4 * i], list[6 + 4 * i], x, y, z);
                if (k < best) best = k;
              }
              uint32_t w = 0x80000000u;
              if (n) {
                w = list[4 + ((best & 0x1F0u) >> 2) + 3];
                // the key's distance part is the true squared distance minus |v|^2, biased
                const int GX = X + g.fx + g.wx, GY = Y + g.fy + g.wy, GZ = Z + g.fz + g.wz;  // (the voxel in the words' coordinates)
                int dx, dy, dz;  // the word is decoded relative to its voxel (exact on any grid: the site is < 512 away)
                site_offset<true>(w, GX, GY, GZ, dx, dy, dz);
                const int d2 = dx * dx + dy * dy + dz * dz;
                if ((int)(best >> kSH) - kBias + x * x + y * y + z * z != d2) return 1;
              }
              out[((int64_t)X * g.ay + Y) * g.az + Z] = w;
            }
      }
  stats[0] = failed, stats[1] = entries, stats[2] = longest, stats[3] = (int64_t)sites.size();
  return 0;
}
