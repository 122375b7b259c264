// probe: what does the STORE PATTERN of k_nn_fill_full cost, and is it the strides?  512^3 words (537 MB).
//   linear   every wave 1 KB contiguous per instruction (a fill)
//   quad     k_nn_fill_full's pattern: a wave owns a quad (8 x-slabs x 8 rows x 128 B) and stores it slab by slab -- 8 lines in
//            8 rows per instruction --, a run of 8 quads along z per wave; row stride and plane stride as arguments:
//            2048 / 1 MiB is the real array, padded strides tell whether the powers of two are what it costs
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void k_linear(uint4 *out, size_t n16) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) out[i] = uint4{1, 2, 3, 4};
}
// rows: words between two rows (y), plane: words between two x-planes; the array has 512 x 512 rows of 512 words
__global__ __launch_bounds__(256) void k_quad(uint32_t *out, size_t rows, size_t plane, int per) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int y = lane >> 3, z = lane & 7;
  const uint32_t q0 = (blockIdx.x * 4u + wave) * per;
  for (uint32_t q = q0; q < q0 + per; ++q) {
    const uint32_t row = q >> 4, qq = q & 15, cx = row >> 6, cy = row & 63;   // 16 quads per cell row, 64 x 64 cell rows
    uint32_t *slab = out + (size_t)(8 * cx) * plane + (size_t)(8 * cy) * rows + 32 * qq;
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      *reinterpret_cast<uint4 *>(slab + y * rows + 4 * z) = uint4{q, (uint32_t)x, 3, 4};
      slab += plane;
    }
  }
}
int main(int argc, char **argv) {
  const size_t n = 512ull * 512 * 512;
  uint32_t *d;
  hipMalloc(&d, (n + (64ull << 20)) * 4);   // room for padded strides
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](const char *name, auto launch) {
    float best = 1e9f, sum = 0;
    for (int r = 0; r < 12; ++r) {
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (r >= 2) { best = ms < best ? ms : best; sum += ms; }
    }
    printf("%-44s best %.1f us  mean %.1f us  (%.2f TB/s)\n", name, best * 1e3, sum / 10 * 1e3, n * 4 / (sum / 10 * 1e-3) / 1e12);
  };
  time("linear fill", [&] { k_linear<<<4096, 256>>>((uint4 *)d, n / 4); });
  struct { const char *name; size_t rows, plane; } cases[] = {
      {"quad pattern, rows 512, planes 512*512 (real)", 512, 512ull * 512},
      {"quad pattern, rows 512+32, planes 512*(512+32)", 544, 512ull * 544},
      {"quad pattern, rows 512, planes 512*512+32*8", 512, 512ull * 512 + 256},
      {"quad pattern, rows 512+32, planes +32*8 more", 544, 512ull * 544 + 256},
      {"quad pattern, rows 512+64, planes 512*(512+64)", 576, 512ull * 576},
  };
  for (auto &c : cases) time(c.name, [&] { k_quad<<<2048, 256>>>(d, c.rows, c.plane, 8); });
  return 0;
}
