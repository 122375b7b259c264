#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../fiesta_amd/csrc/nn_core.hpp"
using namespace fiesta::nn;
__device__ __forceinline__ int dotC(uint32_t a, uint32_t b) { int d; asm("v_dot4_i32_i8 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b)); return d; }
template <int V>
__global__ void k(const uint32_t *rec, uint32_t *out) {
  __shared__ __attribute__((aligned(16))) uint32_t lz[128];
  const int lane = threadIdx.x;
  lz[lane] = rec[lane]; lz[64 + lane] = rec[64 + lane];
  __syncthreads();
  const int y = lane >> 3, z = lane & 7;
  const uint32_t ayz = (uint32_t)y | ((uint32_t)z << 8);
  const int cnt = __builtin_amdgcn_readfirstlane((int)lz[0]);
  uint32_t best[8];
#pragma unroll
  for (int x = 0; x < 8; ++x) best[x] = 0xFFFFFFFFu;
  for (int i = 0; i < cnt; i += 2) {
    const uint4 e0 = *reinterpret_cast<const uint4 *>(lz + 4 + 4 * i), e1 = *reinterpret_cast<const uint4 *>(lz + 8 + 4 * i);
    const int d0 = V ? __builtin_amdgcn_sdot4((int)ayz, (int)e0.x, 0, false) : dotC(ayz, e0.x), d1 = V ? __builtin_amdgcn_sdot4((int)ayz, (int)e1.x, 0, false) : dotC(ayz, e1.x);
    uint32_t k0 = ((uint32_t)d0 << 9) + e0.y, k1 = ((uint32_t)d1 << 9) + e1.y;
    if (V == 2) { asm volatile("" : "+v"(k0), "+v"(k1)); }
    best[0] = min(best[0], min(k0, k1));
#pragma unroll
    for (int x = 1; x < 8; ++x) { k0 += e0.z, k1 += e1.z; best[x] = min(best[x], min(k0, k1)); }
  }
#pragma unroll
  for (int x = 0; x < 8; ++x) out[x * 64 + lane] = best[x];
}
int main() {
  const int P[5][3] = {{-3, 9, 2}, {12, -5, 4}, {1, 1, -7}, {20, 3, 3}, {-9, -9, 15}};
  for (int v = 0; v < 3; ++v)
  for (int n = 2; n <= 3; ++n) {
    uint32_t h[128] = {0}, *d, *o, r[512];
    h[0] = n;
    for (int i = 0; i < n; ++i) { uint32_t *e = h + 4 + 4 * i; e[0] = entry_b(P[i][1], P[i][2]); e[1] = entry_k(P[i][0], P[i][1], P[i][2], i); e[2] = entry_m(P[i][0]); e[3] = i; }
    if (n & 1) { uint32_t *e = h + 4 + 4 * n; e[0] = 0; e[1] = kPadK; e[2] = 0; e[3] = 0; }
    (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&o, sizeof(r));
    (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    if (v == 0) k<0><<<1, 64>>>(d, o); else if (v == 1) k<1><<<1, 64>>>(d, o); else k<2><<<1, 64>>>(d, o);
    (void)hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int x = 0; x < 8; ++x) for (int l = 0; l < 64; ++l) {
      uint32_t want = 0xFFFFFFFFu;
      for (int i = 0; i < n; ++i) { const uint32_t kk = key_of(h[4 + 4 * i], h[5 + 4 * i], h[6 + 4 * i], x, l >> 3, l & 7); if (kk < want) want = kk; }
      if (r[x * 64 + l] != want) { if (bad < 3) printf("  n %d x %d lane %d got %08x want %08x\n", n, x, l, r[x * 64 + l], want); ++bad; }
    }
    printf("variant %d n %d bad %d\n", v, n, bad);
  }
  return 0;
}
