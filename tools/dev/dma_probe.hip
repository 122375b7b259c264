// probe: where does global_load_lds_dword put each lane's dword?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const uint32_t *src, uint32_t *out) {
  __shared__ __attribute__((aligned(16))) uint32_t land[4][2][128];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t zone0 = (uint32_t)(size_t)&land[wave][0][0];
#pragma unroll
  for (int zone = 0; zone < 2; ++zone) {
    const uint32_t *lp = src + (wave * 2 + zone) * 128 + lane, *hp = lp + 64;
    const uint32_t at = zone0 + zone * 512u;
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, off\n\t"
                 "s_mov_b32 m0, %2\n\tglobal_load_lds_dword %3, off"
                 : : "s"(at), "v"(lp), "s"(at + 256u), "v"(hp) : "memory", "m0");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int zone = 0; zone < 2; ++zone)
    for (int j = 0; j < 2; ++j) out[((wave * 2 + zone) * 2 + j) * 64 + lane] = land[wave][zone][j * 64 + lane];
}
int main() {
  uint32_t h[4 * 2 * 128], *d, *o, r[4 * 2 * 128];
  for (int i = 0; i < 1024; ++i) h[i] = i;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(h));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipMemset(o, 0xff, sizeof(h));
  k<<<1, 256>>>(d, o);
  hipMemcpy(r, o, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 1024; ++i) if (r[i] != (uint32_t)i) { if (bad < 20) printf("out[%d] = %u\n", i, r[i]); ++bad; }
  printf("bad %d\n", bad);
  return 0;
}
