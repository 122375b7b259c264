// scratch: windowed pass B
#pragma once
#include <type_traits>
#include <utility>
#include "ft_kernels.hpp"
namespace fiesta {
template <int R>
struct FtWin {
  static constexpr int THR = (R + 1) * (R + 1);
  static constexpr int NK = 32;          // planes held as keys in registers (NK / 2 registers); code unrolled NK steps
  static_assert(THR + R * R < 1024 && 2 * R + 1 <= NK, "16-bit keys; the window fits the key registers");
  static constexpr uint32_t half(int off) { return (off >= 0 && off <= 2 * R) ? (uint32_t)((R - off) * (R - off)) << 6 : 0xFFFFu; }
  static constexpr int NC = 2 * R + 2;  // register offsets o = 0 .. 2R+1 touch the window
  static constexpr uint32_t konst(int o) { return half(o) | (half(o == 0 ? NK - 1 : o - 1) << 16); }
};
typedef unsigned short ft_u16x2 __attribute__((ext_vector_type(2)));
template <class F, int... I>
__device__ __forceinline__ void ft_unroll(F &f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }

struct FtWinArgs { unsigned long long *fail_rows; };  // [item][ceil(steps / 64)]: bit = some lane of that position failed

template <int R, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_ft_xw(FtArgs a, FtWinArgs pw) {
  using W = FtWin<R>;
  constexpr int NK = W::NK;
  constexpr int P = 16;
  static_assert(2 * R + 2 * P <= 64, "window");
  __shared__ uint32_t win[WAVES][64 * 64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t n = a.n_items;
  uint32_t pm = 0;
  if (32 * lane < a.nx) {
    const int4 *rc = reinterpret_cast<const int4 *>(a.rowcnt + 32 * lane);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int4 v = rc[q];
      const int x = 32 * lane + 4 * q;
      pm |= ((x < a.nx && v.x) ? 1u : 0u) << (4 * q) | ((x + 1 < a.nx && v.y) ? 2u : 0u) << (4 * q) |
            ((x + 2 < a.nx && v.z) ? 4u : 0u) << (4 * q) | ((x + 3 < a.nx && v.w) ? 8u : 0u) << (4 * q);
    }
  }
  uint32_t cst[W::NC];
#pragma unroll
  for (int o = 0; o < W::NC; ++o) {
    cst[o] = W::konst(o);
    asm volatile("" : "+v"(cst[o]));
  }
  uint32_t *mywin = &win[wave][0];
  const uint32_t win_lds = (uint32_t)(size_t)mywin;
  const int steps = a.nx + R, nblk64 = (steps + 63) / 64;
  for (uint32_t it = blockIdx.x * WAVES + wave; it < n; it += gridDim.x * WAVES) {
    const int y = __builtin_amdgcn_readfirstlane((int)(it / (uint32_t)a.nzc)), c = __builtin_amdgcn_readfirstlane((int)(it % (uint32_t)a.nzc));
    const int z = 64 * c + lane;
    const bool act = z < a.nz;
    const int64_t plane = (int64_t)a.ny * a.nz;
    const uint32_t *in = a.inter + (int64_t)y * a.nz + (act ? z : 0);
    char *orow = reinterpret_cast<char *>(a.coc + (int64_t)y * a.nz - (int64_t)R * plane);
    const uint32_t ooff = (uint32_t)(act ? z : 0) * (uint32_t)sizeof(vox_t);
    uint32_t kreg[NK / 2];
#pragma unroll
    for (int j = 0; j < NK / 2; ++j) kreg[j] = 0xFFFFFFFFu;
    auto issue = [&](const int xb) {
#pragma unroll
      for (int u = 0; u < P; ++u) {
        const int xx = min(xb + u, a.nx - 1);
        const uint32_t *ptr = in + (int64_t)xx * plane;
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, off" : : "s"(win_lds + (uint32_t)(((xb + u) & 63) * 256)), "v"(ptr) : "memory", "m0");
      }
    };
    issue(0);
    uint32_t tw[P];
    unsigned long long failbits = 0;  // (wave-uniform) positions of the current 64 steps where some lane's window did not suffice
    for (int x0 = 0; x0 < steps; x0 += NK) {
      const uint32_t hi = (uint32_t)(x0 & 32);  // slot of plane x = (x & 31) | hi
      const uint32_t *wbase = mywin + hi * 64 + lane;
      auto step = [&](auto xs_tag) {
        constexpr int XS = decltype(xs_tag)::value;
        const int x = x0 + XS;
        if ((XS & (P - 1)) == 0) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
          for (int u = 0; u < P; ++u) tw[u] = wbase[(XS + u) * 64];   // (XS + u < 32: the batch does not wrap inside a block)
          if (x + P < a.nx) issue(x + P);
        }
        const uint32_t w = tw[XS & (P - 1)];
        const int dy = y - (int)((w >> 10) & 1023u), dz = z - (int)(w & 1023u);
        const int f = ft::mul24(dy, dy) + ft::mul24(dz, dz);
        const bool has = (__builtin_amdgcn_readlane(pm, (x >> 5) & 63) >> (x & 31)) & 1u;
        const uint32_t key = ((uint32_t)(has ? min(f, W::THR) : W::THR) << 6) | ((uint32_t)XS | hi);
        if (XS & 1)
          kreg[XS >> 1] = (kreg[XS >> 1] & 0xFFFFu) | (key << 16);
        else
          kreg[XS >> 1] = (kreg[XS >> 1] & 0xFFFF0000u) | key;
        const int p = x - R;
        ft_u16x2 acc[4] = {{0xFFFF, 0xFFFF}, {0xFFFF, 0xFFFF}, {0xFFFF, 0xFFFF}, {0xFFFF, 0xFFFF}};
#pragma unroll
        for (int j = 0; j < NK / 2; ++j) {
          const int o = (XS - 2 * j) & (NK - 1);
          if (o < W::NC) {
            const ft_u16x2 t = __builtin_elementwise_add_sat(__builtin_bit_cast(ft_u16x2, kreg[j]), __builtin_bit_cast(ft_u16x2, cst[o]));
            acc[j & 3] = __builtin_elementwise_min(acc[j & 3], t);
          }
        }
        const ft_u16x2 am = __builtin_elementwise_min(__builtin_elementwise_min(acc[0], acc[1]), __builtin_elementwise_min(acc[2], acc[3]));
        const uint32_t best = min((uint32_t)am.x, (uint32_t)am.y);
        const uint32_t slot = best & 63u;
        const int pb = p - R;
        const int q = pb + (int)((slot - (uint32_t)pb) & 63u);
        const uint32_t tagw = mywin[slot * 64 + lane];
        const bool inside = act & ((unsigned)p < (unsigned)a.nx);
        if (inside) *reinterpret_cast<vox_t *>(orow + ooff) = ((uint32_t)q << 20) | (tagw & 0xFFFFFu);
        orow += plane * (int64_t)sizeof(vox_t);
        const bool fail = inside & ((best >> 6) >= (uint32_t)W::THR);
        failbits |= (ft_vote(fail) ? 1ull : 0ull) << (x & 63);
        __builtin_amdgcn_sched_barrier(0);
      };
      ft_unroll(step, std::make_integer_sequence<int, NK>{});
      if ((x0 & 32) || x0 + NK >= steps) {  // 64 steps (or the last ones) are done: their flags out
        if (lane == 0) pw.fail_rows[(size_t)it * nblk64 + (x0 >> 6)] = failbits;
        failbits = 0;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}
}  // namespace fiesta
