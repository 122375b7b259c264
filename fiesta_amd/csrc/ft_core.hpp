// fiesta_amd/csrc/ft_core.hpp -- per-lane machinery of the BULK UpdateESDF path (ft_kernels.hpp): the streaming
// lower envelope of parabolas of one grid column, and the nearest set bit of a bitmap row.
//
// Why this exists.  On a fully observed map the fixed point of the reference's 24-neighbour propagation
// (src/ESDFMap.cpp:339-392) is the exact Euclidean feature transform of the occupied set -- up to a handful of voxels
// per 10^5 where the reference itself keeps a slightly larger distance that depends on the order of the inserts
// (pinned on 2 x 134 M voxels by tests/golden/c2_512_*_digest.npz).  When an update touches a large
// part of such a map, recomputing the whole transform with three separable passes costs a few coalesced sweeps over
// the grid -- far less than pushing a wave front through every tile.  The frontier rounds (relax_kernels.hpp) remain
// the engine for partially observed maps, windows and small deltas; DenseMap chooses per update (dense_map.hip:
// update_esdf).
//
// The one-dimensional problem.  Sites arrive in increasing position q with a height f(q) >= 0 (the squared distance
// already accumulated along the other axes); wanted is, for every integer position p of the column, a site minimising
// (p-q)^2 + f(q).  LaneEnvelope keeps the lower envelope as a deque of (site, start) entries -- entry i is the winner
// on [start_i, start_{i+1}) -- exactly Meijster's integer formulation (no real-valued intersections), and emits
// positions from the bottom as soon as they are FINAL: no site that can still arrive (position >= x_next) can beat
// the current winner at p once (x_next - p)^2 >= cost(p).  So the deque only holds the entries between the emission
// point and the newest site: a few dozen for obstacle spacings of tens of voxels, whatever the column length.
//
// The code is host/device neutral: tests/cpp/ft_model.cpp compiles it with g++ and checks it against brute force
// (tests/test_ft_model.py, no GPU needed); the kernels instantiate it with LDS rings.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FT_HD __host__ __device__ __forceinline__
#else
#define FT_HD inline
#endif

namespace fiesta {
namespace ft {

// Every product on this path has factors below 2^23 (positions, position differences and 2 x those, quotients): the
// full-rate 24-bit multiplier does them; v_mul_lo_u32 is a quarter-rate instruction on gfx950.
FT_HD int mul24(int a, int b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __mul24(a, b);
#else
  return a * b;
#endif
}

constexpr int kNoStart = 0x7FFFFFFF;
constexpr int kFarAhead = 1 << 15;  // "no further site will arrive": (kFarAhead - p)^2 still fits an int32

// floor(N / D) for D > 0 and 0 <= N < 4096 * D, both below 2^24: float estimate + exact integer fix-up.
FT_HD int floor_div_small(int N, int D) {
#if defined(__HIP_DEVICE_COMPILE__)
  int q = (int)((float)N * __builtin_amdgcn_rcpf((float)D));
#else
  int q = (int)((float)N / (float)D);
#endif
  const int qd = mul24(q, D);
  if (qd > N)
    --q;
  else if (qd + D <= N)
    ++q;
  return q;
}

// An envelope ENTRY is (q, f, tag): position along the column, height, and an opaque 20-bit tag that comes back when the
// entry wins a position (pass A: the z of the site; pass B: its packed (y', z')).  In the ring an entry is two words,
//     e1 = f << 11 | q            (q < 2048, f < 2^21)
//     e2 = tag << 12 | start      (start < 4096: the first position the entry wins)
// so that a pop or an advance is ONE 8-byte LDS read and a handful of shifts -- the first layout (packed site + 16-bit
// start) re-derived q and f from the site's coordinates on every pop, a third of the instructions of a step.
// Ring:   void get(int i, uint32_t &e1, uint32_t &e2), uint32_t second(int i), void set(int i, uint32_t e1, uint32_t e2)
//         (i already < S)
//
// The operations are written for a WAVE that runs 64 envelopes in lock-step: no data-dependent branch inside -- every
// lane executes every instruction, lanes that have nothing to do pass `doit = false` and get their state back through
// selects -- and the loops around them (pop until no lane wants to, emit while every lane is final) are decided by
// wave votes in the caller.  On the GPU that keeps the control flow scalar (the first version, with per-lane `while`
// and `if`, compiled to ~700 instructions per step, most of them exec-mask bookkeeping; this form needs ~150).
constexpr int kQBits = 11, kStartBits = 12;
template <int S, class Ring>
struct LaneEnvelope {
  static_assert((S & (S - 1)) == 0, "ring size must be a power of two");
  Ring r;
  int bot, top;  // live entries are bot..top (monotone counters, ring slot = counter & (S-1)); empty iff top < bot
  // cached entries: position q and key = q^2 + f(q), so that cost(p) = p (p - 2 q) + key
  int t_q, t_key, t_s;  // top entry
  uint32_t t_tag;
  int c_q, c_key;       // entry `bot` = the winner at the emission point
  uint32_t c_tag;
  int n_s;  // start of entry `bot + 1` (kNoStart: there is none)
  bool overflow;

  FT_HD void init() {
    bot = 0;
    top = -1;
    t_tag = c_tag = 0;
    t_q = t_key = t_s = c_q = c_key = 0;
    n_s = kNoStart;
    overflow = false;
  }
  FT_HD bool empty() const { return top < bot; }
  FT_HD int depth() const { return top - bot + 1; }
  static FT_HD int key_of(int q, int f) { return mul24(q, q) + f; }

  // ---- a new site at position q (beyond every site pushed before), key = q^2 + f: pop while any lane wants to,
  // then place.  The newcomer beats the top strictly at p  <=>  p * D > N  <=>  p >= floor(N / D) + 1.
  // An EMPTY ring keeps t_s = t_key = 0 (init, pop): then N = key >= 0 is never below t_s * D = 0 and the one compare
  // answers "no" by itself -- the vote on it is the compare's own lane mask, no select-and-recompare in between.  A lane
  // that must not pop for another reason passes a key of kNoPop.
  static constexpr int kNoPop = 1 << 30;
  FT_HD bool wants_pop(int q, int key) const {
    const int D = 2 * (q - t_q), N = key - t_key;
    return N < mul24(t_s, D);  // ... already at the top's first position: the top wins nowhere
  }
  FT_HD void pop(bool doit) {
    const int nt = top - 1;
    uint32_t e1, e2;
    r.get(nt & (S - 1), e1, e2);  // (below the bottom this is a stale slot: read, not used)
    const bool more = nt >= bot, ld = doit & more;  // (`&`, not `&&`, here and below: a short circuit becomes an
    top = doit ? nt : top;                            //  exec-mask region with its scalar bookkeeping)
    const int nq = (int)(e1 & ((1u << kQBits) - 1u));
    const int nk = more ? mul24(nq, nq) + (int)(e1 >> kQBits) : 0, ns = more ? (int)(e2 & ((1u << kStartBits) - 1u)) : 0;
    t_q = ld ? nq : t_q;
    t_key = doit ? nk : t_key;  // (the ring ran empty: 0, see wants_pop)
    t_s = doit ? ns : t_s;
    t_tag = ld ? e2 >> kStartBits : t_tag;
  }
  // n_pos = column length, p_out = the next position to be emitted (everything before it is final and gone)
  template <bool BOTTOM = true>
  FT_HD void place(bool doit, int q, int f, uint32_t tag, int key, int n_pos, int p_out) {
    const bool has = top >= bot;
    const int D = has ? 2 * (q - t_q) : 2, N = key - t_key;
    const bool inside = !has | (N < mul24(n_pos, D));  // else it beats the top only beyond the last position
    const int sq = floor_div_small((has & inside) ? N : 0, D) + 1;  // (no lane wants a pop: N >= t_s * D >= 0)
    const int s = has ? sq : p_out;  // alone, it owns everything that is not emitted yet
    bool keep = doit & inside;
    const bool ovf = keep & (top - bot + 1 >= S - 1);  // one slot stays free, see below
    overflow = overflow | ovf;
    keep = keep & !ovf;
    const int ntop = top + 1;
    // every lane stores, no branch around it: the slot after the top is never live (at most S - 1 entries), a lane that
    // keeps nothing just leaves a stale entry there
    r.set(ntop & (S - 1), ((uint32_t)f << kQBits) | (uint32_t)q, (tag << kStartBits) | (uint32_t)s);
    top = keep ? ntop : top;
    t_tag = keep ? tag : t_tag;
    t_q = keep ? q : t_q;
    t_key = keep ? key : t_key;
    t_s = keep ? s : t_s;
    // the cached bottom follows pops and the push -- unless the caller emits only every few sites and reloads it then
    if (BOTTOM) {
      const bool one = top == bot;
      c_tag = one ? t_tag : c_tag;
      c_q = one ? t_q : c_q;
      c_key = one ? t_key : c_key;
      n_s = top == bot + 1 ? t_s : (top <= bot ? kNoStart : n_s);
    }
  }
  // after a run of place<false>() calls: the cached bottom from the ring (two LDS reads instead of five selects per site)
  FT_HD void reload_bottom() {
    uint32_t e1, e2;
    r.get(bot & (S - 1), e1, e2);
    const uint32_t e2n = r.second((bot + 1) & (S - 1));
    const bool has = top >= bot;
    const int nq = (int)(e1 & ((1u << kQBits) - 1u));
    c_q = has ? nq : c_q;
    c_key = has ? mul24(nq, nq) + (int)(e1 >> kQBits) : c_key;
    c_tag = has ? e2 >> kStartBits : c_tag;
    n_s = top > bot ? (int)(e2n & ((1u << kStartBits) - 1u)) : kNoStart;
  }

  // ---- emission.  Before position p is judged, the bottom moves on to the entry that wins there (advance): that is
  // safe although p may not be final yet -- whatever later pops the new bottom beats it at its first position <= p, hence
  // beats the released entry at p too, so the released entry never owns p again.
  FT_HD bool wants_advance(int p) const { return n_s <= p; }
  FT_HD void advance(bool doit) {
    const int nb = bot + 1;
    uint32_t e1, e2;
    r.get(nb & (S - 1), e1, e2);
    const int rst = (int)(r.second((nb + 1) & (S - 1)) & ((1u << kStartBits) - 1u));
    const int nq = (int)(e1 & ((1u << kQBits) - 1u));
    bot = doit ? nb : bot;
    c_tag = doit ? e2 >> kStartBits : c_tag;
    c_q = doit ? nq : c_q;
    c_key = doit ? mul24(nq, nq) + (int)(e1 >> kQBits) : c_key;
    n_s = doit ? (nb < top ? rst : kNoStart) : n_s;
  }
  // Is the winner at position p settled, given that every site still to come lies at x_next or beyond (p < x_next)?
  // (after advance: the winner is the bottom entry)
  // Finality is MONOTONE: sqrt(cost(.)) is 1-Lipschitz (a minimum of 1-Lipschitz functions), so a position that is
  // final makes every position before it final as well.  And judged by the bottom entry BEFORE advancing, the test is
  // merely conservative (the bottom's parabola lies on or above the envelope).  Together: final_at(p + k, x_next) on the
  // un-advanced bottom settles p .. p + k at once -- the emission loops use it to skip k of k + 1 finality votes.
  // The same folding as in wants_pop: while the ring is empty the cached bottom reads c_q = 0, c_key = +kNeverFinal
  // (set_idle(false), the start of every column) -- no position is final; a lane that carries no column reads
  // -kNeverFinal (set_idle(true)) -- it never holds up the wave's vote.  One compare, its mask is the vote.
  static constexpr int kNeverFinal = (1 << 30) + 1;  // above any (x_next - p)^2, and p^2 on top still fits an int
  FT_HD void set_idle(bool idle) { c_key = idle ? -kNeverFinal : kNeverFinal; }
  FT_HD bool final_at(int p, int x_next) const {
    const int g = mul24(p, p - 2 * c_q) + c_key, dx = x_next - p;
    return dx * dx >= g;
  }
  FT_HD int winner_q() const { return c_q; }
  FT_HD uint32_t winner_tag() const { return c_tag; }
  FT_HD int winner_cost(int p) const { return mul24(p, p - 2 * c_q) + c_key; }
};

// Nearest set bit of a bitmap row to position z, for the 64 positions [cbase, cbase + 64) that share the 64-bit chunk
// `chunk` (bit k = position cbase + k).  left_out / right_out: nearest set bit of the row below cbase / at or beyond
// cbase + 64, or -1 if there is none.  The row holds at least one bit.  Returns the position; d receives |z - pos|.
FT_HD int nearest_in_row(unsigned long long chunk, int cbase, int k, int left_out, int right_out, int &d) {
  const unsigned long long le = chunk & ((2ull << k) - 1ull), ge = chunk & (~0ull << k);
  int left = left_out, right = right_out;
#if defined(__HIP_DEVICE_COMPILE__)
  if (le) left = cbase + 63 - __clzll((long long)le);
  if (ge) right = cbase + __ffsll((long long)ge) - 1;
#else
  if (le) left = cbase + 63 - __builtin_clzll(le);
  if (ge) right = cbase + __builtin_ctzll(ge);
#endif
  const int z = cbase + k;
  const int dl = left >= 0 ? z - left : 1 << 12, dr = right >= 0 ? right - z : 1 << 12;
  d = dl <= dr ? dl : dr;
  return dl <= dr ? left : right;
}

}  // namespace ft
}  // namespace fiesta
