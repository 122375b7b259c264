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

// An envelope ENTRY is (q, f, tag): position along the column, height, and an opaque tag that comes back when the entry
// wins a position (pass A: the z of the site; pass B: its packed (y', z')).  In the ring an entry is two words,
//     e1 = f << SB | start        (start: the first position the entry wins; SB = 11, or 12 for columns up to 2048)
//     e2 = q << QSH | tag         -- the OUTPUT WORD itself: what the kernels store when the entry wins a position
// so that a pop is ONE 8-byte LDS read and three shifts, and an emission needs no packing at all (r02's layout
// f << 11 | q, tag << 12 | start re-assembled the output word per position and cached five fields of the bottom entry).
// Ring:   void get(int c, uint32_t &e1, uint32_t &e2), void set(int c, uint32_t e1, uint32_t e2), static int kStep:
//         c is a MONOTONE counter that advances by kStep per entry (the ring reduces it to a slot itself): an LDS ring
//         counts in bytes, so that a slot address is one and-or of the counter (ft_kernels.hpp: LdsRing).
//         void bget(int c, ...), void bset(int c, ...): the BACKING STORE behind the ring, one slot per counter value (in
//         the kernels: global memory) -- see "a deque deeper than its ring" below.
//
// The operations are written for a WAVE that runs 64 envelopes in lock-step: no data-dependent branch inside -- every
// lane executes every instruction, lanes that have nothing to do pass `doit = false` and get their state back through
// selects -- and the loops around them (pop until no lane wants to, emit while every lane is final) are decided by
// wave votes in the caller.  What a wave executes per step is therefore the MAXIMUM over its lanes, not the mean: with 64
// lanes some lane advances its bottom entry at 98 % of the positions and some lane pops at every other site (measured on
// config 2's scene), so the operations that used to sit behind a vote "because most lanes skip them" ran nearly always --
// r03 makes the common ones unconditional and cheap instead:
//   * the bottom entry follows the emission point by ONE ring read, one compare and two selects per position (step_to),
//     no vote and no branch (r02: a vote per position and a 20-instruction advance behind it);
//   * place() needs no "does it win anywhere inside the column" test -- the start is clamped to the column length, an
//     entry that starts there simply never wins -- and, on columns up to 1024, no fix-up of the division (see start_of).
// QSH: bit position of q in the output word (pass A: 10 or 11, pass B: 20).  LONG: columns up to 2048 positions.
template <int S, class Ring, int QSH, bool LONG = false>
struct LaneEnvelope {
  static_assert((S & (S - 1)) == 0, "ring size must be a power of two");
  static constexpr int SB = LONG ? 12 : 11;               // bits of `start` in e1; f gets the other 32 - SB
  static constexpr uint32_t kSMask = (1u << SB) - 1u;
  static constexpr int K = Ring::kStep;
  Ring r;
  int bot, top;  // live entries are bot..top (monotone counters in units of K, see Ring); empty iff top < bot
  int lo;        // spill mode only: the oldest entry still in the ring; entries bot .. lo - K live in the backing store
  // cached top entry: position q and key = q^2 + f(q), so that cost(p) = p (p - 2 q) + key; t_s = its start
  int t_q, t_key, t_s;
  // cached bottom entry = the winner at the emission point: its output word and height
  uint32_t c_word;
  int c_f;

  FT_HD void init() {
    bot = 0;
    top = -K;
    lo = 0;
    t_q = t_key = t_s = 0;
    c_word = 0;
    c_f = 0;
  }
  FT_HD bool empty() const { return top < bot; }
  FT_HD int depth() const { return (top - bot) / K + 1; }
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
    const int nt = top - K;
    uint32_t e1, e2;
    r.get(nt, e1, e2);  // (below the bottom this is a stale slot: read, not used)
    const bool more = nt >= bot;   // (`&`, not `&&`, here and below: a short circuit becomes an exec-mask region)
    const int nq = (int)(e2 >> QSH);
    const int nk = more ? mul24(nq, nq) + (int)(e1 >> SB) : 0, ns = more ? (int)(e1 & kSMask) : 0;
    top = doit ? nt : top;
    t_q = doit ? nq : t_q;      // (an emptied ring: whatever the stale slot held -- harmless, N = key >= 0 = t_s * D)
    t_key = doit ? nk : t_key;  // (the ring ran empty: 0, see wants_pop)
    t_s = doit ? ns : t_s;
  }
  // floor(N / (2 d)) + 1 for the start of a newcomer d positions beyond the top, N = key - t_key >= 0 (no lane wants a
  // pop).  floor(N / 2d) = floor((N >> 1) / d).  Short columns: ONE float multiply decides it -- (M + 1/2) / d lies at
  // least 1 / 2d >= 4.9e-4 away from every integer (d <= 1023), the product with the 1-ulp reciprocal is off by at most
  // 1.8e-7 relative, i.e. < 3.7e-4 up to the quotient 2048, beyond which the start is clamped anyway.  Long columns
  // (d up to 2047) keep the exact integer fix-up.
  static FT_HD int start_of(int N, int d) {
    const int M = N >> 1;
#if defined(__HIP_DEVICE_COMPILE__)
    if (!LONG) return (int)(((float)M + 0.5f) * __builtin_amdgcn_rcpf((float)d)) + 1;
#else
    if (d <= 0) return 0;  // (an empty ring's stale top: the caller replaces the result)
    if (!LONG) return (int)(((float)M + 0.5f) / (float)d) + 1;
#endif
    return floor_div_small(M, d) + 1;
  }
  // n_pos = column length, p_out = the next position to be emitted (everything before it is final and gone)
  FT_HD void place(bool doit, int q, int f, uint32_t word, int key, int n_pos, int p_out) {
    const bool has = top >= bot;
    int s = start_of(key - t_key, q - t_q);  // (an empty ring: garbage in, garbage out, replaced below)
    s = s < n_pos ? s : n_pos;               // clamped: an entry that starts at the end of the column never wins
    s = has ? s : p_out;                     // alone, it owns everything that is not emitted yet
    const bool keep = doit;  // (the caller has made room: full() / evict())
    const int ntop = top + K;
    // every lane stores, no branch around it: the slot after the top is never live (at most S - 1 entries), a lane that
    // keeps nothing just leaves a stale entry there
    r.set(ntop, ((uint32_t)f << SB) | ((uint32_t)s & kSMask), word);
    top = keep ? ntop : top;
    t_q = keep ? q : t_q;
    t_key = keep ? key : t_key;
    t_s = keep ? s : t_s;
  }
  // ---- a deque deeper than its ring.  The ring holds S - 1 entries (one slot stays free for the unconditional store of
  // place).  When a lane's deque outgrows that, its OLDEST ring entry moves to the backing store -- one slot per counter
  // value, so an entry keeps its address for life -- and the lane goes on where it stands: the ring is then the window
  // lo .. top of the deque, bot .. lo - K lie in the backing store and come back one at a time when the emission point
  // (step_to) or, rarely, a run of pops reaches them.  The wave as a whole is in SPILL MODE while any lane has entries
  // out there (a wave-uniform flag in the kernels): only then are the *_sp variants below executed, so a scene whose
  // deques fit their rings pays one vote per batch of sites for all of this.
  // Room is checked once per BATCH of sites (the kernels place up to P sites between two emission runs): a batch that
  // starts with P free slots in every lane's ring needs no check at all; otherwise it runs "carefully" -- spill mode,
  // where every site asks full_sp() first.  (Bare compares: the wave's vote on them is the compare's own mask.)
  FT_HD bool near_full(int P) const { return top - bot >= (S - 1 - P) * K; }  // fewer than P free slots (plain mode)
  FT_HD bool full_sp() const { return top - lo >= (S - 2) * K; }              // place() would need the free slot
  FT_HD void enter_spill() { lo = bot; }                                             // (every lane, spilling or not)
  FT_HD bool spilled() const { return bot < lo; }
  FT_HD void evict(bool doit) {  // the oldest entry of the ring -> backing store (doit => the window is not empty)
    uint32_t e1, e2;
    r.get(lo, e1, e2);
    if (doit) r.bset(lo, e1, e2);
    lo = doit ? lo + K : lo;
  }
  FT_HD void pop_sp(bool doit) {
    const int nt = top - K;
    uint32_t e1, e2;
    r.get(nt, e1, e2);
    const bool more = nt >= bot;
    if (doit & more & (nt < lo)) r.bget(nt, e1, e2);  // the new top was evicted earlier
    const int nq = (int)(e2 >> QSH);
    const int nk = more ? mul24(nq, nq) + (int)(e1 >> SB) : 0, ns = more ? (int)(e1 & kSMask) : 0;
    top = doit ? nt : top;
    t_q = doit ? nq : t_q;
    t_key = doit ? nk : t_key;
    t_s = doit ? ns : t_s;
    lo = (doit & (nt + K < lo)) ? nt + K : lo;  // the window never reaches beyond the slot place() writes next
  }
  FT_HD void reload_bottom_sp() {
    uint32_t e1, e2;
    r.get(bot, e1, e2);
    const bool has = top >= bot;
    if (has & (bot < lo)) r.bget(bot, e1, e2);
    c_word = has ? e2 : c_word;
    c_f = has ? (int)(e1 >> SB) : c_f;
  }
  FT_HD void step_to_sp(int p) {
    const int nb = bot + K;
    uint32_t e1, e2;
    r.get(nb, e1, e2);
    if ((nb <= top) & (nb < lo)) r.bget(nb, e1, e2);
    const bool adv = (nb <= top) & ((int)(e1 & kSMask) <= p);
    bot = adv ? nb : bot;
    c_word = adv ? e2 : c_word;
    c_f = adv ? (int)(e1 >> SB) : c_f;
    lo = lo < bot ? bot : lo;  // (a lane whose backing store has run dry: window = deque again)
  }

  // before a run of emissions: the cached bottom from the ring (sites placed since the last run did not maintain it)
  FT_HD void reload_bottom() {
    uint32_t e1, e2;
    r.get(bot, e1, e2);
    const bool has = top >= bot;
    c_word = has ? e2 : c_word;
    c_f = has ? (int)(e1 >> SB) : c_f;
  }

  // ---- emission.  Before position p is emitted the bottom moves on to the entry that wins there: the entry after the
  // bottom takes over iff its start has been reached -- starts are strictly increasing, positions are visited one by one,
  // so one look ahead per position suffices.  Safe although p may not be final yet: whatever later pops the new bottom
  // beats it at its first position <= p, hence beats the released entry at p too -- it never owns p again.
  FT_HD void step_to(int p) {
    const int nb = bot + K;
    uint32_t e1, e2;
    r.get(nb, e1, e2);  // (beyond the top: a stale slot, read and ignored)
    const bool adv = (nb <= top) & ((int)(e1 & kSMask) <= p);
    bot = adv ? nb : bot;
    c_word = adv ? e2 : c_word;
    c_f = adv ? (int)(e1 >> SB) : c_f;
  }
  // Is the winner at position p settled, given that every site still to come lies at x_next or beyond (p < x_next)?
  // Finality is MONOTONE: sqrt(cost(.)) is 1-Lipschitz (a minimum of 1-Lipschitz functions), so a position that is
  // final makes every position before it final as well.  And judged by the bottom entry BEFORE stepping, the test is
  // merely conservative (the bottom's parabola lies on or above the envelope).  Together: final_at(p + k, x_next) on the
  // un-stepped bottom settles p .. p + k at once -- the emission loops use it to skip k of k + 1 finality votes.
  // While the ring is empty the cached bottom reads c_f = +kNeverFinal (set_idle(false), the start of every column) --
  // no position is final; a lane that carries no column reads -kNeverFinal (set_idle(true)) -- it never holds up the
  // wave's vote.  One compare, its mask is the vote.
  static constexpr int kNeverFinal = (1 << 30) + 1;  // above any (x_next - p)^2, and p^2 on top still fits an int
  FT_HD void set_idle(bool idle) {
    c_f = idle ? -kNeverFinal : kNeverFinal;
    c_word = 0;
  }
  FT_HD int winner_q() const { return (int)(c_word >> QSH); }
  FT_HD uint32_t winner_word() const { return c_word; }
  FT_HD int winner_cost(int p) const {
    const int dq = p - winner_q();
    return mul24(dq, dq) + c_f;
  }
  FT_HD bool final_at(int p, int x_next) const {
    const int dx = x_next - p;
    return dx * dx >= winner_cost(p);
  }
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
