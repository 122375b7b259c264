// tests/cpp/ft_model.cpp -- TEST INFRASTRUCTURE: host model of the bulk feature-transform kernels.
//
// Compiles fiesta_amd/csrc/ft_core.hpp (the per-lane envelope machine the HIP kernels instantiate with LDS rings) with
// g++ and drives it exactly the way ft_kernels.hpp does -- 64-lane "waves", lock-step emission, a ring that spills
// its oldest entries into a backing store when a deque outgrows it (the wave's spill mode) -- so that the integer logic
// is checked against brute force on a machine without a GPU
// (tests/test_ft_model.py).  Not part of the product: libfiesta_hip.so never links this file.
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../fiesta_amd/csrc/ft_core.hpp"

using namespace fiesta::ft;

namespace {
constexpr uint32_t kNone = 0x80000000u;
constexpr int W = 64;  // lanes per wave

template <int S>
struct VecRing {  // two words per entry, as the LDS ring of ft_kernels.hpp; counters advance by kStep (there: bytes)
  static constexpr int kStep = 8;
  uint32_t *e;
  uint32_t *back;  // backing store: one slot per counter value (there: global memory)
  void bget(int c, uint32_t &e1, uint32_t &e2) const { e1 = back[2 * (c / kStep)], e2 = back[2 * (c / kStep) + 1]; }
  void bset(int c, uint32_t e1, uint32_t e2) { back[2 * (c / kStep)] = e1, back[2 * (c / kStep) + 1] = e2; }
  void get(int c, uint32_t &e1, uint32_t &e2) const {
    const int i = (c / kStep) & (S - 1);
    e1 = e[2 * i], e2 = e[2 * i + 1];
  }
  void set(int c, uint32_t e1, uint32_t e2) {
    const int i = (c / kStep) & (S - 1);
    e[2 * i] = e1, e[2 * i + 1] = e2;
  }
};

struct Model {
  int nx, ny, nz, nzw;
  const uint32_t *bits;  // [nx][ny][nzw]
  std::vector<uint16_t> rowlist;  // [nx][ny]: the non-empty rows of plane x, ascending
  std::vector<int> rowcnt;        // [nx]
  std::vector<uint32_t> inter;    // [nx][ny][nz]: y' << 10 | z' of the in-plane nearest site (planes with sites only)
  int max_depth = 0;
  int spilled_items = 0, evictions = 0;
  bool wide = false;  // the kernels' WIDE site packing (regions up to 2048 per axis, ids reach 512 voxels): pass B then emits
                      // d^2 (0x7FFFFFFF: nothing in reach) instead of a packed site

  // the wave-level steps the two passes share, as the kernels do them: pop (any lane), make room, place; emission runs
  template <class Env>
  void pop_all(Env *env, const bool *want, bool sp) {
    for (int k = 0; k < W; ++k) sp ? env[k].pop_sp(want[k]) : env[k].pop(want[k]);
  }
  // a batch of up to P sites begins: plain if every lane has P free slots and nothing is out in the backing store
  template <class Env>
  bool careful_batch(Env *env, int P, bool &sp, bool &item_spilled) {
    bool any = sp;
    if (!any)
      for (int k = 0; k < W; ++k) any = any || env[k].near_full(P);
    if (!any) return false;
    if (!sp)
      for (int k = 0; k < W; ++k) env[k].enter_spill();
    if (!item_spilled) ++spilled_items;
    item_spilled = true;
    return true;
  }
  template <class Env>
  void make_room(Env *env) {  // (careful batches only)
    bool any = false, full[W];
    for (int k = 0; k < W; ++k) any = (full[k] = env[k].full_sp()) || any;
    if (!any) return;
    for (int k = 0; k < W; ++k) {
      env[k].evict(full[k]);
      evictions += full[k];
    }
  }
  template <class Env, class Emit>
  void drain_run(Env *env, int &p_out, int n_pos, int x_next, const bool careful, bool &sp, Emit emit) {
    for (int k = 0; k < W; ++k) careful ? env[k].reload_bottom_sp() : env[k].reload_bottom();
    while (p_out + 3 < n_pos && p_out + 3 < x_next) {  // four positions per finality vote (ft_core.hpp: monotone)
      bool all4 = true;
      for (int k = 0; k < W; ++k) all4 = all4 && (env[k].final_at(p_out + 3, x_next));
      if (!all4) break;
      for (int j = 0; j < 4; ++j) {
        for (int k = 0; k < W; ++k) careful ? env[k].step_to_sp(p_out) : env[k].step_to(p_out);
        emit();
        ++p_out;
      }
    }
    while (p_out < n_pos && p_out < x_next) {
      bool all = true;
      for (int k = 0; k < W; ++k) careful ? env[k].step_to_sp(p_out) : env[k].step_to(p_out);
      for (int k = 0; k < W; ++k) all = all && (env[k].final_at(p_out, x_next));
      if (!all) break;
      emit();
      ++p_out;
    }
    if (careful) {  // back to the plain ring once no lane has anything left in the backing store
      bool any = false;
      for (int k = 0; k < W; ++k) any = any || env[k].spilled();
      sp = any;
    }
  }

  template <int S, bool WIDE>
  bool plane_item(int x, int c) {  // pass A for plane x, lanes z = 64 c + k
    std::vector<uint32_t> rs(2 * S * W), bk((size_t)2 * (ny + 2) * W);
    LaneEnvelope<S, VecRing<S>, WIDE ? 11 : 10, WIDE> env[W];  // pass A: q = y', f = (z - z')^2, tag = z'
    bool act[W];
    for (int k = 0; k < W; ++k) {
      env[k].r = VecRing<S>{&rs[2 * k * S], &bk[(size_t)2 * k * (ny + 2)]};
      env[k].init();
      act[k] = 64 * c + k < nz;
      env[k].set_idle(!act[k]);
    }
    const int cnt = rowcnt[x];
    int p_out = 0;
    bool sp = false, item_spilled = false, careful = false;
    auto emit = [&]() {
      for (int k = 0; k < W; ++k)
        if (act[k]) inter[((size_t)x * ny + p_out) * nz + 64 * c + k] = env[k].winner_word();
    };
    for (int i = 0; i < cnt; ++i) {
      if ((i & 7) == 0) careful = careful_batch(env, 8, sp, item_spilled);  // (the kernel: rows between two emission runs)
      const int yr = rowlist[(size_t)x * ny + i];
      const uint32_t *row = bits + ((size_t)x * ny + yr) * nzw;
      unsigned long long chunk = row[2 * c];
      if (2 * c + 1 < nzw) chunk |= (unsigned long long)row[2 * c + 1] << 32;
      int left_out = -1, right_out = -1;
      for (int w = 0; w < 2 * c && w < nzw; ++w)
        if (row[w]) left_out = 32 * w + 31 - __builtin_clz(row[w]);
      for (int w = nzw - 1; w >= 2 * c + 2; --w)
        if (row[w]) right_out = 32 * w + __builtin_ctz(row[w]);
      uint32_t tag[W];
      int f[W], key[W];
      for (int k = 0; k < W; ++k) {
        int d;
        tag[k] = (uint32_t)nearest_in_row(chunk, 64 * c, k, left_out, right_out, d);
        if (wide && d > 1023) d = 1023;  // (out of an id's reach anyway; keeps f inside its 21 bits)
        f[k] = d * d;
        key[k] = yr * yr + f[k];
      }
      for (;;) {  // pop while any lane wants to (wave vote), then place
        bool any = false;
        bool want[W];
        for (int k = 0; k < W; ++k) any = (want[k] = env[k].wants_pop(yr, key[k])) || any;
        if (!any) break;
        pop_all(env, want, careful);
      }
      if (careful) make_room(env);
      for (int k = 0; k < W; ++k) {
        env[k].place(act[k], yr, f[k], ((uint32_t)yr << (WIDE ? 11 : 10)) | tag[k], key[k], ny, p_out);
        if (act[k] && env[k].depth() > max_depth) max_depth = env[k].depth();
      }
      // the kernel emits every eighth site row of a staged batch of 16 (and after the last row)
      if ((i & 7) == 7 || i + 1 == cnt) drain_run(env, p_out, ny, i + 1 < cnt ? (int)rowlist[(size_t)x * ny + i + 1] : kFarAhead, careful, sp, emit);
    }
    return p_out == ny;
  }

  template <int S, bool WIDE>
  bool column_item(int y, int c, uint32_t *out) {  // pass B for row y, lanes z = 64 c + k
    std::vector<uint32_t> rs(2 * S * W), bk((size_t)2 * (nx + 2) * W);
    LaneEnvelope<S, VecRing<S>, 20, WIDE> env[W];  // pass B: q = x', f = (y - y')^2 + (z - z')^2, tag = y' << 10 | z'
    bool act[W];
    for (int k = 0; k < W; ++k) {
      env[k].r = VecRing<S>{&rs[2 * k * S], &bk[(size_t)2 * k * (nx + 2)]};
      env[k].init();
      act[k] = 64 * c + k < nz;
      env[k].set_idle(!act[k]);
    }
    int p_out = 0;
    bool sp = false, item_spilled = false, careful = false;
    bool no_site[W] = {};
    auto word = [&](int k) -> uint32_t {  // what the kernel's emit() stores (wide: as a squared distance)
      if (!wide) return env[k].winner_word();
      const int cost = env[k].winner_cost(p_out);
      return (no_site[k] || cost >= (1 << 18)) ? 0x7FFFFFFFu : (uint32_t)cost;
    };
    auto emit = [&]() {
      for (int k = 0; k < W; ++k)
        if (act[k]) out[((size_t)p_out * ny + y) * nz + 64 * c + k] = word(k);
    };
    auto drain = [&](int x_next) { drain_run(env, p_out, nx, x_next, careful, sp, emit); };
    int last = -1;
    for (int x = 0; x < nx; ++x)
      if (rowcnt[x]) last = x;
    for (int x = 0; x <= last; ++x) {
      if ((x & 7) == 0) careful = careful_batch(env, 8, sp, item_spilled);  // (the kernel: a batch of 8 planes)
      if (rowcnt[x]) {
        uint32_t tag[W];
        int f[W], key[W];
        bool use[W];
        int pkey[W];
        for (int k = 0; k < W; ++k) {
          const uint32_t w = act[k] ? inter[((size_t)x * ny + y) * nz + 64 * c + k] : 0u;
          use[k] = act[k];
          int dy, dz;
          if (wide) {  // offsets from the column as signed 10-bit fields; out of an id's reach: no candidate
            dy = (int)(w >> 11) - y, dz = (int)(w & 2047u) - (64 * c + k);
            use[k] = use[k] && (unsigned)(dy + 511) < 1023u && (unsigned)(dz + 511) < 1023u;
            tag[k] = (((uint32_t)dy & 1023u) << 10) | ((uint32_t)dz & 1023u);
            if (!use[k]) dy = dz = 0;
          } else {
            tag[k] = w & 0xFFFFFu;
            dy = y - (int)(w >> 10), dz = 64 * c + k - (int)(w & 1023u);
          }
          f[k] = act[k] ? dy * dy + dz * dz : 0;
          key[k] = env[k].key_of(x, f[k]);
          pkey[k] = (wide && !use[k]) ? env[k].kNoPop : key[k];
        }
        for (;;) {
          bool any = false;
          bool want[W];
          for (int k = 0; k < W; ++k) any = (want[k] = env[k].wants_pop(x, pkey[k])) || any;
          if (!any) break;
          pop_all(env, want, careful);
        }
        if (careful) make_room(env);
        for (int k = 0; k < W; ++k) {
          env[k].place(use[k], x, f[k], ((uint32_t)x << 20) | tag[k], key[k], nx, p_out);
          if (act[k] && env[k].depth() > max_depth) max_depth = env[k].depth();
        }
      }
      // the kernel emits once per batch of 8 planes (and after the last plane)
      if (x == last) {
        bool any_site = false;
        for (int k = 0; k < W; ++k) any_site = any_site || (act[k] && !env[k].empty());
        if (!any_site) {  // (wide: every site out of reach for the whole wave)
          for (int p = p_out; p < nx; ++p)
            for (int k = 0; k < W; ++k)
              if (act[k]) out[((size_t)p * ny + y) * nz + 64 * c + k] = wide ? 0x7FFFFFFFu : kNone;
          return true;
        }
        for (int k = 0; k < W; ++k)
          if (wide && act[k] && env[k].empty()) no_site[k] = true, env[k].set_idle(true);
        drain(kFarAhead);
      } else if ((x & 7) == 7)
        drain(x + 1);
    }
    if (last < 0) {
      for (int p = 0; p < nx; ++p)
        for (int k = 0; k < W; ++k)
          if (act[k]) out[((size_t)p * ny + y) * nz + 64 * c + k] = wide ? 0x7FFFFFFFu : kNone;
      return true;
    }
    return p_out == nx;
  }
};

template <int S>
int run_all(Model &m, uint32_t *out, const std::vector<int> &items_a, const std::vector<int> &items_b) {
  const int nzc = (m.nz + 63) / 64;
  for (int it : items_a)
    if (!(m.wide ? m.plane_item<S, true>(it / nzc, it % nzc) : m.plane_item<S, false>(it / nzc, it % nzc))) return -2;
  for (int it : items_b)
    if (!(m.wide ? m.column_item<S, true>(it / nzc, it % nzc, out) : m.column_item<S, false>(it / nzc, it % nzc, out))) return -3;
  return 0;
}
}  // namespace

// occ: nx*ny*nz bytes (x-major, z fastest).  out: packed closest site x<<20|y<<10|z, or 0x80000000 when there is no
// site at all.  S0: ring size (4, 8, 16, 32 or 64; a ring of S holds S - 1 entries, deeper deques spill into the backing store).
// stats[0] = deepest deque seen, stats[1] = items (pass A + pass B) that went through spill mode, stats[2] = entries evicted.
static int model_run(const uint8_t *occ, int nx, int ny, int nz, int S0, uint32_t *out, int *stats, bool wide);
extern "C" int ft_model_run(const uint8_t *occ, int nx, int ny, int nz, int S0, uint32_t *out, int *stats) {
  if (nx > 1024 || ny > 1024 || nz > 1024) return -1;
  return model_run(occ, nx, ny, nz, S0, out, stats, false);
}
// The WIDE packing of the kernels (regions up to 2048 per axis): out receives SQUARED DISTANCES, 0x7FFFFFFF where no
// occupied voxel lies within the reach of an id (d^2 >= 2^18).
extern "C" int ft_model_run_wide(const uint8_t *occ, int nx, int ny, int nz, int S0, uint32_t *out, int *stats) {
  if (nx > 2048 || ny > 2048 || nz > 2048) return -1;
  return model_run(occ, nx, ny, nz, S0, out, stats, true);
}
static int model_run(const uint8_t *occ, int nx, int ny, int nz, int S0, uint32_t *out, int *stats, bool wide) {
  Model m;
  m.wide = wide;
  m.nx = nx, m.ny = ny, m.nz = nz, m.nzw = (nz + 31) / 32;
  std::vector<uint32_t> bits((size_t)nx * ny * m.nzw, 0u);
  for (size_t i = 0; i < (size_t)nx * ny * nz; ++i)
    if (occ[i]) {
      const size_t row = i / nz;
      const int z = (int)(i % nz);
      bits[row * m.nzw + (z >> 5)] |= 1u << (z & 31);
    }
  m.bits = bits.data();
  m.rowlist.assign((size_t)nx * ny, 0);
  m.rowcnt.assign(nx, 0);
  for (int x = 0; x < nx; ++x)
    for (int y = 0; y < ny; ++y) {
      bool any = false;
      for (int w = 0; w < m.nzw; ++w) any = any || bits[((size_t)x * ny + y) * m.nzw + w];
      if (any) m.rowlist[(size_t)x * ny + m.rowcnt[x]++] = (uint16_t)y;
    }
  m.inter.assign((size_t)nx * ny * nz, 0xFFFFFFFFu);
  const int nzc = (nz + 63) / 64;
  std::vector<int> ia, ib;
  for (int x = 0; x < nx; ++x)
    if (m.rowcnt[x])
      for (int c = 0; c < nzc; ++c) ia.push_back(x * nzc + c);
  for (int y = 0; y < ny; ++y)
    for (int c = 0; c < nzc; ++c) ib.push_back(y * nzc + c);
  int r;
  switch (S0) {
    case 4: r = run_all<4>(m, out, ia, ib); break;
    case 8: r = run_all<8>(m, out, ia, ib); break;
    case 16: r = run_all<16>(m, out, ia, ib); break;
    case 32: r = run_all<32>(m, out, ia, ib); break;
    default: r = run_all<64>(m, out, ia, ib); break;
  }
  stats[0] = m.max_depth;
  stats[1] = m.spilled_items;
  stats[2] = m.evictions;
  return r;
}
