// oracle/esdf_port.cpp -- TEST INFRASTRUCTURE ONLY (see oracle/oracle_api.h).
//
// CPU restatement ("port") of the FIESTA hot path, written from the algorithm's description, with
// every function citing the reference file:line it follows (paths are into /root/reference).
// It exists so that the parity checker travels to the GPU box, where /root/reference does not exist.
// PINNING: tests/test_oracle_port_vs_ref.py runs this file and the verbatim-compiled reference
// (oracle/_ref) on identical call sequences and requires every distance, closest-obstacle id,
// occupancy bit, queue size and expansion counter to be identical; tests/golden/*.npz holds outputs of
// the verbatim build that this file must reproduce on any machine.
//
// Storage is a struct-of-flat-arrays keyed by an int slot, like the reference, because the FIFO order
// and the intrusive per-obstacle lists decide which of several equidistant obstacles a voxel ends
// up pointing at; reproducing them makes this port id-exact against the reference, not just
// distance-exact.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <unordered_map>
#include <unordered_set>
#include <algorithm>
#include <vector>

#include "oracle_api.h"

namespace {

constexpr int kUndef = -10000;  // undefined_  (src/ESDFMap.cpp:182)
constexpr int kInf = 10000;     // infinity_   (src/ESDFMap.cpp:181)

struct I3 {
  int x, y, z;
};
inline I3 operator+(I3 a, I3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline bool defined(I3 a) { return a.x != kUndef; }
const I3 kNone = {kUndef, kUndef, kUndef};

// The 24-direction stencil in the reference's order: 6 faces, 12 edges, 6 two-step faces
// (include/parameters.h:54-68). Order matters only for the delete re-seed's "first valid neighbour".
const I3 kDirs[24] = {{-1, 0, 0}, {1, 0, 0},  {0, -1, 0}, {0, 1, 0},  {0, 0, -1}, {0, 0, 1},
                      {-1, -1, 0}, {1, 1, 0}, {0, -1, -1}, {0, 1, 1}, {-1, 0, -1}, {1, 0, 1},
                      {-1, 1, 0},  {1, -1, 0}, {0, -1, 1}, {0, 1, -1}, {1, 0, -1}, {-1, 0, 1},
                      {-2, 0, 0},  {2, 0, 0},  {0, -2, 0}, {0, 2, 0},  {0, 0, -2}, {0, 0, 2}};

struct Item {  // QueueElement (include/ESDFMap.h:39-45)
  I3 p;
  double d;
};

struct Port {
  int mode = 0;  // 0 dense, 1 hash-of-8^3-blocks
  double org[3], res, res_inv;
  // dense geometry (src/ESDFMap.cpp:171-186)
  double lo[3], hi[3];
  int gs[3] = {0, 0, 0}, gs_yz = 0;
  int total = 0;
  // hash geometry (src/ESDFMap.cpp:130-145)
  std::unordered_map<uint64_t, int> blocks;
  int count = 1, reserve = 0;
  std::vector<I3> slot_vox;
  // per-slot state
  std::vector<double> logodds, dist;
  std::vector<int> hits, seen;  // num_hit_, num_miss_ (num_miss_ counts ALL observations, :424)
  std::vector<I3> coc;
  std::vector<int> head, prev, next;
  int undef_slot = 0;  // reserved_idx_4_undefined_
  // queues
  std::deque<Item> q_occ, q_ins, q_del, q_upd;
  // log-odds parameters
  double l_hit = 0, l_miss = 0, l_min = 0, l_max = 0, l_occ = 0;
  // update window
  I3 wmin, wmax, wmin_prev, wmax_prev;
  // fifo probe (relax(): intra-layer dependency depth of the update queue; measurement only)
  struct FifoLayer { int64_t entries, dependent, depth; };
  bool fifo_probe = false;
  std::vector<FifoLayer> fifo_layers;
  std::vector<int> fifo_wlayer, fifo_wdepth;
  // raycast front-end state (include/Fiesta.h:107-110,287)
  std::vector<int> stamp_free, stamp_occ;
  std::unordered_set<int> hstamp_free, hstamp_occ;
  int frame = 0;

  void grow(int n) {  // IncreaseCapacity (src/ESDFMap.cpp:705-730)
    logodds.resize(n, 0.0);
    dist.resize(n, (double)kUndef);
    hits.resize(n, 0);
    seen.resize(n, 0);
    coc.resize(n, kNone);
    slot_vox.resize(n, kNone);
    head.resize(n, kUndef);
    prev.resize(n, kUndef);
    next.resize(n, kUndef);
    reserve = n;
  }

  static uint64_t block_key(int bx, int by, int bz) {
    return ((uint64_t)(uint32_t)(bx + (1 << 20)) << 42) | ((uint64_t)(uint32_t)(by + (1 << 20)) << 21) |
           (uint64_t)(uint32_t)(bz + (1 << 20));
  }

  // FindAndInsert, BLOCK+BITWISE flavour (src/ESDFMap.cpp:732-765): a lookup of a voxel whose 8^3 block
  // does not exist yet appends the whole block (x-major) and records every member's coordinates.
  int slot_in_hash(I3 v) {
    if (count + 512 > reserve) grow(reserve * 2);
    const int within = ((v.x & 7) << 6) + ((v.y & 7) << 3) + (v.z & 7);
    const int bx = v.x >> 3, by = v.y >> 3, bz = v.z >> 3;
    auto it = blocks.find(block_key(bx, by, bz));
    if (it != blocks.end()) return it->second + within;
    blocks.emplace(block_key(bx, by, bz), count);
    for (int i = 0; i < 8; ++i)
      for (int j = 0; j < 8; ++j)
        for (int k = 0; k < 8; ++k) slot_vox[count++] = {(bx << 3) + i, (by << 3) + j, (bz << 3) + k};
    return count - 512 + within;
  }

  int slot(I3 v) {  // Vox2Idx (src/ESDFMap.cpp:84-93)
    if (v.x == kUndef) return undef_slot;
    if (mode == 1) return slot_in_hash(v);
    return v.x * gs_yz + v.y * gs[2] + v.z;
  }
  I3 vox_of(int s) const {  // Idx2Vox (src/ESDFMap.cpp:109-118)
    if (mode == 1) return slot_vox[s];
    return {s / gs_yz, s % gs_yz / gs[2], s % gs[2]};
  }
  I3 pos2vox(const double *p) const {  // Pos2Vox (src/ESDFMap.cpp:74-77)
    return {(int)std::floor((p[0] - org[0]) / res), (int)std::floor((p[1] - org[1]) / res),
            (int)std::floor((p[2] - org[2]) / res)};
  }
  void vox2pos(I3 v, double *p) const {  // Vox2Pos (src/ESDFMap.cpp:79-82)
    p[0] = (v.x + 0.5) * res + org[0];
    p[1] = (v.y + 0.5) * res + org[1];
    p[2] = (v.z + 0.5) * res + org[2];
  }
  bool pos_in_map(const double *p) const {  // PosInMap (src/ESDFMap.cpp:46-61)
    if (mode == 1) return true;
    for (int i = 0; i < 3; ++i)
      if (p[i] < lo[i] || p[i] > hi[i]) return false;
    return true;
  }
  bool in_window(I3 v, bool current = true) const {  // VoxInRange (src/ESDFMap.cpp:63-72)
    const I3 &a = current ? wmin : wmin_prev, &b = current ? wmax : wmax_prev;
    return v.x >= a.x && v.x <= b.x && v.y >= a.y && v.y <= b.y && v.z >= a.z && v.z <= b.z;
  }
  bool occupied(int s) const { return logodds[s] > l_occ; }  // Exist (src/ESDFMap.cpp:16-22)
  double metric(I3 a, I3 b) const {                           // Dist (src/ESDFMap.cpp:122-124)
    const double dx = b.x - a.x, dy = b.y - a.y, dz = b.z - a.z;
    return std::sqrt(dx * dx + dy * dy + dz * dz) * res;
  }

  // Intrusive doubly-linked list of "voxels whose closest obstacle is `owner`"
  // (DeleteFromList / InsertIntoList, src/ESDFMap.cpp:24-42). Insert is push-front.
  void unlink(int owner, int s) {
    if (prev[s] != kUndef)
      next[prev[s]] = next[s];
    else
      head[owner] = next[s];
    if (next[s] != kUndef) prev[next[s]] = prev[s];
    prev[s] = next[s] = kUndef;
  }
  void link_front(int owner, int s) {
    if (head[owner] == kUndef) {
      head[owner] = s;
    } else {
      prev[head[owner]] = s;
      next[s] = head[owner];
      head[owner] = s;
    }
  }

  void full_window() {  // SetOriginalRange (src/ESDFMap.cpp:812-824)
    if (mode == 1) {
      wmin = {-kInf, -kInf, -kInf};
      wmax = {kInf, kInf, kInf};
    } else {
      wmin = {0, 0, 0};
      wmax = {gs[0] - 1, gs[1] - 1, gs[2] - 1};
    }
    wmin_prev = wmin;
    wmax_prev = wmax;
  }

  void set_window(const double *a_in, const double *b_in, bool new_vec) {  // SetUpdateRange (:792-810)
    double a[3] = {a_in[0], a_in[1], a_in[2]}, b[3] = {b_in[0], b_in[1], b_in[2]};
    if (mode == 0)
      for (int i = 0; i < 3; ++i) {
        a[i] = std::max(a[i], lo[i]);
        b[i] = std::min(b[i], hi[i]);
      }
    if (new_vec) {
      wmin_prev = wmin;
      wmax_prev = wmax;
    }
    wmin = pos2vox(a);
    double bb[3] = {b[0] - res / 2, b[1] - res / 2, b[2] - res / 2};
    wmax = pos2vox(bb);
  }

  int observe_vox(I3 v, int occ) {  // SetOccupancy(Vector3i,int), PROBABILISTIC branch (:417-437)
    const int s = slot(v);
    if (!in_window(v)) return s;
    seen[s]++;
    hits[s] += occ;
    if (seen[s] == 1) q_occ.push_back({v, 0.0});
    return s;
  }
  int observe_pos(const double *p, int occ) {  // SetOccupancy(Vector3d,int) (:401-415)
    if (occ != 1 && occ != 0) return kUndef;
    if (!pos_in_map(p)) return kUndef;
    return observe_vox(pos2vox(p), occ);
  }

  bool fuse(bool global_map) {  // UpdateOccupancy (src/ESDFMap.cpp:235-271)
    while (!q_occ.empty()) {
      const Item e = q_occ.front();
      q_occ.pop_front();
      const int s = slot(e.p);
      const bool was = occupied(s);
      const double step = (hits[s] >= seen[s] - hits[s]) ? l_hit : l_miss;  // majority vote
      hits[s] = seen[s] = 0;
      if (dist[s] < 0) {  // first observation ever: unobserved -> observed/no obstacle
        dist[s] = kInf;
        link_front(undef_slot, s);
      }
      if ((step >= 0 && logodds[s] >= l_max) || (step <= 0 && logodds[s] <= l_min)) continue;
      if (!global_map && !in_window(e.p, false)) {  // local-window reset quirk (:256-259)
        logodds[s] = 0;
        dist[s] = kInf;
      }
      logodds[s] = std::min(std::max(logodds[s] + step, l_min), l_max);
      const bool now = occupied(s);
      if (now && !was)
        q_ins.push_back({e.p, 0.0});
      else if (!now && was)
        q_del.push_back({e.p, (double)kInf});
    }
    return !q_ins.empty() || !q_del.empty();
  }

  void relax(oracle_esdf_stats *st) {  // UpdateESDF (src/ESDFMap.cpp:273-398)
    // --- phase 1: newly occupied voxels become their own closest obstacle (:278-291)
    while (!q_ins.empty()) {
      const Item e = q_ins.front();
      q_ins.pop_front();
      const int s = slot(e.p);
      if (!occupied(s)) continue;
      unlink(slot(coc[s]), s);
      coc[s] = e.p;
      dist[s] = 0.0;
      link_front(s, s);
      q_upd.push_back(e);
    }
    // --- phase 2: every voxel that pointed at a vanished obstacle is re-seeded from the FIRST
    //     neighbour (stencil order) whose own closest obstacle is still occupied (:292-337)
    while (!q_del.empty()) {
      const Item e = q_del.front();
      q_del.pop_front();
      const int s = slot(e.p);
      if (occupied(s)) continue;
      int nxt;
      for (int o = head[s]; o != kUndef; o = nxt) {
        coc[o] = kNone;
        const I3 ov = vox_of(o);
        double d = kInf;
        for (const I3 &dir : kDirs) {
          const I3 nv = ov + dir;
          const int ns = slot(nv);
          if (in_window(nv) && defined(coc[ns]) && occupied(slot(coc[ns]))) {
            const double t = metric(ov, coc[ns]);
            if (t < d) {
              d = t;
              coc[o] = coc[ns];
            }
            break;
          }
        }
        prev[o] = kUndef;
        nxt = next[o];
        next[o] = kUndef;
        dist[o] = d;
        if (d < kInf) q_upd.push_back({ov, d});
        link_front(slot(coc[o]), o);
      }
      head[s] = kUndef;
    }
    // --- phase 3: FIFO relaxation, pull then push over the 24-stencil (:339-392)
    // MEASUREMENT (fifo_probe, off by default; tools/dev/fifo_depth.py): the queue is processed in LAYERS (layer L + 1 = what
    // layer L enqueued).  How far is the order INSIDE a layer from being irrelevant?  Every processed entry gets a depth:
    // 1 + the largest depth of an EARLIER entry of the SAME layer that wrote a word this entry reads (its own voxel's state,
    // the 24 neighbours' obstacle in the pull, their distance in the push).  A layer of depth 1 can be processed in any order
    // -- or all at once; depth k needs k ordered sub-steps.  The probe only watches; the order and the result are untouched.
    int64_t expanded = 0, changes = 0;
    size_t layer_left = q_upd.size();
    int layer_id = 1, layer_depth = 0;
    int64_t layer_entries = 0, layer_dependent = 0;
    auto close_layer = [&]() {
      if (fifo_probe && layer_entries) fifo_layers.push_back({layer_entries, layer_dependent, (int64_t)layer_depth});
      ++layer_id, layer_depth = 0, layer_entries = 0, layer_dependent = 0;
    };
    if (fifo_probe) {
      fifo_wlayer.assign(dist.size(), 0);
      fifo_wdepth.assign(dist.size(), 0);
    }
    while (!q_upd.empty()) {
      if (layer_left == 0) {
        close_layer();
        layer_left = q_upd.size();
      }
      --layer_left;
      const Item e = q_upd.front();
      q_upd.pop_front();
      const int s = slot(e.p);
      int depth = 1;
      auto reads = [&](int x) {
        if (fifo_probe && fifo_wlayer[x] == layer_id && fifo_wdepth[x] + 1 > depth) depth = fifo_wdepth[x] + 1;
      };
      auto writes = [&](int x) {
        if (!fifo_probe) return;
        if (fifo_wlayer[x] != layer_id) fifo_wlayer[x] = layer_id, fifo_wdepth[x] = 0;
        if (depth > fifo_wdepth[x]) fifo_wdepth[x] = depth;
      };
      reads(s);
      if (e.d != dist[s]) {  // stale entry (made stale by an earlier entry of this layer, if reads(s) raised the depth)
        if (fifo_probe && depth > 1) ++layer_entries, ++layer_dependent, layer_depth = std::max(layer_depth, depth);
        continue;
      }
      ++expanded;
      bool improved = false;
      for (int i = 0; i < 24; ++i) {  // pull
        const I3 nv = e.p + kDirs[i];
        if (!in_window(nv)) continue;
        const int ns = slot(nv);
        reads(ns);
        if (!defined(coc[ns])) continue;
        const double t = metric(e.p, coc[ns]);
        if (dist[s] > t) {
          dist[s] = t;
          improved = true;
          unlink(slot(coc[s]), s);
          link_front(slot(coc[ns]), s);
          coc[s] = coc[ns];
        }
      }
      if (fifo_probe) ++layer_entries, layer_dependent += depth > 1, layer_depth = std::max(layer_depth, depth);
      if (improved) {
        ++changes;
        writes(s);
        q_upd.push_back({e.p, dist[s]});
        continue;
      }
      const int owner = slot(coc[s]);
      for (const I3 &dir : kDirs) {  // push; unobserved voxels hold -10000 and never satisfy '>'
        const I3 nv = e.p + dir;
        if (!in_window(nv)) continue;
        const int ns = slot(nv);
        const double t = metric(nv, coc[s]);
        if (dist[ns] > t) {
          dist[ns] = t;
          unlink(slot(coc[ns]), ns);
          link_front(owner, ns);
          coc[ns] = coc[s];
          writes(ns);
          q_upd.push_back({nv, t});
        }
      }
    }
    close_layer();
    if (st) {
      st->expanded = expanded;
      st->change_num = changes;
    }
  }

  // ---- NOT the reference: a CPU model of the GPU's level engine (fiesta_amd/csrc/level_kernels.hpp) --------------------
  // The reference's update queue is a FIFO, so its entries are processed in LAYERS: layer L + 1 is what the processing
  // of layer L enqueued.  A parallel engine cannot reproduce the order inside a layer, but it can keep the layers: every
  // entry of a layer first PULLS (from the field as the layer found it), the voxels that improved go to the next layer,
  // the others PUSH (a minimum per target voxel; a target that improves goes to the next layer).  This model runs exactly
  // that on the port's arrays, so that the schedule can be judged against the reference's own order spread without a GPU
  // (tests/test_levelsync_model.py).  The per-obstacle lists are not maintained (the engine has none: the orphans of a
  // delete are found by a scan); a map driven through this schedule must never go back to relax().
  // Orphans of a delete are reset and enter layer 0 as pull-only entries (their first pull is their re-seed, :308-321).
  // Orphans OUTSIDE the update window: see level_kernels.hpp (k_level_outside) -- the same rule, the same order of tests.
  int schedule = 0;  // 0: relax() (the reference's FIFO); 1: relax_levels()
  int64_t levels_run = 0;
  int nslots() const { return mode == 1 ? count : total; }
  void relax_levels(oracle_esdf_stats *st) {
    std::vector<int> F, Fn;
    std::vector<char> inF(reserve + 1, 0);
    auto add = [&](std::vector<int> &L, int s) {
      if ((size_t)s >= inF.size()) inF.resize(reserve + 1, 0);
      if (!inF[s]) inF[s] = 1, L.push_back(s);
    };
    while (!q_ins.empty()) {  // :278-291
      const Item e = q_ins.front();
      q_ins.pop_front();
      const int s = slot(e.p);
      if (!occupied(s)) continue;
      coc[s] = e.p;
      dist[s] = 0.0;
      add(F, s);
    }
    bool any_del = false;
    while (!q_del.empty()) {
      const Item e = q_del.front();
      q_del.pop_front();
      if (!occupied(slot(e.p))) any_del = true;
    }
    if (any_del) {  // :292-337 without the lists: every voxel whose (possibly stale) link names a vanished obstacle
      std::vector<int> orphans;
      const int n0 = nslots();
      for (int s = (mode == 1 ? 1 : 0); s < n0; ++s)
        if (defined(coc[s]) && dist[s] >= 0 && !occupied(slot(coc[s]))) orphans.push_back(s);
      // Orphans OUTSIDE the window (:308-321 gates the neighbour, not the orphan): the reference re-seeds one from its first
      // in-window neighbour that is valid AT THAT MOMENT of the list walk -- a live obstacle, or an orphan walked earlier.
      // The list is push-front in adoption order, i.e. walked from the rim of the dead cell inwards: "walked earlier" is
      // modelled as "orphan of another vanished obstacle, or of the same one and not closer to it".  One that finds nothing
      // stays at infinity for good (never queued :329, never pushed into :378).
      std::vector<char> may_wait(reserve + 1, 0);
      for (int o : orphans) {
        const I3 ov = vox_of(o);
        if (in_window(ov)) continue;
        const I3 X = coc[o];
        const double dox = metric(ov, X);
        for (const I3 &dir : kDirs) {
          const I3 nv = ov + dir;
          if (!in_window(nv)) continue;
          const int ns = slot(nv);
          if (!defined(coc[ns]) || dist[ns] < 0) continue;
          const bool dead = !occupied(slot(coc[ns]));
          if (!dead || !(coc[ns].x == X.x && coc[ns].y == X.y && coc[ns].z == X.z) || metric(nv, X) >= dox) {
            may_wait[o] = 1;
            break;
          }
        }
      }
      std::vector<I3> was(schedule >= 4 ? orphans.size() : 0);
      if (schedule >= 4)
        for (size_t k = 0; k < orphans.size(); ++k) was[k] = coc[orphans[k]];
      const std::vector<I3> &was5 = was;
      for (int o : orphans) coc[o] = kNone, dist[o] = kInf;
      if (schedule == 4) {
        // EXPERIMENT (schedule 4): the list walk itself, without the lists.  A vanished obstacle's list is push-front in
        // adoption order, i.e. (roughly) walked from the voxels farthest from it to the nearest; every orphan takes the
        // first valid neighbour AT THAT MOMENT -- orphans walked earlier included (:300-321).  Here: per vanished obstacle,
        // its orphans by decreasing distance from it, one after the other.
        std::vector<size_t> order(orphans.size());
        for (size_t k = 0; k < order.size(); ++k) order[k] = k;
        auto key = [&](size_t k) { return ((long long)was[k].x * 2048 + was[k].y) * 2048 + was[k].z; };
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
          if (key(a) != key(b)) return key(a) < key(b);
          return metric(vox_of(orphans[a]), was[a]) > metric(vox_of(orphans[b]), was[b]);
        });
        for (size_t k : order) {
          const int o = orphans[k];
          const I3 ov = vox_of(o);
          if (!in_window(ov)) continue;
          for (const I3 &dir : kDirs) {
            const I3 nv = ov + dir;
            if (!in_window(nv)) continue;
            const int ns = slot(nv);
            if (!defined(coc[ns]) || dist[ns] < 0 || !occupied(slot(coc[ns]))) continue;
            coc[o] = coc[ns], dist[o] = metric(ov, coc[ns]);
            add(F, o);
            break;
          }
        }
        for (int o : orphans)
          if (!in_window(vox_of(o)) && may_wait[o]) add(F, o);
      } else if (schedule == 5 || schedule == 6) {
        // EXPERIMENT (schedules 5, 6): the same, shaped for a parallel machine -- all vanished obstacles at once, their orphans
        // in SHELLS of decreasing distance from their own obstacle (5: one shell per squared distance; 6: per whole voxel
        // of distance), every shell one Jacobi step: an orphan takes its first neighbour that was valid when the shell began.
        std::vector<std::pair<long long, int>> byd;
        for (size_t k = 0; k < orphans.size(); ++k) {
          const int o = orphans[k];
          if (!in_window(vox_of(o))) continue;
          const double m = metric(vox_of(o), was5[k]);
          const long long d2 = (long long)std::llround(m * m * res_inv * res_inv);
          const long long shell = schedule == 5 ? d2 : std::min<long long>(63, (long long)std::floor(std::sqrt((double)d2)));  // (6: the GPU's 64 bins)
          byd.push_back({-shell, o});
        }
        std::stable_sort(byd.begin(), byd.end());
        for (size_t b = 0; b < byd.size();) {
          size_t e = b;
          while (e < byd.size() && byd[e].first == byd[b].first) ++e;
          std::vector<std::pair<int, I3>> got;
          for (size_t k = b; k < e; ++k) {
            const int o = byd[k].second;
            const I3 ov = vox_of(o);
            for (const I3 &dir : kDirs) {
              const I3 nv = ov + dir;
              if (!in_window(nv)) continue;
              const int ns = slot(nv);
              if (!defined(coc[ns]) || dist[ns] < 0 || !occupied(slot(coc[ns]))) continue;
              got.push_back({o, coc[ns]});
              break;
            }
          }
          for (auto &g : got) coc[g.first] = g.second, dist[g.first] = metric(vox_of(g.first), g.second), add(F, g.first);
          b = e;
        }
        for (int o : orphans)
          if (!in_window(vox_of(o)) && may_wait[o]) add(F, o);
      } else if (schedule == 2) {
        // EXPERIMENT (schedule 2): the reference's list walk re-seeds an orphan from its first valid neighbour AT THAT MOMENT,
        // and orphans walked earlier are valid -- the dead cell fills from its rim inwards during the delete drain, and every
        // filled orphan is queued in layer 0 (:300-331).  Modelled as rounds: in round k every still-empty orphan with a valid
        // neighbour (as round k-1 left the field) adopts the FIRST such neighbour's obstacle, in the reference's direction order.
        std::vector<int> empty;
        for (int o : orphans)
          if (in_window(vox_of(o))) empty.push_back(o);
        for (;;) {
          std::vector<std::pair<int, I3>> got;
          std::vector<int> still;
          for (int o : empty) {
            const I3 ov = vox_of(o);
            bool found = false;
            for (const I3 &dir : kDirs) {
              const I3 nv = ov + dir;
              if (!in_window(nv)) continue;
              const int ns = slot(nv);
              if (!defined(coc[ns]) || dist[ns] < 0 || !occupied(slot(coc[ns]))) continue;
              got.push_back({o, coc[ns]});
              found = true;
              break;
            }
            if (!found) still.push_back(o);
          }
          if (got.empty()) break;
          for (auto &g : got) coc[g.first] = g.second, dist[g.first] = metric(vox_of(g.first), g.second), add(F, g.first);
          empty.swap(still);
        }
        for (int o : orphans)
          if (!in_window(vox_of(o)) && may_wait[o]) add(F, o);
      } else {
        for (int o : orphans)
          if (in_window(vox_of(o)) || may_wait[o]) add(F, o);
      }
    }
    int64_t expanded = 0, changes = 0;
    std::vector<std::pair<int, I3>> better, pushers;
    std::vector<int> waiting;
    while (!F.empty()) {
      ++levels_run;
      better.clear(), pushers.clear();
      for (int s : F) {  // pull, from the field as this layer found it (:349-367)
        const I3 v = vox_of(s);
        double best = dist[s];
        I3 bc = kNone;
        for (int i = 0; i < 24; ++i) {
          const I3 nv = v + kDirs[i];
          if (!in_window(nv)) continue;
          const int ns = slot(nv);
          if (!defined(coc[ns])) continue;
          const double t = metric(v, coc[ns]);
          if (best > t) best = t, bc = coc[ns];
        }
        ++expanded;
        if (defined(bc))
          better.push_back({s, bc});
        else if (defined(coc[s]))
          pushers.push_back({s, coc[s]});
        else if (!in_window(v))
          waiting.push_back(s);  // an orphan outside the window: nobody will push into it, it has to ask again
      }
      for (int s : F) inF[s] = 0;
      Fn.clear();
      for (auto &b : better) {  // :369-373
        coc[b.first] = b.second;
        dist[b.first] = metric(vox_of(b.first), b.second);
        add(Fn, b.first);
        ++changes;
      }
      for (auto &p : pushers) {  // push (:375-391): a minimum per target
        const I3 v = vox_of(p.first);
        for (const I3 &dir : kDirs) {
          const I3 nv = v + dir;
          if (!in_window(nv)) continue;
          const int ns = slot(nv);
          const double t = metric(nv, p.second);
          if (dist[ns] > t) {
            dist[ns] = t;
            coc[ns] = p.second;
            add(Fn, ns);
          }
        }
      }
      if (!Fn.empty())
        for (int s : waiting) add(Fn, s);
      waiting.clear();
      F.swap(Fn);
    }
    if (st) st->expanded = expanded, st->change_num = changes;
  }

  // EXPERIMENT (schedule 3): two lineages.  The reference's queue holds the inserted voxels AHEAD of the re-seeded orphans
  // (:278-337), and a FIFO keeps that order through every layer: in each layer the descendants of the inserts are processed
  // before the descendants of the orphans, and an entry whose voxel was improved since it was queued is skipped (:345).
  // Here: every level has an I part and an O part, each a Jacobi step (pull from the field as the part found it, then push),
  // the O part after the I part, entries carrying the distance they were queued with.
  void relax_lineages(oracle_esdf_stats *st) {
    struct Ent { int s; double d; };
    std::vector<Ent> FI, FO, NI, NO;
    while (!q_ins.empty()) {
      const Item e = q_ins.front();
      q_ins.pop_front();
      const int s = slot(e.p);
      if (!occupied(s)) continue;
      coc[s] = e.p;
      dist[s] = 0.0;
      FI.push_back({s, 0.0});
    }
    bool any_del = false;
    while (!q_del.empty()) {
      const Item e = q_del.front();
      q_del.pop_front();
      if (!occupied(slot(e.p))) any_del = true;
    }
    if (any_del) {
      std::vector<int> orphans;
      const int n0 = nslots();
      for (int s = (mode == 1 ? 1 : 0); s < n0; ++s)
        if (defined(coc[s]) && dist[s] >= 0 && !occupied(slot(coc[s]))) orphans.push_back(s);
      for (int o : orphans) coc[o] = kNone, dist[o] = kInf;
      std::vector<int> empty;
      for (int o : orphans)
        if (in_window(vox_of(o))) empty.push_back(o);
      for (;;) {  // the rim-inward fill of schedule 2
        std::vector<std::pair<int, I3>> got;
        std::vector<int> still;
        for (int o : empty) {
          const I3 ov = vox_of(o);
          bool found = false;
          for (const I3 &dir : kDirs) {
            const I3 nv = ov + dir;
            if (!in_window(nv)) continue;
            const int ns = slot(nv);
            if (!defined(coc[ns]) || dist[ns] < 0 || !occupied(slot(coc[ns]))) continue;
            got.push_back({o, coc[ns]});
            found = true;
            break;
          }
          if (!found) still.push_back(o);
        }
        if (got.empty()) break;
        for (auto &g : got) coc[g.first] = g.second, dist[g.first] = metric(vox_of(g.first), g.second), FO.push_back({g.first, dist[g.first]});
        empty.swap(still);
      }
    }
    int64_t expanded = 0, changes = 0;
    auto part = [&](std::vector<Ent> &F, std::vector<Ent> &N) {
      std::vector<std::pair<int, I3>> better, pushers;
      for (const Ent &e : F) {
        const int s = e.s;
        if (e.d != dist[s]) continue;  // stale (:345)
        const I3 v = vox_of(s);
        double best = dist[s];
        I3 bc = kNone;
        for (int i = 0; i < 24; ++i) {
          const I3 nv = v + kDirs[i];
          if (!in_window(nv)) continue;
          const int ns = slot(nv);
          if (!defined(coc[ns])) continue;
          const double t = metric(v, coc[ns]);
          if (best > t) best = t, bc = coc[ns];
        }
        ++expanded;
        if (defined(bc))
          better.push_back({s, bc});
        else if (defined(coc[s]))
          pushers.push_back({s, coc[s]});
      }
      for (auto &b : better) {
        const double t = metric(vox_of(b.first), b.second);
        if (dist[b.first] > t) {
          coc[b.first] = b.second, dist[b.first] = t;
          N.push_back({b.first, t});
          ++changes;
        }
      }
      for (auto &p : pushers) {
        const I3 v = vox_of(p.first);
        for (const I3 &dir : kDirs) {
          const I3 nv = v + dir;
          if (!in_window(nv)) continue;
          const int ns = slot(nv);
          const double t = metric(nv, p.second);
          if (dist[ns] > t) {
            dist[ns] = t;
            coc[ns] = p.second;
            N.push_back({ns, t});
          }
        }
      }
    };
    while (!FI.empty() || !FO.empty()) {
      ++levels_run;
      NI.clear(), NO.clear();
      part(FI, NI);
      part(FO, NO);
      FI.swap(NI), FO.swap(NO);
    }
    if (st) st->expanded = expanded, st->change_num = changes;
  }

  double distance_vox(I3 v) {  // GetDistance(Vector3i) (src/ESDFMap.cpp:477-479): no bounds check
    const int s = slot(v);
    return dist[s] < 0 ? (double)kInf : dist[s];
  }
  double distance_pos(const double *p) {  // GetDistance(Vector3d) (:467-475)
    if (!pos_in_map(p)) return kUndef;
    return distance_vox(pos2vox(p));
  }
  double trilinear(const double *p, double *g) {  // GetDistWithGradTrilinear (:481-540)
    if (!pos_in_map(p)) return -1;
    double pm[3] = {p[0] - 0.5 * res * 1.0, p[1] - 0.5 * res * 1.0, p[2] - 0.5 * res * 1.0};
    const I3 b = pos2vox(pm);
    double c[3];
    vox2pos(b, c);
    const double fx = (p[0] - c[0]) * res_inv, fy = (p[1] - c[1]) * res_inv, fz = (p[2] - c[2]) * res_inv;
    double v[2][2][2];
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j)
        for (int k = 0; k < 2; ++k) v[i][j][k] = distance_vox({b.x + i, b.y + j, b.z + k});
    const double v00 = (1 - fx) * v[0][0][0] + fx * v[1][0][0];
    const double v01 = (1 - fx) * v[0][0][1] + fx * v[1][0][1];
    const double v10 = (1 - fx) * v[0][1][0] + fx * v[1][1][0];
    const double v11 = (1 - fx) * v[0][1][1] + fx * v[1][1][1];
    const double v0 = (1 - fy) * v00 + fy * v10;
    const double v1 = (1 - fy) * v01 + fy * v11;
    const double d = (1 - fz) * v0 + fz * v1;
    g[2] = (v1 - v0) * res_inv;
    g[1] = ((1 - fz) * (v10 - v00) + fz * (v11 - v01)) * res_inv;
    double gx = (1 - fz) * (1 - fy) * (v[1][0][0] - v[0][0][0]);
    gx += (1 - fz) * fy * (v[1][1][0] - v[0][1][0]);
    gx += fz * (1 - fy) * (v[1][0][1] - v[0][0][1]);
    gx += fz * fy * (v[1][1][1] - v[0][1][1]);
    g[0] = gx * res_inv;
    return d;
  }

  bool lists_consistent() {  // CheckConsistency (src/ESDFMap.cpp:856-902)
    auto bad = [&](int s) {
      if ((prev[s] != kUndef && next[prev[s]] != s) || (next[s] != kUndef && prev[next[s]] != s)) return true;
      if (prev[s] == kUndef && dist[s] >= 0 && head[slot(coc[s])] != s) return true;
      return false;
    };
    if (mode == 1) {
      for (int i = 1; i < count; ++i)
        if (bad(slot(slot_vox[i]))) return false;
    } else {
      for (int s = 0; s < total; ++s)
        if (bad(s)) return false;
    }
    return true;
  }
};

// ---- Amanatides-Woo traversal with the reference's exact arithmetic (src/raycast.cpp:6-23,56-158) ----
inline int sgn(int v) { return v == 0 ? 0 : (v < 0 ? -1 : 1); }
inline double wrap1(double v) { return std::fmod(std::fmod(v, 1.0) + 1.0, 1.0); }
double first_crossing(double s, double ds) {  // intbound
  if (ds < 0) return first_crossing(-s, -ds);
  return (1 - wrap1(s)) / ds;
}

// Returns false if the reference would throw (more than 1500 voxels emitted).
bool walk_ray(const double *a, const double *b, const double *lo, const double *hi, std::vector<I3> *out) {
  int c[3] = {(int)std::floor(a[0]), (int)std::floor(a[1]), (int)std::floor(a[2])};
  const int e[3] = {(int)std::floor(b[0]), (int)std::floor(b[1]), (int)std::floor(b[2])};
  const double r0 = b[0] - a[0], r1 = b[1] - a[1], r2 = b[2] - a[2];
  const double reach2 = r0 * r0 + r1 * r1 + r2 * r2;
  // NB: the stepping direction is the INTEGER voxel delta, not the true ray direction (:89-107)
  double delta[3], tmax[3], tstep[3];
  int step[3];
  for (int i = 0; i < 3; ++i) {
    delta[i] = e[i] - c[i];
    step[i] = sgn((int)delta[i]);
    tmax[i] = first_crossing(a[i], delta[i]);
    tstep[i] = ((double)step[i]) / delta[i];
  }
  out->clear();
  if (step[0] == 0 && step[1] == 0 && step[2] == 0) return true;
  for (;;) {
    if (c[0] >= lo[0] && c[0] < hi[0] && c[1] >= lo[1] && c[1] < hi[1] && c[2] >= lo[2] && c[2] < hi[2]) {
      out->push_back({c[0], c[1], c[2]});
      const double q0 = c[0] - a[0], q1 = c[1] - a[1], q2 = c[2] - a[2];
      if (q0 * q0 + q1 * q1 + q2 * q2 > reach2) return true;
      if (out->size() > 1500) return false;
    }
    if (c[0] == e[0] && c[1] == e[1] && c[2] == e[2]) break;
    int ax;  // strict '<' tie rules of :139-157
    if (tmax[0] < tmax[1])
      ax = (tmax[0] < tmax[2]) ? 0 : 2;
    else
      ax = (tmax[1] < tmax[2]) ? 1 : 2;
    c[ax] += step[ax];
    tmax[ax] += tstep[ax];
  }
  return true;
}

}  // namespace

struct oracle_map {
  Port p;
};

extern "C" {

const char *oracle_kind(void) { return "port"; }

oracle_map *oracle_create(int mode, const double origin[3], double resolution, const double map_size[3],
                          int reserve_size) {
  oracle_map *m = new oracle_map;
  Port &p = m->p;
  p.mode = mode;
  p.res = resolution;
  p.res_inv = 1 / resolution;
  for (int i = 0; i < 3; ++i) p.org[i] = origin[i];
  if (mode == 0) {  // src/ESDFMap.cpp:171-213
    for (int i = 0; i < 3; ++i) {
      p.gs[i] = (int)std::ceil(map_size[i] / resolution);
      p.lo[i] = origin[i];
      p.hi[i] = origin[i] + map_size[i];
    }
    p.gs_yz = p.gs[1] * p.gs[2];
    p.total = p.gs[0] * p.gs_yz;
    p.undef_slot = p.total;
    p.grow(p.total);
    p.head.resize(p.total + 1, kUndef);
    p.stamp_free.assign(p.total, 0);
    p.stamp_occ.assign(p.total, 0);
  } else {  // src/ESDFMap.cpp:130-167
    if (reserve_size < 512) reserve_size = 512;
    p.undef_slot = 0;
    p.count = 1;
    p.grow(reserve_size + 1);
  }
  p.full_window();
  return m;
}
void oracle_destroy(oracle_map *m) { delete m; }

void oracle_set_parameters(oracle_map *m, double p_hit, double p_miss, double p_min, double p_max,
                           double p_occ) {  // SetParameters + Logit (src/ESDFMap.cpp:12-14,218-224)
  auto logit = [](double x) { return std::log(x / (1 - x)); };
  m->p.l_hit = logit(p_hit);
  m->p.l_miss = logit(p_miss);
  m->p.l_min = logit(p_min);
  m->p.l_max = logit(p_max);
  m->p.l_occ = logit(p_occ);
}
int64_t oracle_grid_total_size(oracle_map *m) { return m->p.mode == 0 ? m->p.total : m->p.count; }
void oracle_grid_size(oracle_map *m, int32_t out[3]) {
  for (int i = 0; i < 3; ++i) out[i] = m->p.gs[i];
}
void oracle_set_original_range(oracle_map *m) { m->p.full_window(); }
void oracle_set_update_range(oracle_map *m, const double a[3], const double b[3], int new_vec) {
  m->p.set_window(a, b, new_vec != 0);
}
void oracle_set_occupancy_vox(oracle_map *m, const int32_t *vox, const int32_t *occ, int64_t n, int32_t *ret) {
  for (int64_t i = 0; i < n; ++i) {
    int r = m->p.observe_vox({vox[3 * i], vox[3 * i + 1], vox[3 * i + 2]}, occ[i]);
    if (ret) ret[i] = r;
  }
}
void oracle_set_occupancy_pos(oracle_map *m, const double *pos, const int32_t *occ, int64_t n, int32_t *ret) {
  for (int64_t i = 0; i < n; ++i) {
    int r = m->p.observe_pos(pos + 3 * i, occ[i]);
    if (ret) ret[i] = r;
  }
}
int oracle_check_update(oracle_map *m) { return !m->p.q_occ.empty(); }  // CheckUpdate (:227-233)
int oracle_update_occupancy(oracle_map *m, int global_map, int64_t *n_insert, int64_t *n_delete) {
  bool r = m->p.fuse(global_map != 0);
  if (n_insert) *n_insert = (int64_t)m->p.q_ins.size();
  if (n_delete) *n_delete = (int64_t)m->p.q_del.size();
  return r;
}
void oracle_update_esdf(oracle_map *m, oracle_esdf_stats *st) {
  oracle_esdf_stats tmp;
  if (!st) st = &tmp;
  st->inserted = (int64_t)m->p.q_ins.size();
  st->deleted = (int64_t)m->p.q_del.size();
  auto t0 = std::chrono::steady_clock::now();
  if (m->p.schedule == 3)
    m->p.relax_lineages(st);
  else if (m->p.schedule)
    m->p.relax_levels(st);
  else
    m->p.relax(st);
  st->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
void oracle_get_distance_vox(oracle_map *m, const int32_t *vox, int64_t n, double *out) {
  for (int64_t i = 0; i < n; ++i) out[i] = m->p.distance_vox({vox[3 * i], vox[3 * i + 1], vox[3 * i + 2]});
}
void oracle_get_distance_pos(oracle_map *m, const double *pos, int64_t n, double *out) {
  for (int64_t i = 0; i < n; ++i) out[i] = m->p.distance_pos(pos + 3 * i);
}
void oracle_get_dist_grad(oracle_map *m, const double *pos, int64_t n, double *dist, double *grad) {
  for (int64_t i = 0; i < n; ++i) {
    grad[3 * i] = grad[3 * i + 1] = grad[3 * i + 2] = 0;
    dist[i] = m->p.trilinear(pos + 3 * i, grad + 3 * i);
  }
}
void oracle_get_occupancy_vox(oracle_map *m, const int32_t *vox, int64_t n, int32_t *out) {  // :462-465
  for (int64_t i = 0; i < n; ++i) out[i] = m->p.occupied(m->p.slot({vox[3 * i], vox[3 * i + 1], vox[3 * i + 2]}));
}
void oracle_get_occupancy_pos(oracle_map *m, const double *pos, int64_t n, int32_t *out) {  // :452-460
  for (int64_t i = 0; i < n; ++i)
    out[i] = m->p.pos_in_map(pos + 3 * i) ? (int)m->p.occupied(m->p.slot(m->p.pos2vox(pos + 3 * i))) : kUndef;
}
void oracle_dump_dense(oracle_map *m, double *dist, int32_t *coc, uint8_t *occ, double *logodds) {
  Port &p = m->p;
  if (p.mode != 0) return;
  for (int64_t i = 0; i < p.total; ++i) {
    if (dist) dist[i] = p.dist[i];
    if (coc) {
      coc[3 * i] = p.coc[i].x;
      coc[3 * i + 1] = p.coc[i].y;
      coc[3 * i + 2] = p.coc[i].z;
    }
    if (occ) occ[i] = p.occupied((int)i);
    if (logodds) logodds[i] = p.logodds[i];
  }
}
void oracle_dump_counts(oracle_map *m, int32_t *num_hit, int32_t *num_miss) {
  Port &p = m->p;
  if (p.mode != 0) {  // hash mode: the order of oracle_dump_hash
    for (int64_t k = 0; k < p.count - 1; ++k) {
      if (num_hit) num_hit[k] = p.hits[k + 1];
      if (num_miss) num_miss[k] = p.seen[k + 1];
    }
    return;
  }
  for (int64_t i = 0; i < p.total; ++i) {
    if (num_hit) num_hit[i] = p.hits[i];
    if (num_miss) num_miss[i] = p.seen[i];
  }
}
int64_t oracle_dump_hash(oracle_map *m, int32_t *vox, double *dist, int32_t *coc, uint8_t *occ) {
  Port &p = m->p;
  if (p.mode != 1) return 0;
  const int64_t n = p.count - 1;
  for (int64_t k = 0; k < n; ++k) {
    const int i = (int)k + 1;
    if (vox) {
      vox[3 * k] = p.slot_vox[i].x;
      vox[3 * k + 1] = p.slot_vox[i].y;
      vox[3 * k + 2] = p.slot_vox[i].z;
    }
    if (dist) dist[k] = p.dist[i];
    if (coc) {
      coc[3 * k] = p.coc[i].x;
      coc[3 * k + 1] = p.coc[i].y;
      coc[3 * k + 2] = p.coc[i].z;
    }
    if (occ) occ[k] = p.occupied(i);
  }
  return n;
}
int oracle_check_consistency(oracle_map *m) { return m->p.lists_consistent(); }
// port only: the schedule UpdateESDF runs (0 = the reference's FIFO; 1 = the model of the GPU's level engine, relax_levels;
// 2 .. 6 = experiments on that model kept for tools/dev/schedule_experiment.py: 2 the dead cell filled from its rim inwards
// ahead of level 0, 3 also the inserts' descendants ahead of the orphans' in every level, 4 the reference's list walk emulated
// (per vanished obstacle, orphans by decreasing distance from it, one after the other), 5 / 6 the same in parallel shells of
// equal squared / whole-voxel distance.  DESIGN.md section 3c says what they showed.)
void oracle_set_schedule(oracle_map *m, int schedule) { m->p.schedule = schedule; }
// port only: the probe of relax() -- per layer of the update queue (entries processed, entries that read a word an earlier
// entry of the same layer wrote, longest such chain).  enable: 1 on / 0 off (clears).  oracle_fifo_layers copies up to cap
// rows of 3 int64 and returns the number of layers recorded since the probe was switched on.
void oracle_fifo_probe(oracle_map *m, int enable) {
  m->p.fifo_probe = enable != 0;
  m->p.fifo_layers.clear();
}
int64_t oracle_fifo_layers(oracle_map *m, int64_t *out, int64_t cap) {
  const int64_t n = (int64_t)m->p.fifo_layers.size();
  for (int64_t i = 0; i < n && i < cap; ++i)
    out[3 * i] = m->p.fifo_layers[i].entries, out[3 * i + 1] = m->p.fifo_layers[i].dependent, out[3 * i + 2] = m->p.fifo_layers[i].depth;
  return n;
}
int64_t oracle_levels_run(oracle_map *m) { return m->p.levels_run; }

// GetPointCloud (src/ESDFMap.cpp:544-582), restated: occupied voxels inside the update range (hash flavour: x and y
// only) whose z index lies within the visualisation bounds, as voxel centres narrowed to float (Point32).
int64_t oracle_get_point_cloud(oracle_map *m, int vis_lower_bound, int vis_upper_bound, float *xyz, int64_t cap) {
  Port &p = m->p;
  int64_t n = 0;
  auto emit = [&](I3 v) {
    double c[3];
    p.vox2pos(v, c);
    if (n < cap) xyz[3 * n] = (float)c[0], xyz[3 * n + 1] = (float)c[1], xyz[3 * n + 2] = (float)c[2];
    ++n;
  };
  if (p.mode == 1) {
    for (int s = 1; s < p.count; ++s) {
      const I3 v = p.slot_vox[s];
      if (!p.occupied(s) || v.z < vis_lower_bound || v.z > vis_upper_bound || v.x < p.wmin.x || v.x > p.wmax.x ||
          v.y < p.wmin.y || v.y > p.wmax.y)
        continue;
      emit(v);
    }
  } else {
    for (int x = p.wmin.x; x <= p.wmax.x; ++x)
      for (int y = p.wmin.y; y <= p.wmax.y; ++y)
        for (int z = p.wmin.z; z <= p.wmax.z; ++z) {
          if (!p.occupied(p.slot({x, y, z})) || z < vis_lower_bound || z > vis_upper_bound) continue;
          emit({x, y, z});
        }
  }
  return n;
}
// The rainbow of GetSliceMarker (src/ESDFMap.cpp:584-636): hue h in [0,1) around the colour circle at full
// saturation and value.
static void rainbow(double h, float *rgba) {
  h -= std::floor(h);
  h *= 6;
  const int i = (int)std::floor(h);
  double f = h - i;
  if (!(i & 1)) f = 1 - f;
  const double hi = 1.0, mid = 1.0 - f, lo = 0.0;
  double r, g, b;
  switch (i) {
    case 1: r = mid, g = hi, b = lo; break;
    case 2: r = lo, g = hi, b = mid; break;
    case 3: r = lo, g = mid, b = hi; break;
    case 4: r = mid, g = lo, b = hi; break;
    case 5: r = hi, g = lo, b = mid; break;
    default: r = hi, g = mid, b = lo; break;  // 0 and 6
  }
  rgba[0] = (float)r, rgba[1] = (float)g, rgba[2] = (float)b, rgba[3] = 1.0f;
}
// GetSliceMarker (src/ESDFMap.cpp:639-699), restated: voxels of the plane z = slice inside the x/y update range with a
// finite, defined distance; colour = rainbow(min(d / max_dist, 1)).
int64_t oracle_get_slice_marker(oracle_map *m, int slice, double max_dist, double *xyz, float *rgba, int64_t cap) {
  Port &p = m->p;
  int64_t n = 0;
  auto emit = [&](I3 v, double d) {
    if (n < cap) {
      p.vox2pos(v, xyz + 3 * n);
      rainbow(d <= max_dist ? d / max_dist : 1, rgba + 4 * n);
    }
    ++n;
  };
  if (p.mode == 1) {
    for (int s = 1; s < p.count; ++s) {
      const I3 v = p.slot_vox[s];
      if (v.z != slice || p.dist[s] < 0 || p.dist[s] >= (double)kInf || v.x < p.wmin.x || v.x > p.wmax.x || v.y < p.wmin.y ||
          v.y > p.wmax.y)
        continue;
      emit(v, p.dist[s]);
    }
  } else {
    for (int x = p.wmin.x; x <= p.wmax.x; ++x)
      for (int y = p.wmin.y; y <= p.wmax.y; ++y) {
        const int s = p.slot({x, y, slice});
        if (p.dist[s] < 0 || p.dist[s] >= (double)kInf) continue;
        emit({x, y, slice}, p.dist[s]);
      }
  }
  return n;
}

int oracle_raycast(const double start[3], const double end[3], const double minv[3], const double maxv[3],
                   double *out, int cap) {
  std::vector<I3> v;
  const bool ok = walk_ray(start, end, minv, maxv, &v);
  for (int i = 0; i < (int)v.size() && i < cap; ++i) {
    out[3 * i] = v[i].x;
    out[3 * i + 1] = v[i].y;
    out[3 * i + 2] = v[i].z;
  }
  return ok ? (int)v.size() : -1;
}

// Fiesta::RaycastProcess(0, n, tt), single thread (include/Fiesta.h:194-278) + the frame stamp bump of
// RaycastMultithread (:281-303).
static void frame_impl(oracle_map *m, oracle_map *inv, const float *pts, int64_t n, const double T[16], const double o[3],
                       const oracle_raycast_params *prm);
void oracle_raycast_frame(oracle_map *m, const float *pts, int64_t n, const double T[16], const double o[3],
                          const oracle_raycast_params *prm) {
  frame_impl(m, nullptr, pts, n, T, o, prm);
}
// ... with -DSIGNED_NEEDED (:216-218, :249-251): the inverse map sees every observation inverted (end points free,
// traversed voxels occupied) and its return value is the one the de-duplication keys on
void oracle_raycast_frame_signed(oracle_map *m, oracle_map *inv, const float *pts, int64_t n, const double T[16],
                                 const double o[3], const oracle_raycast_params *prm) {
  frame_impl(m, inv, pts, n, T, o, prm);
}
static void frame_impl(oracle_map *m, oracle_map *inv, const float *pts, int64_t n, const double T[16], const double o[3],
                       const oracle_raycast_params *prm) {
  Port &p = m->p;
  if (p.mode == 1) {
    p.hstamp_free.clear();
    p.hstamp_occ.clear();
  }
  const int tt = ++p.frame;
  const double res = p.res;
  double lo[3], hi[3], ov[3];
  for (int i = 0; i < 3; ++i) {
    lo[i] = prm->l_cornor[i] / res;
    hi[i] = prm->r_cornor[i] / res;
    ov[i] = o[i] / res;
  }
  auto norm3 = [](double a, double b, double c) { return std::sqrt(a * a + b * b + c * c); };
  auto first_visit = [&](std::vector<int> &dense, std::unordered_set<int> &sparse, int s) {
    if (p.mode == 1) return sparse.insert(s).second;
    if (s < 0 || s >= (int)dense.size()) return true;  // guard (the reference indexes unchecked)
    if (dense[s] == tt) return false;
    dense[s] = tt;
    return true;
  };
  std::vector<I3> cells;
  for (int64_t k = 0; k < n; ++k) {
    const double px = pts[3 * k], py = pts[3 * k + 1], pz = pts[3 * k + 2];
    if (std::isnan(px) || std::isnan(py) || std::isnan(pz)) continue;
    double h[4];
    for (int r = 0; r < 4; ++r) h[r] = T[4 * r] * px + T[4 * r + 1] * py + T[4 * r + 2] * pz + T[4 * r + 3] * 1.0;
    double q[3] = {h[0] / h[3], h[1] / h[3], h[2] / h[3]};
    double len = norm3(q[0] - o[0], q[1] - o[1], q[2] - o[2]);
    int s;
    if (len < prm->min_ray_length) continue;
    if (len > prm->max_ray_length) {  // clip the endpoint to max range and mark it FREE (:211-213)
      for (int i = 0; i < 3; ++i) q[i] = (q[i] - o[i]) / len * prm->max_ray_length + o[i];
      s = p.observe_pos(q, 0);
    } else {
      s = p.observe_pos(q, 1);
    }
    if (inv) s = inv->p.observe_pos(q, 0);
    if (s != kUndef && !first_visit(p.stamp_occ, p.hstamp_occ, s)) continue;
    const double qv[3] = {q[0] / res, q[1] / res, q[2] / res};
    walk_ray(ov, qv, lo, hi, &cells);
    for (int i = (int)cells.size() - 2; i >= 0; --i) {  // far -> near, endpoint voxel excluded
      const double c[3] = {(cells[i].x + 0.5) * res, (cells[i].y + 0.5) * res, (cells[i].z + 0.5) * res};
      len = norm3(c[0] - o[0], c[1] - o[1], c[2] - o[2]);
      if (len < prm->min_ray_length) break;
      if (len > prm->max_ray_length) continue;
      int f = p.observe_pos(c, 0);
      if (inv) f = inv->p.observe_pos(c, 1);
      if (f != kUndef && !first_visit(p.stamp_free, p.hstamp_free, f)) break;
    }
  }
}

#include "depth_filter.inc"

}  // extern "C"
