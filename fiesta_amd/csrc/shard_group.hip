// fiesta_amd/csrc/shard_group.hip -- ONE map cut into spatial shards: the host protocol, in C++ over RCCL.
//
// The reference is single-process; this is what replaces its ESDFMap when a grid is spread over the GPUs of a node
// (BASELINE config 5: 2048^3 as 2 x 2 x 2 shards of 1024^3).  Layouts: 1 -> 1x1x1, 2 -> 2x1x1, 4 -> 2x2x1, 8 -> 2x2x2
// (with 8 GPUs every pair of shards is adjacent = the fully connected xGMI mesh: every message is one hop on its own
// link).  A shard is an ordinary array-mode DenseMap created with global_grid / shard_lo: owned box + 2-voxel ghost
// layer (the stencil radius of the reference's 24 directions, include/parameters.h:54-68), ids in global coordinates.
//
//   UpdateOccupancy   local k_fuse -> export the occupancy transitions -> all-gather (counts, then the entries padded to
//                     the longest list) -> every shard applies all of them to its replica of the GLOBAL occupancy
//                     bitmap (a closest obstacle may live on any shard; a remote delete arms the invalidation scan).
//   UpdateESDF        seed (insert drain + delete invalidation) on every shard, then sweeps of
//                       diff     every shard compares the owned cells its <= 26 neighbours keep as ghosts (faces, edges AND
//                                corners: one phase, no forwarding) with what it last sent and compacts the CHANGED ones into
//                                {ghost cell index at the receiver, word} entries -- wave ballot + prefix, per neighbour
//                       counts   one all-gather of (entries per neighbour, pending tiles): the receivers' message sizes
//                                and the convergence test in ONE host read per sweep
//                       send     ncclGroupStart; ncclSend/ncclRecv of exactly the changed entries per peer; ncclGroupEnd
//                       apply    ghost cells that differ are replaced, tagged as sources, their tiles woken
//                       relax    the frontier rounds on the woken tiles (dense_map.hip: relax_pending)
//                     until a sweep in which nobody sent anything and no tile is pending anywhere.
//
// Two transports behind the same code: RCCL (one shard per process, one rank per GPU; librccl is dlopen()ed so that a
// single-GPU user never loads it) and "local" (all shards in this process: messages are device-to-device copies) -- the
// GPU tests run 2/4/8 shards multiplexed on one MI355X through the very same diff/apply kernels and sweep loop.
#include "shard_group.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstring>

namespace fiesta {

// ---- librccl, resolved at run time -------------------------------------------------------------------------------
struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  static Rccl &get() {
    static Rccl r;
    if (!r.lib) {
      for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) break;
      }
      if (!r.lib) throw Error(FIESTA_HIP_ERR_DEVICE, std::string("cannot load librccl: ") + dlerror());
      auto sym = [&](const char *n) {
        void *p = dlsym(r.lib, n);
        if (!p) throw Error(FIESTA_HIP_ERR_DEVICE, std::string("librccl lacks ") + n);
        return p;
      };
      r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
      r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
      r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
      r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
      r.CommUserRank = (decltype(r.CommUserRank))sym("ncclCommUserRank");
      r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
      r.Send = (decltype(r.Send))sym("ncclSend");
      r.Recv = (decltype(r.Recv))sym("ncclRecv");
      r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
      r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
      r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    }
    return r;
  }
};
#define FIESTA_RCCL_CHECK(expr)                                                                                   \
  do {                                                                                                            \
    ncclResult_t _r = (expr);                                                                                     \
    if (_r != ncclSuccess) throw Error(FIESTA_HIP_ERR_DEVICE, std::string(#expr) + ": " + Rccl::get().GetErrorString(_r)); \
  } while (0)

void rccl_unique_id(uint8_t out[128]) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  FIESTA_RCCL_CHECK(Rccl::get().GetUniqueId(&id));
  memcpy(out, &id, 128);
}

// ---- geometry of the cut -------------------------------------------------------------------------------------------
static void layout_of(int world, int l[3]) {
  switch (world) {
    case 1: l[0] = 1, l[1] = 1, l[2] = 1; break;
    case 2: l[0] = 2, l[1] = 1, l[2] = 1; break;
    case 4: l[0] = 2, l[1] = 2, l[2] = 1; break;
    case 8: l[0] = 2, l[1] = 2, l[2] = 2; break;
    default: throw Error(FIESTA_HIP_ERR_INVALID, "unsupported shard count (1, 2, 4 or 8)");
  }
}
void shard_box(const int gg[3], int world, int rank, int lo[3], int size[3]) {
  int l[3];
  layout_of(world, l);
  const int c[3] = {rank / (l[1] * l[2]), (rank / l[2]) % l[1], rank % l[2]};
  for (int a = 0; a < 3; ++a) {
    const int base = gg[a] / l[a];
    if (base < 2 * kGhost) throw Error(FIESTA_HIP_ERR_INVALID, "shards thinner than two ghost layers are not supported");
    lo[a] = c[a] * base;
    size[a] = c[a] < l[a] - 1 ? base : gg[a] - base * (l[a] - 1);  // remainders go to the last shard of an axis
  }
}

struct ShardGroup::Link {  // one directed neighbour relation of a local shard
  int peer = -1;                        // global rank of the neighbour
  int32_t send_lo[3], send_hi[3];       // my owned cells the peer keeps as ghosts, MY local coordinates (inclusive)
  int32_t recv_lo[3], recv_dims[3];     // the same cells in the PEER's local array: origin of the box, array extents
  int64_t cells = 0;
  DevBuf<uint32_t> shadow, send, recv;  // last sent words; outgoing / incoming entries (2 words each)
  int64_t recv_cap = 0;                 // cells of the box the peer sends to me (capacity of `recv`)
  int slot = 0;                         // index of this link in the peer's link table (where the peer finds my count)
};
struct ShardGroup::Local {
  DenseMap *map = nullptr;
  int rank = 0;
  std::vector<Link> links;
  DevBuf<unsigned long long> counts;  // per link: entries to send this sweep; [nlinks]: changed ghost cells
  DevBuf<uint32_t> trans;             // exported occupancy transitions
};

// local array of `rank`: global origin and extents (owned box grown by the ghost layers that exist)
static void local_array(const int gg[3], int world, int rank, int org[3], int dims[3], int olo[3], int osz[3]) {
  shard_box(gg, world, rank, olo, osz);
  for (int a = 0; a < 3; ++a) {
    const int glo = olo[a] > 0 ? kGhost : 0, ghi = olo[a] + osz[a] < gg[a] ? kGhost : 0;
    org[a] = olo[a] - glo;
    dims[a] = osz[a] + glo + ghi;
  }
}

static thread_local bool g_precheck_skips_rccl = false;  // (the hosted constructor: one shard per process, no librccl)
void ShardGroup::precheck(const std::vector<DenseMap *> &maps, const std::vector<int> &ranks, int world, bool rccl) {
  if (maps.empty() || maps.size() != ranks.size()) throw Error(FIESTA_HIP_ERR_INVALID, "shard group: bad shard list");
  const Geom &g0 = maps[0]->geom();
  if (!g0.sharded) throw Error(FIESTA_HIP_ERR_INVALID, "shard group: maps must be created as shards (global_grid / shard_lo)");
  const int gg[3] = {g0.GX, g0.GY, g0.GZ};
  int l[3];
  layout_of(world, l);
  if (rccl && maps.size() != 1) throw Error(FIESTA_HIP_ERR_INVALID, "shard group: one shard per process under RCCL / a hosted transport");
  if (!rccl && (int)maps.size() != world) throw Error(FIESTA_HIP_ERR_INVALID, "shard group: without a transport every shard must be local");
  for (size_t i = 0; i < maps.size(); ++i) {
    if (ranks[i] < 0 || ranks[i] >= world) throw Error(FIESTA_HIP_ERR_INVALID, "shard group: rank out of range");
    const Geom &g = maps[i]->geom();
    int org[3], dims[3], olo[3], osz[3];
    local_array(gg, world, ranks[i], org, dims, olo, osz);
    if (!g.sharded || g.GX != gg[0] || g.GY != gg[1] || g.GZ != gg[2] || org[0] != g.gx0 || org[1] != g.gy0 || org[2] != g.gz0 ||
        dims[0] != g.nx || dims[1] != g.ny || dims[2] != g.nz)
      throw Error(FIESTA_HIP_ERR_INVALID, "shard group: a shard's box does not match the regular cut of the global grid");
  }
  if (rccl && !g_precheck_skips_rccl) (void)Rccl::get();  // (throws when librccl cannot be loaded)
}

void ShardGroup::comm_info(int *nranks, int *rank) const {
  int n = 0, r = locals_.empty() ? 0 : locals_[0]->rank;
  if (comm_) {
    FIESTA_RCCL_CHECK(Rccl::get().CommCount((ncclComm_t)comm_, &n));
    FIESTA_RCCL_CHECK(Rccl::get().CommUserRank((ncclComm_t)comm_, &r));
  }
  if (nranks) *nranks = n;
  if (rank) *rank = r;
}

ShardGroup::ShardGroup(const std::vector<DenseMap *> &maps, const std::vector<int> &ranks, int world, const uint8_t *rccl_id,
                       const fiesta_hip_shard_transport *hosted)
    : world_(world) {
  if (hosted && rccl_id) throw Error(FIESTA_HIP_ERR_INVALID, "shard group: RCCL or a hosted transport, not both");
  if (hosted) {
    host_ = *hosted, hosted_ = true;
    g_precheck_skips_rccl = true;
  }
  try {
    precheck(maps, ranks, world, rccl_id != nullptr || hosted_);
  } catch (...) {
    g_precheck_skips_rccl = false;
    throw;
  }
  g_precheck_skips_rccl = false;
  const Geom &g0 = maps[0]->geom();
  gg_[0] = g0.GX, gg_[1] = g0.GY, gg_[2] = g0.GZ;
  int l[3];
  layout_of(world, l);
  for (size_t i = 0; i < maps.size(); ++i) {
    auto L = std::make_unique<Local>();
    L->map = maps[i];
    L->rank = ranks[i];
    const Geom &g = maps[i]->geom();
    int org[3], dims[3], olo[3], osz[3];
    local_array(gg_, world, ranks[i], org, dims, olo, osz);  // (validated by precheck)
    (void)g;
    const int c[3] = {ranks[i] / (l[1] * l[2]), (ranks[i] / l[2]) % l[1], ranks[i] % l[2]};
    // neighbours in a fixed order (dx, dy, dz lexicographic): both sides enumerate their links the same way
    for (int dx = -1; dx <= 1; ++dx)
      for (int dy = -1; dy <= 1; ++dy)
        for (int dz = -1; dz <= 1; ++dz) {
          if (!dx && !dy && !dz) continue;
          const int n[3] = {c[0] + dx, c[1] + dy, c[2] + dz};
          if (n[0] < 0 || n[0] >= l[0] || n[1] < 0 || n[1] >= l[1] || n[2] < 0 || n[2] >= l[2]) continue;
          Link k;
          k.peer = (n[0] * l[1] + n[1]) * l[2] + n[2];
          int porg[3], pdims[3], polo[3], posz[3];
          local_array(gg_, world, k.peer, porg, pdims, polo, posz);
          // my owned box  intersected with  the peer's local array (its ghost layer towards me), global coordinates
          int64_t cells = 1, rcells = 1;
          for (int a = 0; a < 3; ++a) {
            const int lo = std::max(olo[a], porg[a]), hi = std::min(olo[a] + osz[a], porg[a] + pdims[a]) - 1;
            k.send_lo[a] = lo - org[a], k.send_hi[a] = hi - org[a];
            k.recv_lo[a] = lo - porg[a];
            k.recv_dims[a] = pdims[a];
            cells *= std::max(0, hi - lo + 1);
            // what the peer sends me: its owned box intersected with MY local array
            const int rlo = std::max(polo[a], org[a]), rhi = std::min(polo[a] + posz[a], org[a] + dims[a]) - 1;
            rcells *= std::max(0, rhi - rlo + 1);
          }
          k.cells = cells;
          k.recv_cap = rcells;
          L->links.push_back(std::move(k));
        }
    // where does the peer list ME?  (same enumeration order on its side)
    for (Link &k : L->links) {
      const int pc[3] = {k.peer / (l[1] * l[2]), (k.peer / l[2]) % l[1], k.peer % l[2]};
      int slot = 0;
      bool found = false;
      for (int dx = -1; dx <= 1 && !found; ++dx)
        for (int dy = -1; dy <= 1 && !found; ++dy)
          for (int dz = -1; dz <= 1 && !found; ++dz) {
            if (!dx && !dy && !dz) continue;
            const int n[3] = {pc[0] + dx, pc[1] + dy, pc[2] + dz};
            if (n[0] < 0 || n[0] >= l[0] || n[1] < 0 || n[1] >= l[1] || n[2] < 0 || n[2] >= l[2]) continue;
            if ((n[0] * l[1] + n[1]) * l[2] + n[2] == L->rank) {
              found = true;
              break;
            }
            ++slot;
          }
      k.slot = slot;
    }
    hipStream_t s = L->map->stream();
    FIESTA_HIP_CHECK(hipSetDevice(L->map->device()));
    for (Link &k : L->links) {
      k.shadow.ensure((size_t)k.cells, s);
      k.send.ensure((size_t)k.cells * 2, s);
      k.recv.ensure((size_t)k.recv_cap * 2, s);
      // ghost cells start out "never observed", and so does what was "last sent"
      FIESTA_HIP_CHECK(hipMemsetAsync(k.shadow.p, 0xFF, (size_t)k.cells * sizeof(uint32_t), s));
    }
    L->counts.ensure(kMaxLinks + 2, s);
    FIESTA_HIP_CHECK(hipStreamSynchronize(s));
    L->map->set_alone_in_group(world_ == 1);
    L->map->bulk_reserve(margin_);  // (the transform's scratch: not inside the first update)
    locals_.push_back(std::move(L));
  }
  FIESTA_HIP_CHECK(hipHostMalloc((void **)&h_table_, (size_t)world_ * kRow * sizeof(long long)));
  if (rccl_id) {
    Local &L = *locals_[0];
    FIESTA_HIP_CHECK(hipSetDevice(L.map->device()));
    ncclUniqueId id;
    memcpy(&id, rccl_id, sizeof(id));
    ncclComm_t comm = nullptr;
    FIESTA_RCCL_CHECK(Rccl::get().CommInitRank(&comm, world_, id, L.rank));
    comm_ = comm;
    FIESTA_HIP_CHECK(hipMalloc((void **)&d_row_, kRow * sizeof(long long)));
    FIESTA_HIP_CHECK(hipMalloc((void **)&d_table_, (size_t)world_ * kRow * sizeof(long long)));
  }
}

ShardGroup::~ShardGroup() {
  if (comm_) (void)Rccl::get().CommDestroy((ncclComm_t)comm_);
  if (d_row_) (void)hipFree(d_row_);
  if (d_table_) (void)hipFree(d_table_);
  if (h_table_) (void)hipHostFree(h_table_);
}

// Every shard contributes one row of kRow numbers; afterwards h_table_[rank * kRow + j] holds them all on the host.
// (RCCL: ONE small all-gather + one stream synchronisation; local: the rows are simply written in place.)
void ShardGroup::gather_rows(const std::vector<std::vector<long long>> &rows) {
  if (comm_) {  // (a group of one on the local transport has nobody to ask; one that was GIVEN a communicator uses it --
                //  that is how the RCCL entry points are exercised on a one-GPU box)
    Local &L = *locals_[0];
    hipStream_t s = L.map->stream();
    memcpy(&h_table_[(size_t)L.rank * kRow], rows[0].data(), kRow * sizeof(long long));
    FIESTA_HIP_CHECK(hipMemcpyAsync(d_row_, &h_table_[(size_t)L.rank * kRow], kRow * sizeof(long long), hipMemcpyHostToDevice, s));
    FIESTA_RCCL_CHECK(Rccl::get().AllGather(d_row_, d_table_, kRow, ncclInt64, (ncclComm_t)comm_, s));
    FIESTA_HIP_CHECK(hipMemcpyAsync(h_table_, d_table_, (size_t)world_ * kRow * sizeof(long long), hipMemcpyDeviceToHost, s));
    FIESTA_HIP_CHECK(hipStreamSynchronize(s));
  } else if (hosted_) {
    std::vector<long long> all((size_t)world_ * kRow);
    host_all_gather(rows[0].data(), all.data(), kRow * (int64_t)sizeof(long long));
    memcpy(h_table_, all.data(), all.size() * sizeof(long long));
  } else {
    for (size_t i = 0; i < locals_.size(); ++i)
      memcpy(&h_table_[(size_t)locals_[i]->rank * kRow], rows[i].data(), kRow * sizeof(long long));
  }
}

void ShardGroup::host_all_gather(const void *send, void *recv, int64_t bytes) {
  if (host_.all_gather(host_.ctx, send, recv, bytes) != 0) throw Error(FIESTA_HIP_ERR_DEVICE, "shard group: the hosted transport's all_gather failed");
}

ShardGroup::Local *ShardGroup::find_local(int rank) {
  for (auto &L : locals_)
    if (L->rank == rank) return L.get();
  return nullptr;
}

bool ShardGroup::update_occupancy(bool global_map, int64_t *n_ins, int64_t *n_del) {
  // 1. local fusion; 2. every shard's transitions to every shard
  if (world_ == 1 && !remote()) {  // nobody to tell: the shard's own fusion already keeps its replica of the global bitmap
    int64_t ni = 0, nd = 0;
    const bool any = locals_[0]->map->update_occupancy(global_map, &ni, &nd);
    if (n_ins) *n_ins = ni;
    if (n_del) *n_del = nd;
    return any;
  }
  std::vector<std::vector<long long>> rows(locals_.size(), std::vector<long long>(kRow, 0));
  for (size_t i = 0; i < locals_.size(); ++i) {
    Local &L = *locals_[i];
    int64_t ni = 0, nd = 0;
    L.map->update_occupancy(global_map, &ni, &nd);
    const int64_t n = L.map->export_transitions(nullptr, 0);
    L.trans.ensure((size_t)std::max<int64_t>(1, 2 * n), L.map->stream());
    if (n) L.map->export_transitions(L.trans.p, n);
    rows[i][0] = n, rows[i][1] = ni, rows[i][2] = nd;
  }
  gather_rows(rows);
  long long tot_i = 0, tot_d = 0, max_n = 0;
  for (int r = 0; r < world_; ++r) {
    max_n = std::max(max_n, h_table_[(size_t)r * kRow]);
    tot_i += h_table_[(size_t)r * kRow + 1];
    tot_d += h_table_[(size_t)r * kRow + 2];
  }
  if (max_n > 0) {
    if (comm_) {
      Local &L = *locals_[0];
      hipStream_t s = L.map->stream();
      L.trans.ensure((size_t)2 * max_n, s, (size_t)2 * h_table_[(size_t)L.rank * kRow]);
      gathered_.ensure((size_t)2 * max_n * world_, s);
      FIESTA_RCCL_CHECK(Rccl::get().AllGather(L.trans.p, gathered_.p, (size_t)2 * max_n, ncclUint32, (ncclComm_t)comm_, s));
      for (int r = 0; r < world_; ++r)
        L.map->apply_transitions(gathered_.p + (size_t)2 * max_n * r, h_table_[(size_t)r * kRow]);
    } else if (hosted_) {  // the same all-gather, through host buffers and the caller's transport
      Local &L = *locals_[0];
      hipStream_t s = L.map->stream();
      const size_t words = (size_t)2 * max_n;
      h_words_.assign(words, 0u);
      const long long mine = h_table_[(size_t)L.rank * kRow];
      if (mine) L.map->copy_to_host(h_words_.data(), L.trans.p, (size_t)2 * mine * sizeof(uint32_t));
      h_gathered_.resize(words * world_);
      host_all_gather(h_words_.data(), h_gathered_.data(), (int64_t)(words * sizeof(uint32_t)));
      gathered_.ensure(words * world_, s);
      L.map->copy_to_device(gathered_.p, h_gathered_.data(), words * world_ * sizeof(uint32_t));
      for (int r = 0; r < world_; ++r) L.map->apply_transitions(gathered_.p + words * r, h_table_[(size_t)r * kRow]);
    } else {
      for (auto &src : locals_) src->map->synchronize();
      for (auto &dst : locals_)
        for (auto &src : locals_) dst->map->apply_transitions(src->trans.p, h_table_[(size_t)src->rank * kRow]);
    }
  }
  if (n_ins) *n_ins = tot_i;
  if (n_del) *n_del = tot_d;
  return tot_i != 0 || tot_d != 0;
}

void ShardGroup::update_esdf(fiesta_hip_stats *st, int32_t *sweeps_out, int64_t *entries_out) {
  fiesta_hip_stats total;
  memset(&total, 0, sizeof(total));
  auto add = [&](const fiesta_hip_stats &s, bool first) {
    if (first) total.inserted += s.inserted, total.deleted += s.deleted;
    total.invalidated += s.invalidated;
    total.tile_visits += s.tile_visits, total.sweeps += s.sweeps, total.voxel_writes += s.voxel_writes;
    total.relax_ms += s.relax_ms, total.relax_launches += s.relax_launches;
  };
  const auto h0 = std::chrono::steady_clock::now();
  int64_t rounds = 0;
  // ---- engine choice, identical on every rank.  On a fully observed map a large delta is served by the bulk feature
  // transform (ft_kernels.hpp), shard by shard and WITHOUT any exchange: every shard transforms its own array grown by a
  // margin, reading the replicated global occupancy bitmap; the result is exact once every voxel found its obstacle
  // within the margin (checked from the largest distance written; otherwise the margin doubles).
  {
    std::vector<std::vector<long long>> rows(locals_.size(), std::vector<long long>(kRow, 0));
    for (size_t i = 0; i < locals_.size(); ++i) {
      unsigned long long ni = 0, nd = 0;
      long long nocc = 0;
      bool el = false;
      locals_[i]->map->bulk_probe(&ni, &nd, &nocc, &el);
      rows[i][0] = el ? 1 : 0, rows[i][1] = (long long)ni, rows[i][2] = (long long)nd, rows[i][3] = nocc;
      rows[i][4] = locals_[i]->map->bulk_pinned() ? 2 : locals_[i]->map->update_engine();
      // what the cost model needs, as numbers every rank will see (the decision used to read rank-local state --
      // the last transform's time, this shard's voxel count -- and two ranks could enter different collectives)
      rows[i][5] = (long long)locals_[i]->map->total();
      rows[i][6] = (long long)std::llround(locals_[i]->map->ft_last_ms() * 1e6);   // ns
      rows[i][7] = locals_[i]->map->bulk_ratio() >= 0 ? (long long)std::llround(locals_[i]->map->bulk_ratio() * 1e9) : -1;
    }
    gather_rows(rows);
    bool all = true, force = true;
    long long ni = 0, nd = 0, nocc = 0, n_max = 0, ft_ns_max = 0, ratio = -1;
    for (int r = 0; r < world_; ++r) {
      const long long *t = &h_table_[(size_t)r * kRow];
      all = all && t[0] != 0;
      force = force && t[4] == 2;
      ni += t[1], nd += t[2], nocc += t[3];
      n_max = std::max(n_max, t[5]), ft_ns_max = std::max(ft_ns_max, t[6]), ratio = std::max(ratio, t[7]);
    }
    // (the cost model per shard: its share of the delta and of the obstacles against the LARGEST shard's array and the
    //  SLOWEST shard's last transform -- all from the gathered table, so every rank computes the same answer)
    const bool want = all && (ni + nd) > 0 &&
                      (force || DenseMap::bulk_pays_model((double)(ni + nd) / world_, (double)nocc / world_, (double)n_max,
                                                          (double)ft_ns_max * 1e-6, ratio >= 0 ? (double)ratio * 1e-9 : -1.0));
    for (int m = margin_; want; m *= 2) {
      std::vector<fiesta_hip_stats> ss(locals_.size());
      for (size_t i = 0; i < locals_.size(); ++i) {
        bool exact = false;
        const bool ok = locals_[i]->map->bulk_try(&ss[i], m, &exact);
        rows[i].assign(kRow, 0);
        rows[i][0] = ok ? 1 : 0, rows[i][1] = exact ? 1 : 0, rows[i][2] = ok ? ss[i].ft_max_d2 : 0;
      }
      gather_rows(rows);
      bool ok = true, exact = true;
      long long dmax2 = 0;
      for (int r = 0; r < world_; ++r) {
        ok = ok && h_table_[(size_t)r * kRow] != 0;
        exact = exact && h_table_[(size_t)r * kRow + 1] != 0;
        dmax2 = std::max(dmax2, h_table_[(size_t)r * kRow + 2]);
      }
      if (!ok) break;  // a region outgrew the transform's 1024-voxel limit: the frontier rounds below take over
      if (!exact) continue;
      size_t cells_shards = 0;  // shards of this process whose transform was the cell transform (nn_kernels.hpp)
      for (size_t i = 0; i < locals_.size(); ++i) {
        locals_[i]->map->bulk_commit(&ss[i]);
        // The transform rewrote owned AND ghost cells on every shard behind the exchange's back: what was "last sent" no
        // longer describes what the peers hold (a later frontier update could find w == shadow for a cell whose ghost
        // copy differs and send nothing -- ADVICE r2).  Forget it: the next diff resends the boundary, and
        // halo_apply_sparse skips the cells that are equal anyway.
        for (Link &k : locals_[i]->links) {
          FIESTA_HIP_CHECK(hipSetDevice(locals_[i]->map->device()));
          FIESTA_HIP_CHECK(hipMemsetAsync(k.shadow.p, 0xFF, (size_t)k.cells * sizeof(uint32_t), locals_[i]->map->stream()));
        }
        total.inserted += ss[i].inserted, total.deleted += ss[i].deleted;
        total.relax_ms = std::max(total.relax_ms, ss[i].relax_ms);
        total.ft_rows_ms = std::max(total.ft_rows_ms, ss[i].ft_rows_ms);
        total.ft_plane_ms = std::max(total.ft_plane_ms, ss[i].ft_plane_ms);
        total.ft_x_ms = std::max(total.ft_x_ms, ss[i].ft_x_ms);
        total.nn_cells_ms = std::max(total.nn_cells_ms, ss[i].nn_cells_ms);
        total.nn_lists_ms = std::max(total.nn_lists_ms, ss[i].nn_lists_ms);
        total.nn_fill_ms = std::max(total.nn_fill_ms, ss[i].nn_fill_ms);
        total.nn_entries += ss[i].nn_entries, total.nn_failed += ss[i].nn_failed;
        cells_shards += ss[i].cells ? 1 : 0;
        for (int k = 0; k < 6; ++k) total.ft_overflow[k] += ss[i].ft_overflow[k];
        total.relax_launches += ss[i].relax_launches;
      }
      total.inserted = ni, total.deleted = nd;
      total.bulk = 1;
      total.cells = cells_shards == locals_.size() ? 1 : 0;
      total.ft_max_d2 = dmax2;
      // next time start from what this scene needed (+ slack), in steps of 16 voxels
      const int need = (int)std::ceil(std::sqrt((double)dmax2)) + 8;
      margin_ = std::min(512, std::max(16, (need + 15) / 16 * 16));
      total.host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
      total.device_ms = total.host_ms;
      if (st) *st = total;
      if (sweeps_out) *sweeps_out = 0;
      if (entries_out) *entries_out = 0;
      return;
    }
  }
  for (auto &L : locals_) {
    fiesta_hip_stats s;
    L->map->update_esdf(&s, /*seed_only=*/true);
    add(s, true);
  }
  int32_t sweeps = 0;
  int64_t sent_total = 0;
  for (;;) {
    // ---- diff: changed boundary cells -> entries, per neighbour
    for (auto &Lp : locals_) {
      Local &L = *Lp;
      hipStream_t s = L.map->stream();
      FIESTA_HIP_CHECK(hipSetDevice(L.map->device()));
      FIESTA_HIP_CHECK(hipMemsetAsync(L.counts.p, 0, (kMaxLinks + 2) * sizeof(unsigned long long), s));
      for (size_t k = 0; k < L.links.size(); ++k) {
        Link &ln = L.links[k];
        L.map->halo_diff(ln.send_lo, ln.send_hi, ln.shadow.p, ln.recv_lo, ln.recv_dims, ln.send.p, &L.counts.p[k]);
      }
    }
    // ---- counts (the receivers' message sizes) + pending tiles: one gather, one host read
    std::vector<std::vector<long long>> rows(locals_.size(), std::vector<long long>(kRow, 0));
    if (comm_) {
      // the row is assembled ON THE DEVICE (entry counts + the pending-tile counter) and goes straight into the
      // all-gather: ONE synchronisation per sweep (r02: one for the counts, one for the tiles, one for the gather)
      Local &L = *locals_[0];
      hipStream_t s = L.map->stream();
      static_assert(sizeof(long long) == sizeof(unsigned long long), "row layout");
      FIESTA_HIP_CHECK(hipMemsetAsync(d_row_, 0, kRow * sizeof(long long), s));
      FIESTA_HIP_CHECK(hipMemcpyAsync(d_row_, L.counts.p, L.links.size() * sizeof(long long), hipMemcpyDeviceToDevice, s));
      FIESTA_HIP_CHECK(hipMemcpyAsync(d_row_ + kMaxLinks, L.map->counter_dev(C_LIST0), sizeof(long long), hipMemcpyDeviceToDevice, s));
      FIESTA_RCCL_CHECK(Rccl::get().AllGather(d_row_, d_table_, kRow, ncclInt64, (ncclComm_t)comm_, s));
      FIESTA_HIP_CHECK(hipMemcpyAsync(h_table_, d_table_, (size_t)world_ * kRow * sizeof(long long), hipMemcpyDeviceToHost, s));
      FIESTA_HIP_CHECK(hipStreamSynchronize(s));
    } else {
      for (size_t i = 0; i < locals_.size(); ++i) {  // local transport: one copy + one synchronisation per shard
        Local &L = *locals_[i];
        unsigned long long h[kMaxLinks + 1];
        FIESTA_HIP_CHECK(hipMemcpyAsync(h, L.counts.p, kMaxLinks * sizeof(unsigned long long), hipMemcpyDeviceToHost, L.map->stream()));
        FIESTA_HIP_CHECK(hipMemcpyAsync(&h[kMaxLinks], L.map->counter_dev(C_LIST0), sizeof(unsigned long long), hipMemcpyDeviceToHost, L.map->stream()));
        FIESTA_HIP_CHECK(hipStreamSynchronize(L.map->stream()));
        for (size_t k = 0; k < L.links.size(); ++k) rows[i][k] = (long long)h[k];
        rows[i][kMaxLinks] = (long long)h[kMaxLinks];
      }
      gather_rows(rows);
    }
    long long any = 0;
    for (int r = 0; r < world_; ++r)
      for (int j = 0; j <= kMaxLinks; ++j) any += h_table_[(size_t)r * kRow + j];
    if (!any) break;
    ++sweeps;
    // ---- send / receive exactly the changed entries
    if (comm_) {
      Local &L = *locals_[0];
      hipStream_t s = L.map->stream();
      FIESTA_RCCL_CHECK(Rccl::get().GroupStart());
      for (size_t k = 0; k < L.links.size(); ++k) {
        Link &ln = L.links[k];
        const long long ns = h_table_[(size_t)L.rank * kRow + k], nr = h_table_[(size_t)ln.peer * kRow + ln.slot];
        if (ns) FIESTA_RCCL_CHECK(Rccl::get().Send(ln.send.p, (size_t)2 * ns, ncclUint32, ln.peer, (ncclComm_t)comm_, s));
        if (nr) FIESTA_RCCL_CHECK(Rccl::get().Recv(ln.recv.p, (size_t)2 * nr, ncclUint32, ln.peer, (ncclComm_t)comm_, s));
        sent_total += ns;
      }
      FIESTA_RCCL_CHECK(Rccl::get().GroupEnd());
      for (size_t k = 0; k < L.links.size(); ++k) {
        Link &ln = L.links[k];
        L.map->halo_apply_sparse(ln.recv.p, h_table_[(size_t)ln.peer * kRow + ln.slot], &L.counts.p[kMaxLinks]);
      }
    } else if (hosted_) {  // the same messages, staged through host memory and handed to the caller's transport
      Local &L = *locals_[0];
      const size_t nl = L.links.size();
      std::vector<std::vector<uint32_t>> out(nl), in(nl);
      std::vector<int32_t> peers(nl);
      std::vector<const void *> sp(nl);
      std::vector<void *> rp(nl);
      std::vector<int64_t> sb(nl), rb(nl);
      for (size_t k = 0; k < nl; ++k) {
        Link &ln = L.links[k];
        const long long ns = h_table_[(size_t)L.rank * kRow + k], nr = h_table_[(size_t)ln.peer * kRow + ln.slot];
        out[k].resize((size_t)2 * ns), in[k].resize((size_t)2 * nr);
        if (ns) L.map->copy_to_host(out[k].data(), ln.send.p, (size_t)2 * ns * sizeof(uint32_t));
        peers[k] = ln.peer, sp[k] = out[k].data(), rp[k] = in[k].data();
        sb[k] = (int64_t)(2 * ns * sizeof(uint32_t)), rb[k] = (int64_t)(2 * nr * sizeof(uint32_t));
        sent_total += ns;
      }
      if (host_.exchange(host_.ctx, (int32_t)nl, peers.data(), sp.data(), sb.data(), rp.data(), rb.data()) != 0)
        throw Error(FIESTA_HIP_ERR_DEVICE, "shard group: the hosted transport's exchange failed");
      for (size_t k = 0; k < nl; ++k) {
        Link &ln = L.links[k];
        const long long nr = h_table_[(size_t)ln.peer * kRow + ln.slot];
        if (!nr) continue;
        L.map->copy_to_device(ln.recv.p, in[k].data(), (size_t)2 * nr * sizeof(uint32_t));
        L.map->halo_apply_sparse(ln.recv.p, nr, &L.counts.p[kMaxLinks]);
      }
    } else {
      for (auto &Lp : locals_) Lp->map->synchronize();  // (every send buffer is complete before anybody copies from it)
      for (auto &Lp : locals_) {
        Local &L = *Lp;
        FIESTA_HIP_CHECK(hipSetDevice(L.map->device()));
        for (size_t k = 0; k < L.links.size(); ++k) {
          Link &ln = L.links[k];
          Local *P = find_local(ln.peer);
          const long long nr = h_table_[(size_t)ln.peer * kRow + ln.slot];
          sent_total += h_table_[(size_t)L.rank * kRow + k];
          if (!nr) continue;
          FIESTA_HIP_CHECK(hipMemcpyAsync(ln.recv.p, P->links[ln.slot].send.p, (size_t)2 * nr * sizeof(uint32_t),
                                          hipMemcpyDeviceToDevice, L.map->stream()));
          L.map->halo_apply_sparse(ln.recv.p, nr, &L.counts.p[kMaxLinks]);
        }
      }
    }
    // ---- relax what the new ghost values woke up
    int64_t r_max = 0;
    for (auto &L : locals_) {
      fiesta_hip_stats s;
      int64_t pending = 0;
      L->map->relax_pending(&s, &pending);
      add(s, false);
      r_max = std::max<int64_t>(r_max, s.rounds);
    }
    rounds += r_max;
    if (sweeps > 100000) throw Error(FIESTA_HIP_ERR_STATE, "shard group: ghost exchange does not converge");
  }
  total.rounds = rounds;
  total.host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
  total.device_ms = total.host_ms;
  if (st) *st = total;
  if (sweeps_out) *sweeps_out = sweeps;
  if (entries_out) *entries_out = sent_total;
}

}  // namespace fiesta
