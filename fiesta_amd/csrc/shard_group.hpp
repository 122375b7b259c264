// fiesta_amd/csrc/shard_group.hpp -- one map cut into spatial shards: host protocol over RCCL (see shard_group.hip).
#pragma once
#include <chrono>
#include <memory>
#include <vector>

#include "dense_map.hpp"

namespace fiesta {

constexpr int kGhost = 2;  // stencil radius of the reference's 24 directions

void rccl_unique_id(uint8_t out[128]);
// owned box of shard `rank` in the regular cut of a global grid into `world` shards
void shard_box(const int global_grid[3], int world, int rank, int lo[3], int size[3]);

class ShardGroup {
 public:
  // maps[i] is the shard of global rank ranks[i].  rccl_id != nullptr: one shard per process, `world` ranks over RCCL.
  // rccl_id == nullptr: all `world` shards live in this process (tests, N shards multiplexed on one GPU).
  // hosted != nullptr (and rccl_id == nullptr): one shard per process, messages handed to the CALLER's functions through
  // host buffers (fiesta_hip_shard_transport: an all-gather and a neighbour exchange) -- the same sweep loop, sparse
  // diff / apply and convergence test as over RCCL, across process boundaries that RCCL refuses on one GPU (two ranks on
  // one device) or that have no RCCL at all; what the multi-process tests bind to torch.distributed / gloo.
  ShardGroup(const std::vector<DenseMap *> &maps, const std::vector<int> &ranks, int world, const uint8_t *rccl_id,
             const fiesta_hip_shard_transport *hosted = nullptr);
  ~ShardGroup();
  // Everything the constructor checks LOCALLY (shard boxes against the regular cut, set-up rules, librccl loadable when
  // `rccl`), without the collective ncclCommInitRank: ranks agree on the outcome of this first, so that one rank's local
  // failure cannot leave the others blocked inside the communicator set-up.  Throws like the constructor would.
  static void precheck(const std::vector<DenseMap *> &maps, const std::vector<int> &ranks, int world, bool rccl);
  // ranks the RCCL communicator reports (ncclCommCount; 0 on the local transport) and this process's rank in it
  void comm_info(int *nranks, int *rank) const;
  bool update_occupancy(bool global_map, int64_t *n_ins, int64_t *n_del);
  void update_esdf(fiesta_hip_stats *st, int32_t *sweeps, int64_t *entries_sent);

 private:
  static constexpr int kMaxLinks = 26, kRow = 28;
  struct Link;
  struct Local;
  void gather_rows(const std::vector<std::vector<long long>> &rows);
  Local *find_local(int rank);
  int world_ = 1;
  int margin_ = 32;  // bulk path: voxels added around a shard's array (adapts to the scene)
  int gg_[3] = {0, 0, 0};
  std::vector<std::unique_ptr<Local>> locals_;
  void *comm_ = nullptr;  // ncclComm_t
  bool hosted_ = false;   // messages go through host_ (the caller's transport)
  fiesta_hip_shard_transport host_{};
  bool remote() const { return comm_ != nullptr || hosted_; }  // peers live in other processes
  void host_all_gather(const void *send, void *recv, int64_t bytes);
  std::vector<uint32_t> h_words_, h_gathered_;
  long long *h_table_ = nullptr, *d_row_ = nullptr, *d_table_ = nullptr;
  DevBuf<uint32_t> gathered_;
};

}  // namespace fiesta
