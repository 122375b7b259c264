// examples/node_shell_demo.cpp -- the reference node's call sites (examples/fiesta_node_shell.hpp) compiled against the
// HIP drop-in class include/fiesta/ESDFMap.h: every observation goes through ESDFMap::SetOccupancy(Vector3d, int) one
// point at a time, exactly as Fiesta::RaycastProcess does, the per-frame de-duplication is keyed by its return value,
// rays come from the free function Raycast, the timer event is CheckUpdate / SetOriginalRange / UpdateOccupancy /
// UpdateESDF.  tests/test_node_shell.py runs this and oracle/node_shell_ref.cpp (the same header against the verbatim
// reference class) on the same frames and compares the counters, queues and fields the two maps end up with.
//   g++ -std=c++17 -O2 -Iinclude examples/node_shell_demo.cpp -Lfiesta_amd -lfiesta_hip -Wl,-rpath,$PWD/fiesta_amd -o node_shell_demo
//   ./node_shell_demo array|hash frames.bin out_dir
#include <cstdio>
#include <cstring>
#include <string>

#include "fiesta/ESDFMap.h"
using namespace fiesta;  // (Raycast: a free function of the global namespace upstream, of namespace fiesta in the drop-in)
#include "fiesta_node_shell.hpp"

static void dump(const std::string &path, int64_t n, const int32_t *vox, const void *a, size_t a_bytes, const void *b, size_t b_bytes) {
  FILE *f = std::fopen(path.c_str(), "wb");
  if (!f) std::exit(3);
  const int32_t has_vox = vox ? 1 : 0;
  std::fwrite(&n, 8, 1, f);
  std::fwrite(&has_vox, 4, 1, f);
  if (vox) std::fwrite(vox, 4, (size_t)3 * n, f);
  std::fwrite(a, 1, a_bytes, f);
  if (b) std::fwrite(b, 1, b_bytes, f);
  std::fclose(f);
}

template <bool HASH>
static int run(const fiesta_shell::Parameters &prm, const fiesta_shell::Frames &fr, const std::string &out) {
  fiesta_shell::NodeShell<ESDFMap, HASH> node(prm);
  ESDFMap &map = *node.esdf_map_;
  auto check = [](int st) {
    if (st != FIESTA_HIP_OK) {
      std::fprintf(stderr, "%s\n", fiesta_hip_last_error());
      std::exit(2);
    }
  };
  for (int k = 0; k < fr.n_frames; ++k) {
    node.SetFrame(&fr.points[(size_t)3 * fr.n_points * k], (size_t)fr.n_points, &fr.T[16 * k],
                  Eigen::Vector3d(fr.origin[3 * k], fr.origin[3 * k + 1], fr.origin[3 * k + 2]));
    node.RaycastMultithread();
    map.Flush();  // (the drop-in buffers SetOccupancy calls: what the counters hold now is what the reference's hold)
    int64_t n = 0;
    std::vector<int32_t> vox, hit, miss;
    if (HASH) {
      check(fiesta_hip_download_hash(map.Handle(), &n, nullptr, nullptr, nullptr, nullptr));
      vox.resize((size_t)3 * n);
      std::vector<int32_t> d2((size_t)n), coc((size_t)3 * n);
      std::vector<uint8_t> occ((size_t)n);
      check(fiesta_hip_download_hash(map.Handle(), &n, vox.data(), d2.data(), coc.data(), occ.data()));
    } else {
      check(fiesta_hip_grid_total_size(map.Handle(), &n));
    }
    hit.resize((size_t)n), miss.resize((size_t)n);
    check(fiesta_hip_download_counts(map.Handle(), hit.data(), miss.data()));
    dump(out + "/counts" + std::to_string(k) + ".bin", n, HASH ? vox.data() : nullptr, hit.data(), 4 * (size_t)n, miss.data(), 4 * (size_t)n);
    node.UpdateEsdfEvent();
    std::printf("frame %d insert %lld delete %lld\n", k, (long long)map.LastInsertCount(), (long long)map.LastDeleteCount());
  }
  int64_t n = 0;
  if (HASH) {
    check(fiesta_hip_download_hash(map.Handle(), &n, nullptr, nullptr, nullptr, nullptr));
    std::vector<int32_t> vox((size_t)3 * n), d2((size_t)n), coc((size_t)3 * n);
    std::vector<uint8_t> occ((size_t)n);
    check(fiesta_hip_download_hash(map.Handle(), &n, vox.data(), d2.data(), coc.data(), occ.data()));
    dump(out + "/field.bin", n, vox.data(), d2.data(), 4 * (size_t)n, occ.data(), (size_t)n);
  } else {
    check(fiesta_hip_grid_total_size(map.Handle(), &n));
    std::vector<int32_t> d2((size_t)n);
    std::vector<uint8_t> occ((size_t)n);
    check(fiesta_hip_download_field(map.Handle(), d2.data(), nullptr, occ.data(), nullptr));
    dump(out + "/field.bin", n, nullptr, d2.data(), 4 * (size_t)n, occ.data(), (size_t)n);
  }
  std::printf("levels-served updates: engine stats of the last update: levels %lld rounds %lld\n", (long long)map.LastStats().levels,
              (long long)map.LastStats().rounds);
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s array|hash frames.bin out_dir\n", argv[0]);
    return 1;
  }
  fiesta_shell::Frames fr;
  if (!fr.read(argv[2])) {
    std::fprintf(stderr, "cannot read %s\n", argv[2]);
    return 1;
  }
  fiesta_shell::Parameters prm;
  prm.resolution_ = 0.1;
  const bool hash = std::strcmp(argv[1], "hash") == 0;
  if (hash) {  // the hash build's ray box is +-100 m (src/parameters.cpp:45-46)
    prm.l_cornor_ = Eigen::Vector3d(-100, -100, -100), prm.r_cornor_ = Eigen::Vector3d(100, 100, 100);
  } else {
    prm.l_cornor_ = Eigen::Vector3d(-6.4, -6.4, -3.2), prm.r_cornor_ = Eigen::Vector3d(6.35, 6.35, 3.15);
  }
  prm.map_size_ = prm.r_cornor_ - prm.l_cornor_;
  prm.radius_ = Eigen::Vector3d(3, 3, 1.5);
  return hash ? run<true>(prm, fr, argv[3]) : run<false>(prm, fr, argv[3]);
}
