// oracle/node_shell_ref.cpp -- TEST INFRASTRUCTURE ONLY (see oracle/oracle_api.h).
//
// examples/fiesta_node_shell.hpp -- the reference node's call sites without ROS -- instantiated with the reference's OWN
// fiesta::ESDFMap and Raycast, compiled verbatim from /root/reference (oracle/Makefile, target `ref`; array and
// -DHASH_TABLE flavours).  The product-side twin is examples/node_shell_demo.cpp (the same header against the HIP
// drop-in class); tests/test_node_shell.py runs both on the same frames.  Nothing here re-implements the map: it drives
// the reference class and dumps what it holds (#define private public around the include, as ref_harness.cpp does).
//   node_shell_ref_{array,hash} frames.bin out_dir
#include <cstdio>
#include <cstring>
#include <iostream>
#include <queue>
#include <sstream>
#include <string>
#include <unordered_map>

#include <Eigen/Eigen>
#include <pcl/kdtree/kdtree_flann.h>
#include <sensor_msgs/PointCloud.h>
#include <visualization_msgs/Marker.h>

#define private public
#include "ESDFMap.h"
#undef private
#include "raycast.h"

#include "../examples/fiesta_node_shell.hpp"

#ifdef HASH_TABLE
constexpr bool kHash = true;
#else
constexpr bool kHash = false;
#endif

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

int main(int argc, char **argv) {
  if (argc < 3) return 1;
  fiesta_shell::Frames fr;
  if (!fr.read(argv[1])) return 1;
  const std::string out = argv[2];
  fiesta_shell::Parameters prm;
  prm.resolution_ = 0.1;
  if (kHash) {
    prm.l_cornor_ = Eigen::Vector3d(-100, -100, -100), prm.r_cornor_ = Eigen::Vector3d(100, 100, 100);
  } else {
    prm.l_cornor_ = Eigen::Vector3d(-6.4, -6.4, -3.2), prm.r_cornor_ = Eigen::Vector3d(6.35, 6.35, 3.15);
  }
  prm.map_size_ = prm.r_cornor_ - prm.l_cornor_;
  prm.radius_ = Eigen::Vector3d(3, 3, 1.5);
  std::stringbuf sink;
  std::streambuf *old = std::cout.rdbuf(&sink);  // the reference prints inside its hot path (src/ESDFMap.cpp:188,237,277,394)
  fiesta_shell::NodeShell<fiesta::ESDFMap, kHash> node(prm);
  fiesta::ESDFMap &map = *node.esdf_map_;
  auto slots = [&]() -> int64_t {
#ifdef HASH_TABLE
    return map.count;
#else
    return map.grid_total_size_;
#endif
  };
  auto voxels = [&](std::vector<int32_t> &vox) {
#ifdef HASH_TABLE
    vox.resize((size_t)3 * map.count);
    for (int i = 0; i < map.count; ++i)
      for (int k = 0; k < 3; ++k) vox[3 * i + k] = map.vox_buffer_[i](k);
#else
    (void)vox;
#endif
  };
  for (int k = 0; k < fr.n_frames; ++k) {
    node.SetFrame(&fr.points[(size_t)3 * fr.n_points * k], (size_t)fr.n_points, &fr.T[16 * k],
                  Eigen::Vector3d(fr.origin[3 * k], fr.origin[3 * k + 1], fr.origin[3 * k + 2]));
    node.RaycastMultithread();
    const int64_t n = slots();
    std::vector<int32_t> vox, hit(map.num_hit_.begin(), map.num_hit_.begin() + n), miss(map.num_miss_.begin(), map.num_miss_.begin() + n);
    voxels(vox);
    dump(out + "/counts" + std::to_string(k) + ".bin", n, kHash ? vox.data() : nullptr, hit.data(), 4 * (size_t)n, miss.data(), 4 * (size_t)n);
    sink.str("");
    node.UpdateEsdfEvent();
    long long ins = -1, del = -1;  // "Insert N\tDelete M" (src/ESDFMap.cpp:277)
    const std::string printed = sink.str();
    const size_t at = printed.find("Insert ");
    if (at != std::string::npos) std::sscanf(printed.c_str() + at, "Insert %lld Delete %lld", &ins, &del);
    std::fprintf(stdout, "%s", "");
    std::fprintf(stderr, "frame %d insert %lld delete %lld\n", k, ins, del);
  }
  std::cout.rdbuf(old);
  const int64_t n = slots();
  std::vector<int32_t> vox;
  voxels(vox);
  std::vector<double> dist(map.distance_buffer_.begin(), map.distance_buffer_.begin() + n);
  std::vector<uint8_t> occ((size_t)n);
  for (int64_t i = 0; i < n; ++i) occ[i] = map.Exist((int)i) ? 1 : 0;
  dump(out + "/field.bin", n, kHash ? vox.data() : nullptr, dist.data(), 8 * (size_t)n, occ.data(), (size_t)n);
  return 0;
}
