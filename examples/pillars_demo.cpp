// examples/pillars_demo.cpp -- the workload the reference documents in test/test_ESDF_Map.cpp:42-104, written
// against the drop-in class: 25 vertical pillars inserted one UpdateESDF at a time, then 13 deleted again.
// Prints a checksum line the GPU test compares with the same workload driven through the Python mirror.
//   g++ -std=c++17 -O2 -Iinclude examples/pillars_demo.cpp -Lfiesta_amd -lfiesta_hip -Wl,-rpath,$PWD/fiesta_amd -o pillars_demo
#include <chrono>
#include <cstdio>
#include <cstring>
#include <set>

#include "fiesta/ESDFMap.h"

// What a ROS build passes to GetPointCloud / GetSliceMarker are sensor_msgs::PointCloud and visualization_msgs::Marker;
// the getters only need their field names, so plain structs of the same shape do here.
namespace demo_msgs {
struct Header { std::string frame_id; };
struct P32 { float x, y, z; };
struct P64 { double x, y, z; };
struct Rgba { float r, g, b, a; };
struct Quat { double x, y, z, w; };
struct PointCloud { Header header; std::vector<P32> points; };
struct Marker {
  enum { POINTS = 8, MODIFY = 0 };
  Header header;
  int id, type, action;
  P64 scale;
  struct { P64 position; Quat orientation; } pose;
  std::vector<P64> points;
  std::vector<Rgba> colors;
};
}  // namespace demo_msgs

template <class Map>
static int run(Map &map, bool hash) {
  using Eigen::Vector3d;
  using Eigen::Vector3i;
  map.SetParameters(0.70, 0.35, 0.12, 0.97, 0.80);
  map.SetOriginalRange();
  if (!hash) std::printf("grid_total_size_ %d\n", map.grid_total_size_);
  // the per-frame de-duplication of the reference's caller keys on SetOccupancy's return value
  // (include/Fiesta.h:221-232,253-273): it must differ from -10000 and identify the voxel
  std::set<int> keys;
  for (int x = 0; x < 50; ++x)
    for (int y = 0; y < 50; ++y)
      for (int z = 0; z < 25; ++z) {
        const int k = map.SetOccupancy(Vector3i(x, y, z), 0);
        if (k == -10000) {
          std::printf("rejected voxel %d %d %d\n", x, y, z);
          return 1;
        }
        keys.insert(k);
      }
  std::printf("distinct keys %zu\n", keys.size());
  map.UpdateOccupancy(true);
  map.UpdateESDF();
  const int order[25] = {5, 2, 19, 16, 11, 22, 17, 24, 23, 14, 1, 10, 13, 8, 6, 18, 4, 9, 7, 20, 3, 0, 21, 15, 12};
  double total_ms = 0;
  auto pillar = [&](int k, int occ, int cycles) {
    const double px = -4 + 2 * (k / 5) + 0.01, py = -4 + 2 * (k % 5) + 0.01;
    for (int c = 0; c < cycles; ++c) {
      for (int i = 0; i < 50; ++i) map.SetOccupancy(Vector3d(px, py, 0.1 * i + 0.01), occ);
      map.UpdateOccupancy(true);
    }
    const auto t0 = std::chrono::steady_clock::now();
    map.UpdateESDF();
    total_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  };
  for (int k = 0; k < 25; ++k) pillar(order[k], 1, 3);
  std::printf("consistent %d\n", (int)map.CheckConsistency());
  for (int k = 0; k < 13; ++k) pillar(order[k], 0, 6);
  std::printf("consistent %d\n", (int)map.CheckConsistency());
  // checksum of the distance field on a lattice + one trilinear query
  double sum = 0;
  for (int x = 0; x < 50; x += 3)
    for (int y = 0; y < 50; y += 3)
      for (int z = 0; z < 25; z += 3) sum += map.GetDistance(Vector3i(x, y, z));
  Vector3d grad;
  const double d = map.GetDistWithGradTrilinear(Vector3d(0.33, -1.27, 2.2), grad);
  std::printf("checksum %.12f trilinear %.12f grad %.12f %.12f %.12f\n", sum, d, grad(0), grad(1), grad(2));
  if (!hash)
    std::printf("outside %.1f %d\n", map.GetDistance(Vector3d(100, 0, 0)), map.SetOccupancy(Vector3d(100, 0, 0), 1));
  // the visualisation getters with the reference's signatures: 12 pillars of 25 voxels stand; the lower 10 layers of each
  demo_msgs::PointCloud cloud;
  map.GetPointCloud(cloud, 0, 9);
  demo_msgs::Marker slice;
  map.GetSliceMarker(slice, 3, 7, 0 /* colour argument, ignored like in the reference */, 2.0);
  double red = 0;
  for (const auto &c : slice.colors) red += c.r;
  std::printf("cloud %zu in %s, slice marker %d: %zu points, type %d, red %.6f\n", cloud.points.size(), cloud.header.frame_id.c_str(),
              slice.id, slice.points.size(), slice.type, red);
  std::printf("38 UpdateESDF calls: %.3f ms total\n", total_ms);
  return 0;
}

// no argument: the array flavour, ESDFMap(origin, resolution, map_size); "hash": the hash-block flavour,
// ESDFMap(origin, resolution, reserve_size) (the reference selects it with -DHASH_TABLE at compile time)
int main(int argc, char **argv) {
  using Eigen::Vector3d;
  if (argc > 1 && std::strcmp(argv[1], "hash") == 0) {
    fiesta::ESDFMap map(Vector3d(-5, -5, 0), 0.2, 100000);
    return run(map, true);
  }
  fiesta::ESDFMap map(Vector3d(-5, -5, 0), 0.2, Vector3d(10, 10, 5));
  return run(map, false);
}
