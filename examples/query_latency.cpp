// examples/query_latency.cpp -- what ONE call of the kept signatures costs through the drop-in class
// (fiesta::ESDFMap::GetDistance / GetDistWithGradTrilinear, include/fiesta/ESDFMap.h): the planner-side view of
// src/ESDFMap.cpp:467-540.  Builds a 256^3 map with scattered obstacles, then times
//   walk     a random walk of small steps (a trajectory being optimised: the same few bricks again and again)
//   uniform  positions drawn uniformly over the map (the worst case for the host-side brick cache: 4096 bricks, 2048 slots)
//   batch    the same positions through the batch entry point, for scale
// Prints one JSON object.  Build: g++ -std=c++17 -O2 -Iinclude examples/query_latency.cpp -Lfiesta_amd -lfiesta_hip
#include <chrono>
#include <cstdio>
#include <random>
#include <vector>

#include "fiesta/ESDFMap.h"

int main() {
  const int n = 256;
  const double res = 0.1;
  fiesta::ESDFMap map(Eigen::Vector3d(0, 0, 0), res, Eigen::Vector3d(n * res, n * res, n * res));
  map.SetParameters(0.70, 0.35, 0.12, 0.97, 0.80);
  map.SetOriginalRange();
  std::mt19937 rng(12345);
  std::uniform_int_distribution<int> vox(0, n - 1);
  {  // every voxel observed free once (the C ABI's box form), then 6250 obstacle voxels: config 2's density
    const int32_t lo[3] = {0, 0, 0}, hi[3] = {n - 1, n - 1, n - 1};
    fiesta_hip_set_occupancy_box(map.Handle(), lo, hi, 0);
    map.UpdateOccupancy(true);
    map.UpdateESDF();
    std::vector<int32_t> v(3 * 6250);
    for (auto &c : v) c = vox(rng);
    std::vector<int32_t> occ(6250, 1);
    for (int k = 0; k < 3; ++k) {
      fiesta_hip_set_occupancy_vox(map.Handle(), v.data(), occ.data(), 6250, nullptr);
      map.UpdateOccupancy(true);
    }
    map.UpdateESDF();
  }
  using clk = std::chrono::steady_clock;
  auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  std::uniform_real_distribution<double> u(0.3, n * res - 0.3), step(-0.02, 0.02);
  const int N = 1000000;
  Eigen::Vector3d g(0, 0, 0), p(u(rng), u(rng), u(rng));
  int64_t f0 = 0, f1 = 0, f2 = 0;
  fiesta_hip_host_cache_fetches(map.Handle(), &f0);
  double acc = 0;
  auto t0 = clk::now();
  for (int i = 0; i < N; ++i) {
    for (int k = 0; k < 3; ++k) {
      p(k) += step(rng);
      if (p(k) < 0.3 || p(k) > n * res - 0.3) p(k) = u(rng);
    }
    acc += map.GetDistWithGradTrilinear(p, g) + g(0);
  }
  auto t1 = clk::now();
  fiesta_hip_host_cache_fetches(map.Handle(), &f1);
  std::vector<double> pos(3 * (size_t)N);
  for (auto &c : pos) c = u(rng);
  auto t2 = clk::now();
  for (int i = 0; i < N; ++i) acc += map.GetDistWithGradTrilinear(Eigen::Vector3d(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]), g) + g(1);
  auto t3 = clk::now();
  fiesta_hip_host_cache_fetches(map.Handle(), &f2);
  std::vector<double> d(N), gr(3 * (size_t)N);
  auto t4 = clk::now();
  map.GetDistWithGradTrilinearBatch(pos.data(), N, d.data(), gr.data());
  auto t5 = clk::now();
  // one isolated miss: a call right after something invalidated the cache
  map.SetOccupancy(Eigen::Vector3i(1, 1, 1), 0);
  map.UpdateOccupancy(true);
  auto t6 = clk::now();
  acc += map.GetDistance(Eigen::Vector3d(12.0, 12.0, 12.0));
  auto t7 = clk::now();
  printf("{\"calls\": %d, \"walk_ns_per_call\": %.1f, \"walk_brick_fetches\": %lld, \"uniform_ns_per_call\": %.1f, "
         "\"uniform_brick_fetches\": %lld, \"batch_host_buffers_ns_per_query\": %.1f, \"first_call_after_an_update_us\": %.1f, "
         "\"checksum\": %.6f}\n",
         N, secs(t0, t1) / N * 1e9, (long long)(f1 - f0), secs(t2, t3) / N * 1e9, (long long)(f2 - f1), secs(t4, t5) / N * 1e9,
         secs(t6, t7) * 1e6, acc);
  return 0;
}
