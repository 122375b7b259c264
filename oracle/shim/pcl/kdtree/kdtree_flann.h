// oracle/shim/pcl/kdtree/kdtree_flann.h -- TEST INFRASTRUCTURE ONLY. Brute-force stand-in for the
// PCL kd-tree that only the reference's debug checker CheckWithGroundTruth uses
// (src/ESDFMap.cpp:905-1054).
#pragma once
#include <limits>
#include <memory>
#include <vector>
namespace pcl {
struct PointXYZ {
  float x, y, z;
  PointXYZ() : x(0), y(0), z(0) {}
  PointXYZ(float a, float b, float c) : x(a), y(b), z(c) {}
};
template <typename T>
struct PointCloud {
  typedef std::shared_ptr<PointCloud<T>> Ptr;
  unsigned width = 0, height = 0;
  std::vector<T> points;
};
template <typename T>
struct KdTreeFLANN {
  typename PointCloud<T>::Ptr cloud;
  void setInputCloud(const typename PointCloud<T>::Ptr &c) { cloud = c; }
  int nearestKSearch(const T &p, int, std::vector<int> &idx, std::vector<float> &d2) const {
    float best = std::numeric_limits<float>::max();
    int bi = -1;
    for (std::size_t i = 0; i < cloud->points.size(); ++i) {
      const T &q = cloud->points[i];
      float dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z;
      float d = dx * dx + dy * dy + dz * dz;
      if (d < best) {
        best = d;
        bi = (int)i;
      }
    }
    idx[0] = bi;
    d2[0] = best;
    return bi >= 0;
  }
};
}  // namespace pcl
