// oracle/shim/sensor_msgs/PointCloud.h -- TEST INFRASTRUCTURE ONLY.
#pragma once
#include <visualization_msgs/Marker.h>
namespace sensor_msgs {
struct PointCloud {
  std_msgs::Header header;
  std::vector<geometry_msgs::Point32> points;
};
}  // namespace sensor_msgs
