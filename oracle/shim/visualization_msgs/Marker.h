// oracle/shim/visualization_msgs/Marker.h -- TEST INFRASTRUCTURE ONLY. POD stand-ins for the ROS
// message fields that the reference's visualisation getters write (src/ESDFMap.cpp:544-699).
#pragma once
#include <string>
#include <vector>
namespace std_msgs {
struct Header {
  std::string frame_id;
};
struct ColorRGBA {
  float r = 0, g = 0, b = 0, a = 0;
};
}  // namespace std_msgs
namespace geometry_msgs {
struct Point {
  double x = 0, y = 0, z = 0;
};
struct Point32 {
  float x = 0, y = 0, z = 0;
};
struct Quaternion {
  double x = 0, y = 0, z = 0, w = 0;
};
struct Vector3 {
  double x = 0, y = 0, z = 0;
};
struct Pose {
  Point position;
  Quaternion orientation;
};
}  // namespace geometry_msgs
namespace visualization_msgs {
struct Marker {
  enum { POINTS = 8, MODIFY = 0 };
  std_msgs::Header header;
  int id = 0, type = 0, action = 0;
  geometry_msgs::Vector3 scale;
  geometry_msgs::Pose pose;
  std::vector<geometry_msgs::Point> points;
  std::vector<std_msgs::ColorRGBA> colors;
};
}  // namespace visualization_msgs
