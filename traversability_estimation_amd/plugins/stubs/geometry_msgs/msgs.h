// stand-ins for the geometry_msgs structs the footprint-path API carries (plain data, real member names)
#pragma once
#include <string>
#include <vector>

namespace std_msgs {
struct Header {
  unsigned seq = 0;
  std::string frame_id;
};
}  // namespace std_msgs

namespace geometry_msgs {
struct Point {
  double x = 0, y = 0, z = 0;
};
struct Quaternion {
  double x = 0, y = 0, z = 0, w = 1;
};
struct Pose {
  Point position;
  Quaternion orientation;
};
struct PoseArray {
  std_msgs::Header header;
  std::vector<Pose> poses;
};
struct Point32 {
  float x = 0, y = 0, z = 0;
};
struct Polygon {
  std::vector<Point32> points;
};
struct PolygonStamped {
  std_msgs::Header header;
  Polygon polygon;
};
}  // namespace geometry_msgs
