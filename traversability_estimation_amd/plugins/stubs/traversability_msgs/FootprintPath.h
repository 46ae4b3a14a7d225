// stand-in for traversability_msgs/FootprintPath (msg/FootprintPath.msg)
#pragma once
#include <geometry_msgs/msgs.h>

namespace traversability_msgs {
struct FootprintPath {
  geometry_msgs::PoseArray poses;
  double radius = 0.0;
  geometry_msgs::PolygonStamped footprint;
  unsigned char conservative = 0;
  unsigned char compute_untraversable_polygon = 0;
};
}  // namespace traversability_msgs
