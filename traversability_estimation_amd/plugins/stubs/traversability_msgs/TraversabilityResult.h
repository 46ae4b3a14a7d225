// stand-in for traversability_msgs/TraversabilityResult (msg/TraversabilityResult.msg)
#pragma once

namespace traversability_msgs {
struct TraversabilityResult {
  unsigned char is_safe = 0;
  double traversability = 0.0;
  double area = 0.0;
};
}  // namespace traversability_msgs
