// te_msg.h -- the wire formats either side of the filter chain (host code, no device work):
//   * the ROS1 serialisation of grid_map_msgs/GridMap, which is what elevation_mapping hands to the reference node
//     (TraversabilityEstimation.cpp:248-270 requestElevationMap, GridMapRosConverter::fromMessage / toMessage)
//   * rosbag V2.0 files holding such messages (GridMapRosConverter::loadFromBag / saveToBag,
//     TraversabilityEstimation.cpp:125-152, 318-329)
// Views point into the caller's buffer: a layer's float payload is handed to the device copy as it lies in the
// message (column-major, circular start index and all), so nothing is reshuffled on the host.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "travgpu.h"

namespace te {
namespace msg {

struct LayerView {
  const char* name;  // not NUL-terminated
  uint32_t name_len;
  size_t data_off;  // byte offset of the rows*cols float32 payload in the message
};

struct View {
  te_msg_info info;
  std::vector<LayerView> layers;
  std::vector<LayerView> basic_layers;  // names only
};

bool parse(const uint8_t* p, size_t n, View& v, std::string& err);

struct Names {
  int n;
  const char* const* v;
};
// bytes of a message with these layers (rows x cols each)
size_t message_size(const te_msg_info& info, Names layers, Names basic);
// everything but the float payloads; payload_off[k] = where layer k's rows*cols floats go
bool write_skeleton(const te_msg_info& info, Names layers, Names basic, uint8_t* out, size_t cap, std::vector<size_t>& payload_off,
                    std::string& err);

// last message of type grid_map_msgs/GridMap on `topic` (loadFromBag keeps the last one it instantiates)
bool bag_find(const uint8_t* bag, size_t n, const char* topic, size_t& off, size_t& len, std::string& err);
size_t bag_size(size_t msg_len, const char* topic);
bool bag_write(const uint8_t* message, size_t msg_len, const char* topic, uint32_t sec, uint32_t nsec, uint8_t* out, size_t cap,
               size_t& written, std::string& err);

}  // namespace msg
}  // namespace te
