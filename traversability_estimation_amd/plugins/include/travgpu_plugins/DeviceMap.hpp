/*
 * DeviceMap.hpp -- host-side glue between a grid_map::GridMap and the C-ABI of libtravgpu.so
 * (include/travgpu.h).  One process-wide context, serialised by a mutex: the ROS node calls the
 * filter chain from AsyncSpinner threads without a lock of its own
 * (traversability_estimation/src/TraversabilityMap.cpp:203-214).
 */
#ifndef TRAVGPU_PLUGINS_DEVICEMAP_HPP
#define TRAVGPU_PLUGINS_DEVICEMAP_HPP

#include <mutex>
#include <string>

#include <grid_map_core/GridMap.hpp>

#include "travgpu.h"

namespace travgpu_plugins {

class DeviceMap {
 public:
  static DeviceMap& instance();
  std::mutex& mutex() { return mutex_; }
  // All calls below expect the caller to hold mutex().  They return false and fill error() on failure.
  bool prepare(const grid_map::GridMap& map);                       // (re)sets the geometry if it changed, notes the start index
  bool params(te_params& p);                                        // current parameter set (to edit and pass back)
  bool setParams(const te_params& p);
  bool upload(const grid_map::GridMap& map, const std::string& layer, int te_layer);
  bool runFilter(int filter);
  bool runChain(unsigned flags);
  bool download(grid_map::GridMap& map, const std::string& layer, int te_layer);
  const std::string& error() const { return error_; }

 private:
  DeviceMap();
  ~DeviceMap();
  bool check(int rc);
  std::mutex mutex_;
  te_ctx* ctx_;
  int rows_, cols_;
  int start_row_, start_col_;  // GridMap::getStartIndex() of the map last passed to prepare()
  double res_, px_, py_;
  std::string error_;
};

}  // namespace travgpu_plugins
#endif
