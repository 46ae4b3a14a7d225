/* SlopeFilter.cpp -- see include/filters/SlopeFilter.hpp */
#include "filters/SlopeFilter.hpp"

#include <cmath>
#include <mutex>

#include <pluginlib/class_list_macros.h>
#include <grid_map_ros/grid_map_ros.hpp>

#include "travgpu_plugins/DeviceMap.hpp"

using travgpu_plugins::DeviceMap;

namespace filters {

template <typename T>
SlopeFilter<T>::SlopeFilter() : criticalValue_(M_PI_4), type_("traversability_slope") {}

template <typename T>
SlopeFilter<T>::~SlopeFilter() {}

template <typename T>
bool SlopeFilter<T>::configure() {
  if (!FilterBase<T>::getParam(std::string("critical_value"), criticalValue_)) {
    ROS_ERROR("SlopeFilter did not find param critical_value");
    return false;
  }
  if (criticalValue_ > M_PI_2 || criticalValue_ < 0.0) {
    ROS_ERROR("Critical slope must be in the interval [0, PI/2]");
    return false;
  }
  if (!FilterBase<T>::getParam(std::string("map_type"), type_)) {
    ROS_ERROR("SlopeFilter did not find param map_type");
    return false;
  }
  return true;
}

template <typename T>
bool SlopeFilter<T>::update(const T& mapIn, T& mapOut) {
  mapOut = mapIn;
  mapOut.add(type_);
  DeviceMap& dev = DeviceMap::instance();
  std::lock_guard<std::mutex> lock(dev.mutex());
  te_params p;
  bool ok = dev.prepare(mapOut) && dev.params(p);
  if (ok) {
    p.slope_critical = criticalValue_;
    // (the layers StepFilter and RoughnessFilter read next -- robot_filter_parameter.yaml:10-28 -- start their way to the
    // device beside this plugin's kernel and the download of its output)
    ok = dev.setParams(p) && dev.upload(mapOut, "surface_normal_z", TE_LAYER_NORMAL_Z) &&
         dev.prefetch(mapOut, {{"elevation", TE_LAYER_ELEVATION}}) && dev.runFilter(TE_FILTER_SLOPE) &&
         dev.download(mapOut, type_, TE_LAYER_SLOPE);
    ok = dev.finishPrefetch() && ok;
  }
  if (!ok) ROS_ERROR("SlopeFilter (MI355X): %s", dev.error().c_str());
  return ok;
}

}  // namespace filters

PLUGINLIB_EXPORT_CLASS(filters::SlopeFilter<grid_map::GridMap>, filters::FilterBase<grid_map::GridMap>)
