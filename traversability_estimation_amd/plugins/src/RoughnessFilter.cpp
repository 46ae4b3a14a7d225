/* RoughnessFilter.cpp -- see include/filters/RoughnessFilter.hpp */
#include "filters/RoughnessFilter.hpp"

#include <mutex>

#include <pluginlib/class_list_macros.h>
#include <grid_map_ros/grid_map_ros.hpp>

#include "travgpu_plugins/DeviceMap.hpp"

using travgpu_plugins::DeviceMap;

namespace filters {

template <typename T>
RoughnessFilter<T>::RoughnessFilter() : criticalValue_(0.3), estimationRadius_(0.3), type_("traversability_roughness") {}

template <typename T>
RoughnessFilter<T>::~RoughnessFilter() {}

template <typename T>
bool RoughnessFilter<T>::configure() {
  if (!FilterBase<T>::getParam(std::string("critical_value"), criticalValue_)) {
    ROS_ERROR("RoughnessFilter did not find param critical_value");
    return false;
  }
  if (criticalValue_ < 0.0) {
    ROS_ERROR("Critical roughness must be greater than zero");
    return false;
  }
  if (!FilterBase<T>::getParam(std::string("estimation_radius"), estimationRadius_)) {
    ROS_ERROR("RoughnessFilter did not find param estimation_radius");
    return false;
  }
  if (estimationRadius_ < 0.0) {
    ROS_ERROR("Roughness estimation radius must be greater than zero");
    return false;
  }
  if (!FilterBase<T>::getParam(std::string("map_type"), type_)) {
    ROS_ERROR("RoughnessFilter did not find param map_type");
    return false;
  }
  return true;
}

template <typename T>
bool RoughnessFilter<T>::update(const T& mapIn, T& mapOut) {
  mapOut = mapIn;
  mapOut.add(type_);
  DeviceMap& dev = DeviceMap::instance();
  std::lock_guard<std::mutex> lock(dev.mutex());
  te_params p;
  bool ok = dev.prepare(mapOut) && dev.params(p);
  if (ok) {
    p.rough_critical = criticalValue_;
    p.rough_radius = estimationRadius_;
    ok = dev.setParams(p) && dev.upload(mapOut, "elevation", TE_LAYER_ELEVATION) &&
         dev.upload(mapOut, "surface_normal_x", TE_LAYER_NORMAL_X) &&
         dev.upload(mapOut, "surface_normal_y", TE_LAYER_NORMAL_Y) &&
         dev.upload(mapOut, "surface_normal_z", TE_LAYER_NORMAL_Z) && dev.runFilter(TE_FILTER_ROUGHNESS) &&
         dev.download(mapOut, type_, TE_LAYER_ROUGHNESS);
  }
  if (!ok) ROS_ERROR("RoughnessFilter (MI355X): %s", dev.error().c_str());
  return ok;
}

}  // namespace filters

PLUGINLIB_EXPORT_CLASS(filters::RoughnessFilter<grid_map::GridMap>, filters::FilterBase<grid_map::GridMap>)
