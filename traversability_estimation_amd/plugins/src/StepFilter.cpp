/* StepFilter.cpp -- see include/filters/StepFilter.hpp */
#include "filters/StepFilter.hpp"

#include <mutex>

#include <pluginlib/class_list_macros.h>
#include <grid_map_ros/grid_map_ros.hpp>

#include "travgpu_plugins/DeviceMap.hpp"

using travgpu_plugins::DeviceMap;

namespace filters {

template <typename T>
StepFilter<T>::StepFilter()
    : criticalValue_(0.3), firstWindowRadius_(0.08), secondWindowRadius_(0.08), nCellCritical_(5), type_("traversability_step") {}

template <typename T>
StepFilter<T>::~StepFilter() {}

template <typename T>
bool StepFilter<T>::configure() {
  if (!FilterBase<T>::getParam(std::string("critical_value"), criticalValue_)) {
    ROS_ERROR("Step filter did not find param critical_value.");
    return false;
  }
  if (criticalValue_ < 0.0) {
    ROS_ERROR("Critical step height must be greater than zero.");
    return false;
  }
  if (!FilterBase<T>::getParam(std::string("first_window_radius"), firstWindowRadius_)) {
    ROS_ERROR("Step filter did not find param 'first_window_radius'.");
    return false;
  }
  if (firstWindowRadius_ < 0.0) {
    ROS_ERROR("'first_window_radius' must be greater than zero.");
    return false;
  }
  if (!FilterBase<T>::getParam(std::string("second_window_radius"), secondWindowRadius_)) {
    ROS_ERROR("Step filter did not find param 'second_window_radius'.");
    return false;
  }
  if (secondWindowRadius_ < 0.0) {
    ROS_ERROR("'second_window_radius' must be greater than zero.");
    return false;
  }
  if (!FilterBase<T>::getParam(std::string("critical_cell_number"), nCellCritical_)) {
    ROS_ERROR("Step filter did not find param 'critical_cell_number'.");
    return false;
  }
  if (nCellCritical_ <= 0) {
    ROS_ERROR("'critical_cell_number' must be greater than zero.");
    return false;
  }
  if (!FilterBase<T>::getParam(std::string("map_type"), type_)) {
    ROS_ERROR("Step filter did not find param map_type.");
    return false;
  }
  return true;
}

template <typename T>
bool StepFilter<T>::update(const T& mapIn, T& mapOut) {
  mapOut = mapIn;
  mapOut.add(type_);
  DeviceMap& dev = DeviceMap::instance();
  std::lock_guard<std::mutex> lock(dev.mutex());
  te_params p;
  bool ok = dev.prepare(mapOut) && dev.params(p);
  if (ok) {
    p.step_critical = criticalValue_;
    p.step_radius1 = firstWindowRadius_;
    p.step_radius2 = secondWindowRadius_;
    p.step_ncrit = nCellCritical_;
    // (the normals RoughnessFilter reads next start their way to the device beside this plugin's kernels and its download)
    ok = dev.setParams(p) && dev.upload(mapOut, "elevation", TE_LAYER_ELEVATION) &&
         dev.prefetch(mapOut, {{"surface_normal_x", TE_LAYER_NORMAL_X}, {"surface_normal_y", TE_LAYER_NORMAL_Y}, {"surface_normal_z", TE_LAYER_NORMAL_Z}}) &&
         dev.runFilter(TE_FILTER_STEP) && dev.download(mapOut, type_, TE_LAYER_STEP);
    ok = dev.finishPrefetch() && ok;
  }
  if (!ok) ROS_ERROR("StepFilter (MI355X): %s", dev.error().c_str());
  return ok;
}

}  // namespace filters

PLUGINLIB_EXPORT_CLASS(filters::StepFilter<grid_map::GridMap>, filters::FilterBase<grid_map::GridMap>)
