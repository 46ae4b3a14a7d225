/* FusedChainFilter.cpp -- see include/filters/FusedChainFilter.hpp */
#include "filters/FusedChainFilter.hpp"

#include <mutex>

#include <pluginlib/class_list_macros.h>
#include <grid_map_ros/grid_map_ros.hpp>

#include "travgpu_plugins/DeviceMap.hpp"

using travgpu_plugins::DeviceMap;

namespace filters {

template <typename T>
FusedChainFilter<T>::FusedChainFilter() : keepNormals_(0), rankRule_(0) {
  te_params_default(&params_);
}

template <typename T>
FusedChainFilter<T>::~FusedChainFilter() {}

template <typename T>
bool FusedChainFilter<T>::configure() {
  te_params_default(&params_);
  FilterBase<T>::getParam(std::string("normals_radius"), params_.normals_radius);
  FilterBase<T>::getParam(std::string("slope_critical_value"), params_.slope_critical);
  FilterBase<T>::getParam(std::string("step_critical_value"), params_.step_critical);
  FilterBase<T>::getParam(std::string("first_window_radius"), params_.step_radius1);
  FilterBase<T>::getParam(std::string("second_window_radius"), params_.step_radius2);
  FilterBase<T>::getParam(std::string("critical_cell_number"), params_.step_ncrit);
  FilterBase<T>::getParam(std::string("roughness_critical_value"), params_.rough_critical);
  FilterBase<T>::getParam(std::string("estimation_radius"), params_.rough_radius);
  FilterBase<T>::getParam(std::string("keep_surface_normals"), keepNormals_);
  FilterBase<T>::getParam(std::string("unit_z_for_planar_discs"), rankRule_);  // TE_OPT_NORMALS_RANK_RULE (see SurfaceNormalsFilter)
  if (te_params_validate(&params_) != TE_OK) {
    ROS_ERROR("%s", te_last_error());
    return false;
  }
  return true;
}

template <typename T>
bool FusedChainFilter<T>::update(const T& mapIn, T& mapOut) {
  mapOut = mapIn;
  static const char* kOut[4] = {"traversability_slope", "traversability_step", "traversability_roughness", "traversability"};
  static const int kLayer[4] = {TE_LAYER_SLOPE, TE_LAYER_STEP, TE_LAYER_ROUGHNESS, TE_LAYER_TRAVERSABILITY};
  static const char* kN[3] = {"surface_normal_x", "surface_normal_y", "surface_normal_z"};
  static const int kNL[3] = {TE_LAYER_NORMAL_X, TE_LAYER_NORMAL_Y, TE_LAYER_NORMAL_Z};
  DeviceMap& dev = DeviceMap::instance();
  std::lock_guard<std::mutex> lock(dev.mutex());
  bool ok = dev.prepare(mapOut) && dev.setParams(params_) && dev.setOption(TE_OPT_NORMALS_RANK_RULE, rankRule_ ? 1 : 0) &&
            dev.upload(mapOut, "elevation", TE_LAYER_ELEVATION) &&
            dev.runChain(keepNormals_ ? TE_RUN_KEEP_NORMALS : 0u);
  for (int k = 0; ok && k < 4; ++k) {
    mapOut.add(kOut[k]);
    ok = dev.download(mapOut, kOut[k], kLayer[k]);
  }
  for (int k = 0; ok && keepNormals_ && k < 3; ++k) {
    mapOut.add(kN[k]);
    ok = dev.download(mapOut, kN[k], kNL[k]);
  }
  if (!ok) ROS_ERROR("FusedChainFilter (MI355X): %s", dev.error().c_str());
  return ok;
}

}  // namespace filters

PLUGINLIB_EXPORT_CLASS(filters::FusedChainFilter<grid_map::GridMap>, filters::FilterBase<grid_map::GridMap>)
