/* SurfaceNormalsFilter.cpp -- see include/filters/SurfaceNormalsFilter.hpp */
#include "filters/SurfaceNormalsFilter.hpp"

#include <mutex>

#include <pluginlib/class_list_macros.h>
#include <grid_map_ros/grid_map_ros.hpp>

#include "travgpu_plugins/DeviceMap.hpp"

using travgpu_plugins::DeviceMap;

namespace filters {

template <typename T>
SurfaceNormalsFilter<T>::SurfaceNormalsFilter() : radius_(0.05), axis_(2), rankRule_(0), inputLayer_("elevation"), prefix_("surface_normal_") {}

template <typename T>
SurfaceNormalsFilter<T>::~SurfaceNormalsFilter() {}

template <typename T>
bool SurfaceNormalsFilter<T>::configure() {
  if (!FilterBase<T>::getParam(std::string("radius"), radius_)) {
    ROS_ERROR("SurfaceNormalsFilter did not find param radius");
    return false;
  }
  if (radius_ < 0.0) {
    ROS_ERROR("SurfaceNormalsFilter: radius must not be negative");
    return false;
  }
  std::string axis("z");
  FilterBase<T>::getParam(std::string("normal_vector_positive_axis"), axis);
  if (axis == "x")
    axis_ = 0;
  else if (axis == "y")
    axis_ = 1;
  else if (axis == "z")
    axis_ = 2;
  else {
    ROS_ERROR("SurfaceNormalsFilter: normal_vector_positive_axis must be x, y or z");
    return false;
  }
  FilterBase<T>::getParam(std::string("input_layer"), inputLayer_);
  FilterBase<T>::getParam(std::string("output_layers_prefix"), prefix_);
  // optional: NormalVectorsFilter's degenerate-plane rule of grid_map <= 1.6 (UnitZ for an exactly planar disc; the filter
  // that wrote the reference's bag: travgpu.h TE_OPT_NORMALS_RANK_RULE); default: today's area method
  FilterBase<T>::getParam(std::string("unit_z_for_planar_discs"), rankRule_);
  return true;
}

template <typename T>
bool SurfaceNormalsFilter<T>::update(const T& mapIn, T& mapOut) {
  static const char kAxis[3] = {'x', 'y', 'z'};
  static const int kLayer[3] = {TE_LAYER_NORMAL_X, TE_LAYER_NORMAL_Y, TE_LAYER_NORMAL_Z};
  mapOut = mapIn;
  DeviceMap& dev = DeviceMap::instance();
  std::lock_guard<std::mutex> lock(dev.mutex());
  te_params p;
  bool ok = dev.prepare(mapOut) && dev.params(p);
  if (ok) {
    p.normals_radius = radius_;
    p.normals_axis = axis_;
    ok = dev.setParams(p) && dev.setOption(TE_OPT_NORMALS_RANK_RULE, rankRule_ ? 1 : 0) && dev.upload(mapOut, inputLayer_, TE_LAYER_ELEVATION) &&
         dev.runFilter(TE_FILTER_NORMALS);
  }
  for (int k = 0; ok && k < 3; ++k) {
    const std::string name = prefix_ + kAxis[k];
    mapOut.add(name);
    // what comes back is what the device layer holds: the plugins behind this one find it resident
    ok = dev.download(mapOut, name, kLayer[k]) && dev.noteResident(mapOut, name, kLayer[k]);
  }
  if (!ok) ROS_ERROR("SurfaceNormalsFilter (MI355X): %s", dev.error().c_str());
  return ok;
}

}  // namespace filters

PLUGINLIB_EXPORT_CLASS(filters::SurfaceNormalsFilter<grid_map::GridMap>, filters::FilterBase<grid_map::GridMap>)
