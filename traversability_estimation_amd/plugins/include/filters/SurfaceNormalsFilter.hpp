/*
 * SurfaceNormalsFilter.hpp -- traversabilityFilters/SurfaceNormalsFilter on MI355X.
 * The reference's README still lists a "Surface Normals Filter" (README.md:173) but ships none; its chain takes the
 * normals from gridMapFilters/NormalVectorsFilter (traversability_estimation/config/robot_filter_parameter.yaml:3-9).
 * This plugin computes the same layers on the device and reads the same parameter keys, so that entry of the YAML can
 * be pointed here by changing its `type:` line only.
 */
#ifndef TRAVGPU_SURFACENORMALSFILTER_HPP
#define TRAVGPU_SURFACENORMALSFILTER_HPP

#include <filters/filter_base.h>
#include <string>

namespace filters {

template <typename T>
class SurfaceNormalsFilter : public FilterBase<T> {
 public:
  SurfaceNormalsFilter();
  virtual ~SurfaceNormalsFilter();
  /*! Keys of NormalVectorsFilter: radius (required), normal_vector_positive_axis (x | y | z, default z),
   *  input_layer (default "elevation"), output_layers_prefix (default "surface_normal_"). */
  virtual bool configure();
  /*! Reads input_layer of mapIn, adds <prefix>x, <prefix>y, <prefix>z (NaN where the input is invalid). */
  virtual bool update(const T& mapIn, T& mapOut);

 private:
  double radius_;
  int axis_;
  int rankRule_;  // unit_z_for_planar_discs (TE_OPT_NORMALS_RANK_RULE)
  std::string inputLayer_, prefix_;
};

}  // namespace filters
#endif
