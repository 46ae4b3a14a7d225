/*
 * RoughnessFilter.hpp -- drop-in for traversabilityFilters/RoughnessFilter on MI355X
 * (reference: traversability_estimation_filters/include/filters/RoughnessFilter.hpp:20-60, src/RoughnessFilter.cpp:20-136).
 */
#ifndef TRAVGPU_ROUGHNESSFILTER_HPP
#define TRAVGPU_ROUGHNESSFILTER_HPP

#include <filters/filter_base.h>
#include <string>

namespace filters {

template <typename T>
class RoughnessFilter : public FilterBase<T> {
 public:
  RoughnessFilter();
  virtual ~RoughnessFilter();
  virtual bool configure();
  /*! Reads layers "elevation" and "surface_normal_{x,y,z}" of mapIn, adds layer map_type. */
  virtual bool update(const T& mapIn, T& mapOut);

 private:
  double criticalValue_;     //! Maximum allowed roughness [m].
  double estimationRadius_;  //! Radius of the submap for the roughness estimation [m].
  std::string type_;         //! Output layer name.
};

}  // namespace filters
#endif
