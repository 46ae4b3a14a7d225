/*
 * SlopeFilter.hpp -- drop-in for traversabilityFilters/SlopeFilter on MI355X.
 * Same class name, template, parameters, defaults and validation as the reference
 * (traversability_estimation_filters/include/filters/SlopeFilter.hpp:22-59, src/SlopeFilter.cpp:20-93);
 * update() runs on the GPU through libtravgpu.so (te_run_filter(TE_FILTER_SLOPE)).
 */
#ifndef TRAVGPU_SLOPEFILTER_HPP
#define TRAVGPU_SLOPEFILTER_HPP

#include <filters/filter_base.h>
#include <string>

namespace filters {

template <typename T>
class SlopeFilter : public FilterBase<T> {
 public:
  SlopeFilter();
  virtual ~SlopeFilter();
  virtual bool configure();
  /*! Reads layer "surface_normal_z" of mapIn, adds layer map_type (1 = flat, 0 = critical slope, NaN = unknown). */
  virtual bool update(const T& mapIn, T& mapOut);

 private:
  double criticalValue_;  //! Maximum allowed slope [rad], in [0, pi/2].
  std::string type_;      //! Output layer name.
};

}  // namespace filters
#endif
