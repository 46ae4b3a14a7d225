/*
 * StepFilter.hpp -- drop-in for traversabilityFilters/StepFilter on MI355X
 * (reference: traversability_estimation_filters/include/filters/StepFilter.hpp:21-64, src/StepFilter.cpp:20-186).
 */
#ifndef TRAVGPU_STEPFILTER_HPP
#define TRAVGPU_STEPFILTER_HPP

#include <filters/filter_base.h>
#include <string>

namespace filters {

template <typename T>
class StepFilter : public FilterBase<T> {
 public:
  StepFilter();
  virtual ~StepFilter();
  virtual bool configure();
  /*! Reads layer "elevation" of mapIn, adds layer map_type; the temporary step_height layer never leaves the device. */
  virtual bool update(const T& mapIn, T& mapOut);

 private:
  double criticalValue_;                            //! Maximum allowed step [m].
  double firstWindowRadius_, secondWindowRadius_;   //! Window radii [m].
  int nCellCritical_;                               //! Critical number of cells above the maximum step.
  std::string type_;                                //! Output layer name.
};

}  // namespace filters
#endif
