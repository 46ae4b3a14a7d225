/*
 * FusedChainFilter.hpp -- NEW plugin (traversabilityFilters/FusedChainFilter): the whole default chain
 * of traversability_estimation/config/robot_filter_parameter.yaml:1-37 (normals -> slope -> step ->
 * roughness -> weighted sum -> deletion of the normals) in ONE update(): one upload of "elevation",
 * one device chain, four layers back.  For users who can edit the YAML; the three reference plugin
 * names keep working unchanged next to it.
 */
#ifndef TRAVGPU_FUSEDCHAINFILTER_HPP
#define TRAVGPU_FUSEDCHAINFILTER_HPP

#include <filters/filter_base.h>
#include <string>

#include "travgpu.h"

namespace filters {

template <typename T>
class FusedChainFilter : public FilterBase<T> {
 public:
  FusedChainFilter();
  virtual ~FusedChainFilter();
  /*! Parameters (all optional, defaults = the shipped YAML): normals_radius, slope_critical_value,
   *  step_critical_value, first_window_radius, second_window_radius, critical_cell_number,
   *  roughness_critical_value, estimation_radius, keep_surface_normals (int). */
  virtual bool configure();
  virtual bool update(const T& mapIn, T& mapOut);

 private:
  te_params params_;
  int keepNormals_;
  int rankRule_;  // unit_z_for_planar_discs (TE_OPT_NORMALS_RANK_RULE)
};

}  // namespace filters
#endif
