/*
 * TraversabilityMap.hpp -- device-backed counterpart of traversability_estimation::TraversabilityMap
 * (traversability_estimation/include/traversability_estimation/TraversabilityMap.hpp) for the part of it that is the
 * hot path: filter chain, footprint layers and footprint-path checks.  Same method names, argument meaning and return
 * conventions; the map stays on the device between calls (its own te_ctx), layers come back only through
 * getTraversabilityMap().  What a maintainer swaps in for the members of the ROS node that do the arithmetic
 * (TraversabilityMap.cpp:156-237, 239-318, 320-645); node handle, publishers, services and tf stay where they are.
 */
#ifndef TRAVERSABILITY_ESTIMATION_GPU_TRAVERSABILITYMAP_HPP
#define TRAVERSABILITY_ESTIMATION_GPU_TRAVERSABILITYMAP_HPP

#include <mutex>
#include <string>
#include <vector>

#include <grid_map_core/GridMap.hpp>
#include <traversability_msgs/FootprintPath.h>
#include <traversability_msgs/TraversabilityResult.h>

#include "travgpu.h"

namespace traversability_estimation_gpu {

class TraversabilityMap {
 public:
  explicit TraversabilityMap(int device = 0);
  ~TraversabilityMap();
  TraversabilityMap(const TraversabilityMap&) = delete;
  TraversabilityMap& operator=(const TraversabilityMap&) = delete;

  /*! Filter and footprint parameters (the YAML of the reference: robot_filter_parameter.yaml,
   *  robot_footprint_parameter.yaml, robot.yaml); false + error() if the reference's configure() would refuse them. */
  bool setParameters(const te_params& params);
  const te_params& getParameters() const { return params_; }
  /*! footprint/footprint_polygon (TraversabilityMap.cpp:91-103). */
  void setFootprintPolygon(const std::vector<geometry_msgs::Point32>& points) { footprintPoints_ = points; }

  /*! footprint/check_robot_inclination (:114): checkFootprintPath then runs checkInclination (:748-762) on the layer
   *  "robot_slope" of the elevation map handed to setElevationMap (the path checks fail if it had none). */
  bool setCheckRobotInclination(bool enabled);
  /*! setElevationMap (:135-154): needs the layer "elevation"; any start index.  A layer "robot_slope" goes along. */
  bool setElevationMap(const grid_map::GridMap& elevationMap);
  /*! computeTraversability (:202-237): the filter chain; false if no elevation map has been set. */
  bool computeTraversability();
  /*! traversabilityFootprint(radius, offset) (:307-318): layer traversability_footprint. */
  bool traversabilityFootprint(const double& radius, const double& offset);
  /*! traversabilityFootprint(footprintYaw) (:239-305): layers traversability_x / traversability_rot. */
  bool traversabilityFootprint(double footprintYaw);
  /*! checkFootprintPath (:320-342): circular footprint when path.footprint has no points, else polygonal.  Returns
   *  false only for a path without poses; an uninitialised map gives is_safe = false and true, like the reference. */
  bool checkFootprintPath(const traversability_msgs::FootprintPath& path, traversability_msgs::TraversabilityResult& result);
  /*! The CheckFootprintPath service body (TraversabilityEstimation.cpp:276-291) for all paths of a request at once:
   *  one device launch per footprint kind.  Returns false if any path has no poses (its result stays unsafe). */
  bool checkFootprintPaths(const std::vector<traversability_msgs::FootprintPath>& paths,
                           std::vector<traversability_msgs::TraversabilityResult>& results);
  /*! The elevation map's geometry with the layers computed so far (getTraversabilityMap :196-199). */
  grid_map::GridMap getTraversabilityMap();
  bool traversabilityMapInitialized() const { return traversabilityMapInitialized_; }
  const std::string& error() const { return error_; }

 private:
  bool check(int rc);
  bool ensureCircularFootprint(double radius);
  mutable std::mutex mutex_;
  te_ctx* ctx_;
  te_params params_;
  std::vector<geometry_msgs::Point32> footprintPoints_;
  grid_map::GridMap geometry_;  // geometry and start index of the last elevation map (no layers)
  bool elevationMapInitialized_, traversabilityMapInitialized_, footprintLayer_, polygonLayers_;
  bool checkRobotInclination_, robotSlopeLayer_;
  double footprintRadius_, footprintOffset_;
  double circularFootprintOffset_;  // :348 "TODO: get this with FootprintPath msg" = 0.15
  std::string error_;
};

}  // namespace traversability_estimation_gpu
#endif
