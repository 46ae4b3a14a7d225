#include "traversability_estimation_gpu/TraversabilityMap.hpp"

#include <cmath>
#include <map>

#include <ros/ros.h>

namespace traversability_estimation_gpu {

TraversabilityMap::TraversabilityMap(int device)
    : ctx_(nullptr),
      elevationMapInitialized_(false),
      traversabilityMapInitialized_(false),
      footprintLayer_(false),
      polygonLayers_(false),
      checkRobotInclination_(false),
      robotSlopeLayer_(false),
      footprintRadius_(-1.0),
      footprintOffset_(-1.0),
      circularFootprintOffset_(0.15) {
  te_params_default(&params_);
  if (check(te_create(device, &ctx_))) check(te_set_params(ctx_, &params_));
}

TraversabilityMap::~TraversabilityMap() {
  if (ctx_) te_destroy(ctx_);
}

bool TraversabilityMap::check(int rc) {
  if (rc == TE_OK) return true;
  error_ = te_last_error();
  ROS_ERROR("TraversabilityMap (MI355X): %s", error_.c_str());
  return false;
}

bool TraversabilityMap::setParameters(const te_params& params) {
  std::lock_guard<std::mutex> lock(mutex_);
  if (!ctx_ || !check(te_set_params(ctx_, &params))) return false;
  params_ = params;
  traversabilityMapInitialized_ = false;  // conservative: computeTraversability() has to run again
  footprintLayer_ = polygonLayers_ = false;
  return true;
}

bool TraversabilityMap::setCheckRobotInclination(bool enabled) {
  std::lock_guard<std::mutex> lock(mutex_);
  if (!ctx_ || !check(te_set_check_robot_inclination(ctx_, enabled ? 1 : 0))) return false;
  checkRobotInclination_ = enabled;
  return true;
}

bool TraversabilityMap::setElevationMap(const grid_map::GridMap& elevationMap) {
  std::lock_guard<std::mutex> lock(mutex_);
  if (!ctx_) return false;
  if (!elevationMap.exists("elevation")) {  // :145-150
    ROS_WARN("Traversability Map: Can't set elevation map because there is no layer %s.", "elevation");
    return false;
  }
  const int rows = elevationMap.getSize()(0), cols = elevationMap.getSize()(1);
  if (!check(te_set_geometry(ctx_, rows, cols, 1, elevationMap.getResolution(), elevationMap.getPosition().x(),
                             elevationMap.getPosition().y())))
    return false;
  const auto start = elevationMap.getStartIndex();
  if (!check(te_upload_layer_circular(ctx_, TE_LAYER_ELEVATION, elevationMap.get("elevation").data(), 0, start(0), start(1))))
    return false;
  // robot_slope (robotSlopeType_ :47) is nowhere computed by the reference: checkInclination reads it off whatever map
  // the node was handed, so it travels with the elevation map when it is there
  robotSlopeLayer_ = elevationMap.exists("robot_slope");
  if (robotSlopeLayer_) {
    if (!check(te_upload_layer_circular(ctx_, TE_LAYER_ROBOT_SLOPE, elevationMap.get("robot_slope").data(), 0, start(0), start(1))))
      return false;
  } else {
    // a map without the layer: the device must not keep checking inclinations against the previous map's layer (the
    // reference's atPosition("robot_slope") throws for a map that lacks it)
    (void)te_set_layer_present(ctx_, TE_LAYER_ROBOT_SLOPE, 0);
  }
  geometry_ = grid_map::GridMap();
  geometry_.setGeometry(elevationMap.getLength(), elevationMap.getResolution(), elevationMap.getPosition());
  geometry_.setStartIndex(start);
  elevationMapInitialized_ = true;
  traversabilityMapInitialized_ = false;
  footprintLayer_ = polygonLayers_ = false;
  return true;
}

bool TraversabilityMap::computeTraversability() {
  std::lock_guard<std::mutex> lock(mutex_);
  if (!elevationMapInitialized_) {  // :228-231
    ROS_ERROR("Traversability Estimation: Elevation map is not initialized!");
    return false;
  }
  if (!check(te_run_chain(ctx_, 0))) {  // :214-218
    ROS_ERROR("Traversability Estimation: Could not update the filter chain! No traversability computed!");
    traversabilityMapInitialized_ = false;
    return false;
  }
  traversabilityMapInitialized_ = true;
  footprintLayer_ = polygonLayers_ = false;
  return true;
}

// the circular footprint pass for `radius` (+ the fixed offset of checkCircularFootprintPath); also what marks the
// untraversable cells for the polygon queries
bool TraversabilityMap::ensureCircularFootprint(double radius) {
  if (footprintLayer_ && footprintRadius_ == radius && footprintOffset_ == circularFootprintOffset_) return true;
  te_params p = params_;
  p.fp_radius = radius;
  p.fp_offset = circularFootprintOffset_;
  if (!check(te_set_params(ctx_, &p)) || !check(te_run_footprint(ctx_))) return false;
  params_ = p;
  footprintLayer_ = true;
  footprintRadius_ = radius;
  footprintOffset_ = circularFootprintOffset_;
  return true;
}

bool TraversabilityMap::traversabilityFootprint(const double& radius, const double& offset) {
  std::lock_guard<std::mutex> lock(mutex_);
  if (!traversabilityMapInitialized_) return false;  // :308
  te_params p = params_;
  p.fp_radius = radius;
  p.fp_offset = offset;
  if (!check(te_set_params(ctx_, &p)) || !check(te_run_footprint(ctx_))) return false;
  params_ = p;
  footprintLayer_ = true;
  footprintRadius_ = radius;
  footprintOffset_ = offset;
  return true;
}

bool TraversabilityMap::traversabilityFootprint(double footprintYaw) {
  std::lock_guard<std::mutex> lock(mutex_);
  if (!traversabilityMapInitialized_) return false;  // :240
  if (footprintPoints_.empty()) {
    error_ = "no footprint polygon set (footprint/footprint_polygon)";
    ROS_ERROR("TraversabilityMap (MI355X): %s", error_.c_str());
    return false;
  }
  if (!footprintLayer_ && !ensureCircularFootprint(params_.fp_radius)) return false;  // marks the untraversable cells
  std::vector<double> xy;
  for (const geometry_msgs::Point32& pt : footprintPoints_) {  // :273-276 float -> double
    xy.push_back(pt.x);
    xy.push_back(pt.y);
  }
  if (!check(te_run_polygon_footprint(ctx_, (int)footprintPoints_.size(), xy.data(), footprintYaw))) return false;
  polygonLayers_ = true;
  return true;
}

bool TraversabilityMap::checkFootprintPath(const traversability_msgs::FootprintPath& path,
                                           traversability_msgs::TraversabilityResult& result) {
  std::vector<traversability_msgs::TraversabilityResult> results;
  const bool ok = checkFootprintPaths(std::vector<traversability_msgs::FootprintPath>(1, path), results);
  if (!results.empty())
    result = results[0];
  else
    result.is_safe = static_cast<unsigned char>(false);
  return ok;
}

bool TraversabilityMap::checkFootprintPaths(const std::vector<traversability_msgs::FootprintPath>& paths,
                                            std::vector<traversability_msgs::TraversabilityResult>& results) {
  std::lock_guard<std::mutex> lock(mutex_);
  results.clear();
  if (paths.empty()) {
    ROS_WARN("No footprint path available to check!");  // TraversabilityEstimation.cpp:281-284
    return false;
  }
  // the service stops at the first path without poses (:330-334 -> TraversabilityEstimation.cpp:290)
  size_t n = paths.size();
  bool complete = true;
  for (size_t k = 0; k < paths.size(); ++k)
    if (paths[k].poses.poses.empty()) {
      ROS_WARN("Traversability Estimation: This path has no poses to check!");
      n = k;
      complete = false;
      break;
    }
  results.assign(n, traversability_msgs::TraversabilityResult());
  if (!traversabilityMapInitialized_) {  // :323-327
    ROS_WARN("Traversability Estimation: check Footprint path: Traversability map not yet initialized.");
    return complete;
  }
  // polygonal footprints: one batch per distinct footprint; circular ones: one batch per distinct radius
  std::map<double, std::vector<size_t>> circular;
  std::vector<size_t> polygonal;
  for (size_t k = 0; k < n; ++k) {
    if (paths[k].footprint.polygon.points.empty())
      circular[paths[k].radius].push_back(k);
    else
      polygonal.push_back(k);
  }
  std::vector<bool> done(n, false);
  for (size_t a = 0; a < polygonal.size(); ++a) {
    const size_t lead = polygonal[a];
    if (done[lead]) continue;
    const std::vector<geometry_msgs::Point32>& pts = paths[lead].footprint.polygon.points;
    std::vector<size_t> group;
    for (size_t b = a; b < polygonal.size(); ++b) {
      const std::vector<geometry_msgs::Point32>& q = paths[polygonal[b]].footprint.polygon.points;
      bool same = !done[polygonal[b]] && q.size() == pts.size();
      for (size_t m = 0; same && m < q.size(); ++m) same = q[m].x == pts[m].x && q[m].y == pts[m].y && q[m].z == pts[m].z;
      if (same) group.push_back(polygonal[b]);
    }
    if (!footprintLayer_ && !ensureCircularFootprint(params_.fp_radius)) return false;
    std::vector<int> offset(1, 0);
    std::vector<double> poses, points;
    std::vector<unsigned char> conservative;
    for (const geometry_msgs::Point32& pt : pts) {  // :502-504 float -> double
      points.push_back(pt.x);
      points.push_back(pt.y);
      points.push_back(pt.z);
    }
    for (size_t k : group) {
      for (const geometry_msgs::Pose& pose : paths[k].poses.poses) {
        const double v[7] = {pose.position.x,    pose.position.y,    pose.position.z,   pose.orientation.x,
                             pose.orientation.y, pose.orientation.z, pose.orientation.w};
        poses.insert(poses.end(), v, v + 7);
      }
      offset.push_back((int)(poses.size() / 7));
      conservative.push_back(paths[k].conservative);
      done[k] = true;
    }
    std::vector<unsigned char> safe(group.size());
    std::vector<double> trav(group.size()), area(group.size());
    std::vector<int> status(group.size());
    if (!check(te_check_polygon_footprint_paths(ctx_, 0, (int)group.size(), offset.data(), poses.data(), (int)pts.size(),
                                                points.data(), conservative.data(), safe.data(), trav.data(), area.data(),
                                                status.data())))
      return false;
    for (size_t m = 0; m < group.size(); ++m) {
      results[group[m]].is_safe = safe[m];
      results[group[m]].traversability = trav[m];
      results[group[m]].area = area[m];
    }
  }
  for (const auto& entry : circular) {
    if (!ensureCircularFootprint(entry.first)) return false;
    const std::vector<size_t>& group = entry.second;
    std::vector<int> offset(1, 0);
    std::vector<double> xy;
    for (size_t k : group) {
      for (const geometry_msgs::Pose& pose : paths[k].poses.poses) {
        xy.push_back(pose.position.x);
        xy.push_back(pose.position.y);
      }
      offset.push_back((int)(xy.size() / 2));
    }
    std::vector<unsigned char> safe(group.size());
    std::vector<double> trav(group.size());
    std::vector<int> status(group.size());
    if (!check(te_check_footprint_paths(ctx_, 0, (int)group.size(), offset.data(), xy.data(), safe.data(), trav.data(),
                                        status.data())))
      return false;
    for (size_t m = 0; m < group.size(); ++m) {
      results[group[m]].is_safe = safe[m];
      results[group[m]].traversability = trav[m];
      results[group[m]].area = 0.0;  // :355: never set for circular footprints
    }
  }
  return complete;
}

grid_map::GridMap TraversabilityMap::getTraversabilityMap() {
  std::lock_guard<std::mutex> lock(mutex_);
  grid_map::GridMap map = geometry_;
  if (!elevationMapInitialized_) return map;
  struct Entry {
    const char* name;
    int layer;
    bool present;
  };
  const Entry entries[] = {{"elevation", TE_LAYER_ELEVATION, true},
                           {"traversability_slope", TE_LAYER_SLOPE, traversabilityMapInitialized_},
                           {"traversability_step", TE_LAYER_STEP, traversabilityMapInitialized_},
                           {"traversability_roughness", TE_LAYER_ROUGHNESS, traversabilityMapInitialized_},
                           {"traversability", TE_LAYER_TRAVERSABILITY, traversabilityMapInitialized_},
                           {"traversability_footprint", TE_LAYER_FOOTPRINT, footprintLayer_},
                           {"traversability_x", TE_LAYER_TRAVERSABILITY_X, polygonLayers_},
                           {"traversability_rot", TE_LAYER_TRAVERSABILITY_ROT, polygonLayers_}};
  const auto start = map.getStartIndex();
  for (const Entry& e : entries) {
    if (!e.present) continue;
    map.add(e.name);
    if (!check(te_download_layer_circular(ctx_, e.layer, map.get(e.name).data(), 0, start(0), start(1)))) map.erase(e.name);
  }
  return map;
}

}  // namespace traversability_estimation_gpu
