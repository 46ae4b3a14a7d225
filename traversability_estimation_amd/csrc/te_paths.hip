// te_paths.hip -- batched TraversabilityMap::checkFootprintPath for circular footprints
// (traversability_estimation/src/TraversabilityMap.cpp:320-342 -> checkCircularFootprintPath :344-462) on the
// resident traversability_footprint layer.  With the layer complete, isTraversable(center, ...) takes its memo
// branch for every centre (:672-677: value = layer, traversable = value != 0), so a path is a walk over
// LineIterator cells (every fourth, nSkip :396) with a running mean and a length-weighted mean over the
// segments.  One thread per path: planners check thousands of short candidate paths per cycle.
// footprint/check_robot_inclination (:114, robot_footprint_parameter.yaml:10): with the layer robot_slope given,
// checkInclination (:748-762) runs before every pose / segment (:366-370, :390-394); the same test batched on its
// own is k_check_inclination (the polygonal path check uses it, :526-528, :553-557).
// Not covered (reference defaults): publishPolygons, compute_untraversable_polygon.
#include "te_geom.h"
#include "te_internal.h"

namespace te {

namespace {

// TraversabilityMap::checkInclination(start, end) :748-762.  outside: a position off the map -- atPosition throws
// there, and the segment branch ignores getIndex()'s failure (undefined indices); reported as status 1.
__device__ __forceinline__ bool inclination_ok(const Geo& g, const float* __restrict__ robot_slope, double sx, double sy,
                                               double ex, double ey, bool& outside) {
  int si, sj, ei, ej;
  outside = false;
  if (ex == sx && ey == sy) {  // :750-751
    if (!pos_inside(g, sx, sy) || !pos_to_index(g, sx, sy, si, sj)) {
      outside = true;
      return false;
    }
    return !((double)robot_slope[(size_t)sj * g.rows + si] == 0.0);
  }
  if (!pos_to_index(g, sx, sy, si, sj) || !pos_to_index(g, ex, ey, ei, ej)) {
    outside = true;
    return false;
  }
  LineIt L;
  for (L.init(si, sj, ei, ej); !L.past_end(); L.next()) {  // from the start index to the end index :756
    const float v = robot_slope[(size_t)L.j * g.rows + L.i];
    if (!isfinite(v)) continue;  // isValid :757
    if ((double)v == 0.0) return false;
  }
  return true;
}

__global__ __launch_bounds__(256) void k_check_inclination(Geo g, const float* __restrict__ robot_slope, int n,
                                                           const double* __restrict__ start_end_xy,
                                                           unsigned char* __restrict__ ok, int* __restrict__ status) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const double* q = start_end_xy + 4 * (size_t)k;
  bool outside;
  ok[k] = inclination_ok(g, robot_slope, q[0], q[1], q[2], q[3], outside) ? 1 : 0;
  status[k] = outside ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_check_circular_paths(Geo g, const float* __restrict__ footprint, double fp_default,
                                                              const float* __restrict__ robot_slope, int n_paths, const int* __restrict__ pose_offset,
                                                              const double* __restrict__ pose_xy,
                                                              unsigned char* __restrict__ is_safe,
                                                              double* __restrict__ traversability, int* __restrict__ status) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_paths) return;
  const int p0 = pose_offset[k], n = pose_offset[k + 1] - p0;
  const double* xy = pose_xy + 2 * (size_t)p0;
  unsigned char safe = 0;
  double out = 0.0;
  int st = 0;
  if (n <= 0) {  // :330-334
    st = 2;
  } else {
    double res_trav = 0.0, length_path = 0.0, ex = 0.0, ey = 0.0;
    bool ok = true;
    for (int i = 0; i < n && ok; ++i) {
      const double sx = ex, sy = ey;
      ex = xy[2 * i];
      ey = xy[2 * i + 1];
      if (robot_slope && (n == 1 || i > 0)) {  // checkRobotInclination_ :366-370, :390-394
        bool outside;
        const bool good = n == 1 ? inclination_ok(g, robot_slope, ex, ey, ex, ey, outside)
                                 : inclination_ok(g, robot_slope, sx, sy, ex, ey, outside);
        if (!good) {
          st = outside ? 1 : 0;
          ok = false;
          break;
        }
      }
      if (n == 1) {  // :365-385
        double t = fp_default;
        if (pos_inside(g, ex, ey)) {  // :663-665 otherwise
          int ci, cj;
          pos_to_index(g, ex, ey, ci, cj);
          t = (double)footprint[(size_t)cj * g.rows + ci];
        }
        if (!(t != 0.0)) {
          ok = false;
          break;
        }
        res_trav = t;
      }
      if (n > 1 && i > 0) {  // :388-456
        int si, sj, ei, ej;
        if (!pos_to_index(g, sx, sy, si, sj) || !pos_to_index(g, ex, ey, ei, ej)) {
          st = 1;  // the reference ignores getIndex()'s result here: undefined indices
          ok = false;
          break;
        }
        double sum = 0.0;
        int nline = 0;
        LineIt L;
        for (L.init(ei, ej, si, sj); !L.past_end(); L.next()) {  // from the end index to the start index
          const double t = (double)footprint[(size_t)L.j * g.rows + L.i];
          if (!(t != 0.0)) {
            ok = false;
            break;
          }
          sum += t;
          nline++;
          for (int s = 0; s < 3; ++s)  // nSkip :396
            if (!L.past_end()) L.next();
        }
        if (!ok) break;
        const double t = sum / (double)nline;
        const double dx = ex - sx, dy = ey - sy;
        const double length_segment = sqrt(dx * dx + dy * dy);
        if (i > 1) {  // :443-447
          const double length_previous = length_path;
          length_path += length_segment;
          res_trav = (length_segment * t + length_previous * res_trav) / length_path;
        } else {
          length_path = length_segment;
          res_trav = t;
        }
      }
    }
    if (ok) {
      safe = 1;
      out = res_trav;
    }
  }
  is_safe[k] = safe;
  traversability[k] = out;
  status[k] = st;
}

}  // namespace

hipError_t launch_check_circular_paths(const Geo& g, const float* footprint, double fp_default, const float* robot_slope,
                                       int n_paths, const int* pose_offset, const double* pose_xy, unsigned char* is_safe,
                                       double* traversability, int* status, hipStream_t stream) {
  if (n_paths <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_check_circular_paths, dim3((unsigned)((n_paths + 255) / 256)), dim3(256), 0, stream, g, footprint,
                     fp_default, robot_slope, n_paths, pose_offset, pose_xy, is_safe, traversability, status);
  return hipGetLastError();
}

hipError_t launch_check_inclination(const Geo& g, const float* robot_slope, int n, const double* start_end_xy,
                                    unsigned char* ok, int* status, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_check_inclination, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, g, robot_slope, n,
                     start_end_xy, ok, status);
  return hipGetLastError();
}

}  // namespace te
