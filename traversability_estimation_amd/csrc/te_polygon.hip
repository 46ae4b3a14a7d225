// te_polygon.hip -- polygon footprints on the resident layers (SURVEY §8f N3 and the polygon half of N2).
//
// TraversabilityMap::isTraversable(polygon, traversability) (TraversabilityMap.cpp:586-645): walk the cells of the
// polygon's bounding box that lie inside the polygon (grid_map::PolygonIterator); the first cell that fails
// isTraversableForFilters (:774-792 -- the untraversable mask k_fp_mask leaves behind) makes the polygon
// untraversable, otherwise the result is the mean traversability of the cells (NaN counts as traversabilityDefault_).
//   k_polygon_footprint     traversabilityFootprint(footprintYaw) :239-305: for every cell the footprint polygon centred
//                           on it, as given (traversability_x) and turned by yaw (traversability_rot)
//   k_polygons_traversable  a batch of arbitrary polygons (the per-segment hulls of checkPolygonalFootprintPath :464-584)
// One thread owns one polygon and walks its bounding box in SubmapIterator order (row index outer), so the double sum is
// the reference's sum bit for bit; lanes of a wavefront own adjacent centre cells, so their reads of the traversability
// and mask layers coalesce.  Inside / outside is grid_map::Polygon::isInside's crossing-number expression evaluated in
// the same order in double: footprints whose edges pass through cell centres (0.45 m at 0.05 m resolution) are decided
// by its rounding, cell by cell.  The division is skipped when the cell is clearly left or right of the whole edge.
#include <algorithm>

#include "te_geom.h"
#include "te_internal.h"

namespace te {
namespace {

// boundPositionToRange (grid_map_core GridMapMath.cpp) for one axis
__device__ __forceinline__ double bound_axis(double position, double len, double mappos) {
  double shifted = position - mappos + 0.5 * len;
  double eps = 10.0 * 2.220446049250313e-16;
  if (fabs(position) > 1.0) eps *= fabs(position);
  if (shifted <= 0)
    shifted = eps;
  else if (shifted >= len)
    shifted = len - eps;
  return shifted + mappos - 0.5 * len;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// vert(k, x, y): vertex k of the polygon
template <class V>
__device__ __forceinline__ bool polygon_inside(int n, V&& vert, double px, double py) {
  int cross = 0;
  double xj, yj;
  vert(n - 1, xj, yj);
  for (int i = 0; i < n; ++i) {
    double xi, yi;
    vert(i, xi, yi);
    if ((yi > py) != (yj > py)) {
      const double lo = fmin(xi, xj), hi = fmax(xi, xj);
      const double margin = 1e-9 * (fabs(lo) + fabs(hi) + 1.0);  // >> the rounding error of the expression below
      if (px < lo - margin)
        ++cross;
      else if (px <= hi + margin && px < (xj - xi) * (py - yi) / (yj - yi) + xi)
        ++cross;
    }
    xj = xi;
    yj = yi;
  }
  return (cross & 1) != 0;
}

template <class V>
__device__ __forceinline__ bool polygon_traversable(const Geo& g, const uint8_t* __restrict__ untrav,
                                                    const float* __restrict__ trav, double def, int n, V&& vert,
                                                    double& value) {
  double tlx, tly;
  vert(0, tlx, tly);
  double brx = tlx, bry = tly;
  for (int k = 1; k < n; ++k) {
    double x, y;
    vert(k, x, y);
    tlx = tlx < x ? x : tlx;
    tly = tly < y ? y : tly;
    brx = x < brx ? x : brx;
    bry = y < bry ? y : bry;
  }
  tlx = bound_axis(tlx, g.len_x, g.pos_x);
  tly = bound_axis(tly, g.len_y, g.pos_y);
  brx = bound_axis(brx, g.len_x, g.pos_x);
  bry = bound_axis(bry, g.len_y, g.pos_y);
  int ti, tj, bi, bj;
  pos_to_index(g, tlx, tly, ti, tj);
  pos_to_index(g, brx, bry, bi, bj);
  ti = clampi(ti, 0, g.rows - 1);
  bi = clampi(bi, 0, g.rows - 1);
  tj = clampi(tj, 0, g.cols - 1);
  bj = clampi(bj, 0, g.cols - 1);
  unsigned ncells = 0;
  double t = 0.0;
  for (int a = ti; a <= bi; ++a) {
    const double px = cell_x(g, a);
    for (int b = tj; b <= bj; ++b) {
      if (!polygon_inside(n, vert, px, cell_y(g, b))) continue;
      const size_t o = (size_t)b * g.rows + a;
      if (untrav[o]) {  // :603-611
        value = 0.0;
        return false;
      }
      ++ncells;
      const float v = trav[o];
      t += (v == v && fabsf(v) != __builtin_inff()) ? (double)v : def;  // :613-618 isValid = finite
    }
  }
  if (ncells == 0) {  // :626-629
    value = def;
    return def != 0.0;
  }
  value = t / (double)ncells;
  return true;
}

__global__ __launch_bounds__(256) void k_polygon_footprint(Geo g, PolygonArgs a, const float* __restrict__ trav,
                                                           const uint8_t* __restrict__ untrav, float* __restrict__ out_x,
                                                           float* __restrict__ out_rot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i >= g.rows) return;
  const size_t map = (size_t)blockIdx.z * g.rows * g.cols;
  const double cx = cell_x(g, i), cy = cell_y(g, j);
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    double t;
    const bool ok = polygon_traversable(
        g, untrav + map, trav + map, a.def, a.n,
        [&](int k, double& x, double& y) {
          x = a.off[which][2 * k] + cx;  // toPosition * orientation * positionToVertex: linear * v, then + translation
          y = a.off[which][2 * k + 1] + cy;
        },
        t);
    (which ? out_rot : out_x)[map + (size_t)j * g.rows + i] = ok ? (float)t : 0.0f;  // :293-300
  }
}

__global__ __launch_bounds__(64) void k_polygons_traversable(Geo g, double def, int n_polygons,
                                                            const int* __restrict__ vertex_offset,
                                                            const double* __restrict__ vertex_xy, const float* __restrict__ trav,
                                                            const uint8_t* __restrict__ untrav,
                                                            unsigned char* __restrict__ is_traversable,
                                                            double* __restrict__ traversability) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_polygons) return;
  const double* v = vertex_xy + 2 * (size_t)vertex_offset[k];
  double t;
  const bool ok = polygon_traversable(
      g, untrav, trav, def, vertex_offset[k + 1] - vertex_offset[k],
      [&](int m, double& x, double& y) {
        x = v[2 * m];
        y = v[2 * m + 1];
      },
      t);
  is_traversable[k] = ok ? 1 : 0;
  traversability[k] = t;
}

}  // namespace

// The rotation part of  toPosition * orientation * positionToVertex  (TraversabilityMap.cpp:250-283) for a yaw-only
// orientation: kindr AngleAxis(yaw, 0, 0, 1) -> quaternion (cos(yaw/2), 0, 0, sin(yaw/2)), Eigen's toRotationMatrix,
// linear * v.  Host side, once per call.
void rotate_footprint(int n_points, const double* points_xy, double yaw, double* out_xy) {
  const double w = cos(yaw / 2.0), z = sin(yaw / 2.0);
  const double tz = 2.0 * z, twz = tz * w, tzz = tz * z;
  const double r00 = 1.0 - (0.0 + tzz), r01 = 0.0 - twz, r10 = 0.0 + twz, r11 = 1.0 - (0.0 + tzz);
  for (int k = 0; k < n_points; ++k) {
    const double px = points_xy[2 * k], py = points_xy[2 * k + 1];
    out_xy[2 * k] = r00 * px + r01 * py;
    out_xy[2 * k + 1] = r10 * px + r11 * py;
  }
}

// ---- checkPolygonalFootprintPath (TraversabilityMap.cpp:464-584), host part: the polygons of every path ----------------
namespace {

struct P2 {
  double x, y;
};

// grid_map::Polygon::monotoneChainConvexHullOfPoints (sortVertices: x then y; vectorsMakeClockwiseTurn: cross <= 0)
void convex_hull(const std::vector<P2>& points, std::vector<P2>& hull) {
  const size_t n = points.size();
  if (n <= 3) {
    hull = points;
    return;
  }
  std::vector<P2> sorted(points);
  std::sort(sorted.begin(), sorted.end(), [](const P2& a, const P2& b) { return a.x < b.x || (a.x == b.x && a.y < b.y); });
  auto clockwise = [](const P2& o, const P2& a, const P2& b) {
    const double ax = a.x - o.x, ay = a.y - o.y, bx = b.x - o.x, by = b.y - o.y;
    return ax * by - bx * ay <= 0.0;
  };
  hull.assign(2 * n, P2{0.0, 0.0});
  int k = 0;
  for (size_t i = 0; i < n; ++i) {
    while (k >= 2 && clockwise(hull[k - 2], hull[k - 1], sorted[i])) k--;
    hull[k++] = sorted[i];
  }
  for (int i = (int)n - 2, t = k + 1; i >= 0; i--) {
    while (k >= t && clockwise(hull[k - 2], hull[k - 1], sorted[i])) k--;
    hull[k++] = sorted[i];
  }
  hull.resize(k - 1);
}

double polygon_area(const std::vector<P2>& v) {  // Polygon::getArea
  double area = 0.0;
  size_t j = v.size() - 1;
  for (size_t i = 0; i < v.size(); i++) {
    area += (v[j].x + v[i].x) * (v[j].y - v[i].y);
    j = i;
  }
  return fabs(area / 2.0);
}

}  // namespace

void build_path_polygons(int n_paths, const int* pose_offset, const double* poses, int n_points, const double* points_xyz,
                         const unsigned char* conservative, PathPolygons& out) {
  out.vertex_offset.assign(1, 0);
  out.vertex_xy.clear();
  out.area.clear();
  out.area_previous.clear();
  out.first.assign(n_paths, 0);
  out.count.assign(n_paths, 0);
  out.status.assign(n_paths, 0);
  std::vector<P2> poly1, poly2, all, hull;
  auto emit = [&](const std::vector<P2>& poly, double area_prev) {
    for (const P2& v : poly) {
      out.vertex_xy.push_back(v.x);
      out.vertex_xy.push_back(v.y);
    }
    out.vertex_offset.push_back((int)(out.vertex_xy.size() / 2));
    out.area.push_back(polygon_area(poly));
    out.area_previous.push_back(area_prev);
  };
  for (int k = 0; k < n_paths; ++k) {
    const int n = pose_offset[k + 1] - pose_offset[k];
    out.first[k] = (int)out.area.size();
    if (n <= 0) {  // :330-334 "This path has no poses to check!"
      out.status[k] = 2;
      continue;
    }
    const bool cons = conservative && conservative[k];
    poly2.clear();
    double ex = 0.0, ey = 0.0;
    for (int i = 0; i < n; ++i) {
      const double* q = poses + 7 * (size_t)(pose_offset[k] + i);
      poly1 = poly2;  // :481
      const double sx = ex, sy = ey;
      ex = q[0];
      ey = q[1];
      // toPosition * orientation * positionToVertex (:497): Eigen Quaternion::toRotationMatrix, linear * v + translation
      const double x = q[3], y = q[4], z = q[5], w = q[6];
      const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
      const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y,
                   tzz = tz * z;
      const double r00 = 1.0 - (tyy + tzz), r01 = txy - twz, r02 = txz + twy;
      const double r10 = txy + twz, r11 = 1.0 - (txx + tzz), r12 = tyz - twx;
      poly2.clear();
      for (int m = 0; m < n_points; ++m) {
        const double px = points_xyz[3 * m], py = points_xyz[3 * m + 1], pz = points_xyz[3 * m + 2];
        poly2.push_back(P2{((r00 * px + r01 * py) + r02 * pz) + q[0], ((r10 * px + r11 * py) + r12 * pz) + q[1]});
      }
      if (cons && i > 0) {  // :512-522
        const double dx = ex - sx, dy = ey - sy;
        const size_t m1 = poly1.size(), m2 = poly2.size();
        for (size_t m = 0; m < m1; ++m) poly2.push_back(P2{poly1[m].x + dx, poly1[m].y + dy});
        for (size_t m = 0; m < m2; ++m) poly1.push_back(P2{poly2[m].x - dx, poly2[m].y - dy});
      }
      if (n == 1) emit(poly2, 0.0);  // :524-546
      if (n > 1 && i > 0) {          // :548-579
        all = poly1;
        all.insert(all.end(), poly2.begin(), poly2.end());
        convex_hull(all, hull);
        emit(hull, polygon_area(poly1));
      }
      if (cons && poly2.size() + (size_t)n_points > (size_t)kMaxPathPolygonVertices && i + 1 < n) {
        out.status[k] = 3;  // the conservative vertex lists grow with every pose: refuse the rest of the path
        break;
      }
    }
    out.count[k] = (int)out.area.size() - out.first[k];
  }
}

hipError_t launch_polygon_footprint(const Geo& g, const PolygonArgs& a, const float* trav, const uint8_t* untrav, float* out_x,
                                    float* out_rot, hipStream_t stream) {
  hipLaunchKernelGGL(k_polygon_footprint, dim3((unsigned)((g.rows + 255) / 256), (unsigned)g.cols, (unsigned)g.batch), dim3(256),
                     0, stream, g, a, trav, untrav, out_x, out_rot);
  return hipGetLastError();
}

hipError_t launch_polygons_traversable(const Geo& g, double def, int n_polygons, const int* vertex_offset, const double* vertex_xy,
                                       const float* trav, const uint8_t* untrav, unsigned char* is_traversable,
                                       double* traversability, hipStream_t stream) {
  if (n_polygons <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_polygons_traversable, dim3((unsigned)((n_polygons + 63) / 64)), dim3(64), 0, stream, g, def, n_polygons,
                     vertex_offset, vertex_xy, trav, untrav, is_traversable, traversability);
  return hipGetLastError();
}

}  // namespace te
