// te_polygon.hip -- polygon footprints on the resident layers (SURVEY §8f N3 and the polygon half of N2).
//
// TraversabilityMap::isTraversable(polygon, traversability) (TraversabilityMap.cpp:586-645): walk the cells of the
// polygon's bounding box that lie inside the polygon (grid_map::PolygonIterator); the first cell that fails
// isTraversableForFilters (:774-792 -- the untraversable mask k_fp_mask leaves behind) makes the polygon
// untraversable, otherwise the result is the mean traversability of the cells (NaN counts as traversabilityDefault_).
//   k_polygon_footprint     traversabilityFootprint(footprintYaw) :239-305: for every cell the footprint polygon centred
//                           on it, as given (traversability_x) and turned by yaw (traversability_rot)
//   k_polygons_traversable  a batch of arbitrary polygons (the per-segment hulls of checkPolygonalFootprintPath :464-584),
//                           one wavefront each
// In k_polygon_footprint one thread owns one polygon and walks its bounding box in SubmapIterator order (row index outer),
// so the double sum is the reference's sum bit for bit; lanes of a wavefront own adjacent centre cells, so their reads of
// the traversability and mask layers coalesce.  Inside / outside is grid_map::Polygon::isInside's crossing-number expression evaluated in
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

// one edge (xi, yi) - (xj, yj) of grid_map::Polygon::isInside's crossing-number test
__device__ __forceinline__ bool edge_crosses(double xi, double yi, double xj, double yj, double px, double py) {
  if ((yi > py) == (yj > py)) return false;
  if (xi == xj) return px < xi;  // the expression below is 0 * (..) / (..) + xi == xi exactly (yj != yi here)
  const double lo = fmin(xi, xj), hi = fmax(xi, xj);
  const double margin = 1e-9 * (fabs(lo) + fabs(hi) + 1.0);  // >> the rounding error of the expression below
  if (px < lo - margin) return true;
  return px <= hi + margin && px < (xj - xi) * (py - yi) / (yj - yi) + xi;
}

// vert(k, x, y): vertex k of the polygon
template <class V>
__device__ __forceinline__ bool polygon_inside(int n, V&& vert, double px, double py) {
  int cross = 0;
  double xj, yj;
  vert(n - 1, xj, yj);
  for (int i = 0; i < n; ++i) {
    double xi, yi;
    vert(i, xi, yi);
    cross += edge_crosses(xi, yi, xj, yj, px, py);
    xj = xi;
    yj = yi;
  }
  return (cross & 1) != 0;
}

// The usual footprint has four vertices: held in registers, a cell evaluation does not wait for per-edge loads.
struct Quad {
  double x[4], y[4];
  template <class V>
  __device__ __forceinline__ void load(V&& vert) {
    vert(0, x[0], y[0]);
    vert(1, x[1], y[1]);
    vert(2, x[2], y[2]);
    vert(3, x[3], y[3]);
  }
  __device__ __forceinline__ bool inside(double px, double py) const {  // edges (0,3) (1,0) (2,1) (3,2) like the loop
    const int cross = (int)edge_crosses(x[0], y[0], x[3], y[3], px, py) + (int)edge_crosses(x[1], y[1], x[0], y[0], px, py) +
                      (int)edge_crosses(x[2], y[2], x[1], y[1], px, py) + (int)edge_crosses(x[3], y[3], x[2], y[2], px, py);
    return (cross & 1) != 0;
  }
};

// PolygonIterator::findSubmapParameters: the index range [ti, bi] x [tj, bj] the iterator walks
template <class V>
__device__ __forceinline__ void polygon_bbox(const Geo& g, int n, V&& vert, int& ti, int& bi, int& tj, int& bj) {
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
  pos_to_index(g, tlx, tly, ti, tj);
  pos_to_index(g, brx, bry, bi, bj);
  ti = clampi(ti, 0, g.rows - 1);
  bi = clampi(bi, 0, g.rows - 1);
  tj = clampi(tj, 0, g.cols - 1);
  bj = clampi(bj, 0, g.cols - 1);
}

template <class V>
__device__ __forceinline__ bool polygon_traversable(const Geo& g, const uint8_t* __restrict__ untrav,
                                                    const float* __restrict__ trav, double def, int n, V&& vert,
                                                    double& value) {
  int ti, tj, bi, bj;
  polygon_bbox(g, n, vert, ti, bi, tj, bj);
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

// ---- the same layers through an offset table ------------------------------------------------------------------------
// The footprint polygon of cell c is the footprint translated to c, so whether cell q lies inside depends, up to
// rounding, only on the index offset q - c.  The host classifies every offset of the bounding box once
// (build_polygon_table): "inside for every centre cell" (farther than a tolerance from every edge), "outside for every
// centre cell" (dropped), or "decided by rounding" -- only the last kind is still evaluated per cell with the exact
// expression.  A workgroup (256 adjacent centre cells of one map column) stages the cells its polygons can touch in LDS
// as doubles ready to add (0 outside the map, NaN on an untraversable cell, traversabilityDefault_ for an invalid one),
// then every lane walks the table in the reference's order: one LDS read and one add per inside cell.  Same cells, same
// order, same values as k_polygon_footprint: the results are identical.
__device__ __forceinline__ unsigned uniform_u32(const unsigned* p) { return __builtin_amdgcn_readfirstlane(*p); }

__global__ __launch_bounds__(256) void k_polygon_footprint_table(Geo g, PolygonArgs a, PolygonTables tabs,
                                                                 const unsigned* __restrict__ stream,
                                                                 const float* __restrict__ trav,
                                                                 const uint8_t* __restrict__ untrav, float* __restrict__ out_x,
                                                                 float* __restrict__ out_rot) {
  extern __shared__ double tile[];
  const int tid = threadIdx.x, i0 = blockIdx.x * 256, i = i0 + tid, j = blockIdx.y;
  const size_t map = (size_t)blockIdx.z * g.rows * g.cols;
  const double cx = cell_x(g, i), cy = cell_y(g, j);
  const double nan = __builtin_nan("");
#pragma unroll 1
  for (int which = 0; which < 2; ++which) {
    const PolygonTable tb = tabs.t[which];
    const int W = 255 + tb.di_span;
    __syncthreads();  // the previous polygon's tile is no longer read
    for (int idx = tid; idx < W * tb.dj_span; idx += 256) {
      const int c = idx / W, r = idx - c * W;
      const int aa = i0 + tb.di_min + r, bb = j + tb.dj_min + c;
      double w = 0.0;
      if (aa >= 0 && aa < g.rows && bb >= 0 && bb < g.cols) {
        const size_t o = map + (size_t)bb * g.rows + aa;
        const float v = trav[o];
        w = untrav[o] ? nan : ((v == v && fabsf(v) != __builtin_inff()) ? (double)v : a.def);
      }
      tile[idx] = w;
    }
    __syncthreads();
    if (i >= g.rows) continue;
    auto vert = [&](int k, double& x, double& y) {
      x = a.off[which][2 * k] + cx;
      y = a.off[which][2 * k + 1] + cy;
    };
    int ti = 0, bi = -1, tj = 0, bj = -1;
    Quad quad = {};
    if (tb.n_uncertain) {
      polygon_bbox(g, a.n, vert, ti, bi, tj, bj);
      if (a.n == 4) quad.load(vert);
    }
    const unsigned* p = stream + tb.first;
    double t = 0.0;
    int ncells = 0;
#pragma unroll 1
    for (int r = 0; r < tb.n_rows; ++r) {
      const unsigned hdr = uniform_u32(p++);
      const int di_idx = hdr & 0xff, items = hdr >> 8;
      const int aa = i + tb.di_min + di_idx;
      const bool a_in = aa >= 0 && aa < g.rows;
      const double* col0 = tile + tid + di_idx;
#pragma unroll 1
      for (int k = 0; k < items; ++k) {
        const unsigned item = uniform_u32(p++);
        const int dj_idx = item & 0xff, len = (item >> 8) & 0xff;
        const int b0 = j + tb.dj_min + dj_idx;
        const double* q = col0 + dj_idx * W;
        if (!(item >> 31)) {
          // inside for every centre cell; the columns of the run that are on the map (uniform)
          const int lo = b0 < 0 ? 0 : b0, hi = b0 + len - 1 > g.cols - 1 ? g.cols - 1 : b0 + len - 1;
          ncells += (a_in && hi >= lo) ? hi - lo + 1 : 0;
#pragma unroll 4
          for (int m = 0; m < len; ++m) t += q[m * W];
        } else if (a_in && b0 >= 0 && b0 < g.cols && aa >= ti && aa <= bi && b0 >= tj && b0 <= bj &&
                   (a.n == 4 ? quad.inside(cell_x(g, aa), cell_y(g, b0))
                             : polygon_inside(a.n, vert, cell_x(g, aa), cell_y(g, b0)))) {
          t += q[0];
          ++ncells;
        }
      }
    }
    float res;
    if (t != t)
      res = 0.0f;  // touched an untraversable cell (:603-611, :295/:299)
    else if (ncells == 0)
      res = a.def != 0.0 ? (float)a.def : 0.0f;  // :626-629
    else
      res = (float)(t / (double)ncells);
    (which ? out_rot : out_x)[map + (size_t)j * g.rows + i] = res;
  }
}

// A batch of arbitrary polygons (the per-segment hulls of a path check are a few thousand cells each): one wavefront per
// polygon.  The lanes take 64 consecutive cells of the bounding-box walk at a time; the values are then added in lane
// order (every lane carries the same running sum), so the double sum is still the reference's sum bit for bit -- cells
// outside the polygon contribute +0.0, which changes nothing.  An untraversable cell anywhere ends the polygon (the
// reference stops at the first one in walk order; the result is 0 either way).
__global__ __launch_bounds__(256) void k_polygons_traversable(Geo g, double def, int n_polygons,
                                                             const int* __restrict__ vertex_offset,
                                                             const double* __restrict__ vertex_xy, const float* __restrict__ trav,
                                                             const uint8_t* __restrict__ untrav,
                                                             unsigned char* __restrict__ is_traversable,
                                                             double* __restrict__ traversability) {
  const int lane = threadIdx.x & 63;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= n_polygons) return;  // the whole wavefront
  const double* v = vertex_xy + 2 * (size_t)vertex_offset[k];
  const int n = vertex_offset[k + 1] - vertex_offset[k];
  auto vert = [&](int m, double& x, double& y) {
    x = v[2 * m];
    y = v[2 * m + 1];
  };
  int ti, bi, tj, bj;
  polygon_bbox(g, n, vert, ti, bi, tj, bj);
  const unsigned long long H = (unsigned long long)(bj - tj + 1), total = (unsigned long long)(bi - ti + 1) * H;
  double t = 0.0;
  unsigned ncells = 0;
  bool dead = false;
  for (unsigned long long base = 0; base < total; base += 64) {
    const unsigned long long idx = base + lane;
    double add = 0.0;
    bool inside = false, bad = false;
    if (idx < total) {
      const int a = ti + (int)(idx / H), b = tj + (int)(idx % H);  // SubmapIterator order: row index outer
      if (polygon_inside(n, vert, cell_x(g, a), cell_y(g, b))) {
        const size_t o = (size_t)b * g.rows + a;
        if (untrav[o]) {  // :603-611
          bad = true;
        } else {
          inside = true;
          const float w = trav[o];
          add = (w == w && fabsf(w) != __builtin_inff()) ? (double)w : def;  // :613-618
        }
      }
    }
    if (__ballot(bad) != 0ull) {
      dead = true;
      break;
    }
    ncells += (unsigned)__popcll(__ballot(inside));
#pragma unroll 8
    for (int l = 0; l < 64; ++l) t += __shfl(add, l);
  }
  if (lane != 0) return;
  if (dead) {
    is_traversable[k] = 0;
    traversability[k] = 0.0;
  } else if (ncells == 0) {  // :626-629
    is_traversable[k] = def != 0.0 ? 1 : 0;
    traversability[k] = def;
  } else {
    is_traversable[k] = 1;
    traversability[k] = t / (double)ncells;
  }
}

// isTraversable(polygon, computeUntraversablePolygon = true, ..) :592-645: the untraversable polygon is the convex hull of the
// positions of the polygon's untraversable cells.  Cells of one row index share x, so only a row's first and last such
// cell can be hull vertices (the cells between are popped by monotoneChainConvexHullOfPoints with a cross product of
// exactly 0): one wavefront per row of the bounding box reduces the row to (x, y of the first, y of the last, y of the
// one between when there are three, count); the host runs the chain over those.  rows5[5 * i ..]: row index i.
__global__ __launch_bounds__(256) void k_polygon_untraversable_rows(Geo g, int n, const double* __restrict__ v,
                                                                   const uint8_t* __restrict__ untrav,
                                                                   double* __restrict__ rows5) {
  const int lane = threadIdx.x & 63;
  const int a = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (a >= g.rows) return;  // the whole wavefront
  auto vert = [&](int m, double& x, double& y) {
    x = v[2 * m];
    y = v[2 * m + 1];
  };
  int ti, bi, tj, bj;
  polygon_bbox(g, n, vert, ti, bi, tj, bj);
  int lo = 0x7fffffff, hi = -1, cnt = 0, sum = 0;
  if (a >= ti && a <= bi) {
    const double px = cell_x(g, a);
    for (int b = tj + lane; b <= bj; b += 64) {
      if (!untrav[(size_t)b * g.rows + a]) continue;
      if (!polygon_inside(n, vert, px, cell_y(g, b))) continue;
      lo = b < lo ? b : lo;
      hi = b > hi ? b : hi;
      cnt++;
      sum += b;
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const int l2 = __shfl_xor(lo, d), h2 = __shfl_xor(hi, d);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
    cnt += __shfl_xor(cnt, d);
    sum += __shfl_xor(sum, d);
  }
  if (lane != 0) return;
  double* o = rows5 + 5 * (size_t)a;
  o[4] = (double)cnt;
  if (cnt > 0) {
    o[0] = cell_x(g, a);
    o[1] = cell_y(g, lo);
    o[2] = cell_y(g, hi);
    o[3] = cell_y(g, cnt == 3 ? sum - lo - hi : lo);
  }
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
// the chain proper (sort, lower hull, upper hull) for n >= 2 points
void monotone_chain(const std::vector<P2>& points, std::vector<P2>& hull) {
  const size_t n = points.size();
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

void convex_hull(const std::vector<P2>& points, std::vector<P2>& hull) {
  if (points.size() <= 3) {
    hull = points;
    return;
  }
  monotone_chain(points, hull);
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

// Classify the index offsets (di, dj) of one footprint polygon (vertex offsets `off` from the centre cell).  The cell at
// offset (di, dj) lies at (-di, -dj) * res from the centre.  Returns false when the polygon does not fit the table
// format (spans beyond 255 offsets, LDS tile too large): the caller then uses the per-cell kernel.
bool build_polygon_table(const Geo& g, int n, const double* off, std::vector<unsigned>& stream, PolygonTable& tb) {
  double xmin = off[0], xmax = off[0], ymin = off[1], ymax = off[1], ext = 0.0;
  for (int k = 0; k < n; ++k) {
    xmin = std::min(xmin, off[2 * k]);
    xmax = std::max(xmax, off[2 * k]);
    ymin = std::min(ymin, off[2 * k + 1]);
    ymax = std::max(ymax, off[2 * k + 1]);
    ext = std::max(ext, std::max(fabs(off[2 * k]), fabs(off[2 * k + 1])));
  }
  // rounding of the per-cell evaluation is ~1e-16 of the coordinates involved; anything closer than `tol` to an edge
  // is left to that evaluation
  const double scale = fabs(g.pos_x) + fabs(g.pos_y) + g.len_x + g.len_y + ext;
  const double tol = 1e-7 * g.res + 1e-9 * scale;
  if (!(ext / g.res < 1e6)) return false;  // far beyond any table; also keeps the casts below defined
  const int di_lo = (int)floor(-xmax / g.res) - 1, di_hi = (int)ceil(-xmin / g.res) + 1;
  const int dj_lo = (int)floor(-ymax / g.res) - 1, dj_hi = (int)ceil(-ymin / g.res) + 1;
  if (di_hi - di_lo + 1 > 255 || dj_hi - dj_lo + 1 > 255) return false;
  auto classify = [&](int di, int dj) {  // 0 outside, 1 inside, 2 decided by rounding
    const double px = -(double)di * g.res, py = -(double)dj * g.res;
    int cross = 0;
    double dmin = 1e300;
    for (int i = 0, j = n - 1; i < n; j = i++) {
      const double xi = off[2 * i], yi = off[2 * i + 1], xj = off[2 * j], yj = off[2 * j + 1];
      if (((yi > py) != (yj > py)) && (px < (xj - xi) * (py - yi) / (yj - yi) + xi)) cross++;
      const double ex = xj - xi, ey = yj - yi, l2 = ex * ex + ey * ey;
      double u = l2 > 0.0 ? ((px - xi) * ex + (py - yi) * ey) / l2 : 0.0;
      u = u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);
      const double qx = xi + u * ex - px, qy = yi + u * ey - py;
      dmin = std::min(dmin, sqrt(qx * qx + qy * qy));
    }
    if (!(dmin > tol)) return 2;
    return cross & 1;
  };
  // the used part of the box
  int a0 = di_hi + 1, a1 = di_lo - 1, b0 = dj_hi + 1, b1 = dj_lo - 1;
  std::vector<signed char> cls((size_t)(di_hi - di_lo + 1) * (dj_hi - dj_lo + 1));
  for (int di = di_lo; di <= di_hi; ++di)
    for (int dj = dj_lo; dj <= dj_hi; ++dj) {
      const int c = classify(di, dj);
      cls[(size_t)(di - di_lo) * (dj_hi - dj_lo + 1) + (dj - dj_lo)] = (signed char)c;
      if (c) {
        a0 = std::min(a0, di);
        a1 = std::max(a1, di);
        b0 = std::min(b0, dj);
        b1 = std::max(b1, dj);
      }
    }
  tb.first = (int)stream.size();
  tb.n_rows = 0;
  tb.n_uncertain = 0;
  if (a1 < a0) {  // covers no cell centre
    tb.di_min = tb.dj_min = 0;
    tb.di_span = tb.dj_span = 1;
    return true;
  }
  tb.di_min = a0;
  tb.dj_min = b0;
  tb.di_span = a1 - a0 + 1;
  tb.dj_span = b1 - b0 + 1;
  if ((size_t)(255 + tb.di_span) * tb.dj_span * sizeof(double) > 64 * 1024) return false;
  for (int di = a0; di <= a1; ++di) {  // SubmapIterator order: row index outer, column index inner
    const size_t hdr = stream.size();
    stream.push_back(0);
    unsigned items = 0;
    for (int dj = b0; dj <= b1;) {
      const int c = cls[(size_t)(di - di_lo) * (dj_hi - dj_lo + 1) + (dj - dj_lo)];
      if (c == 0) {
        ++dj;
        continue;
      }
      int len = 1;
      if (c == 1)
        while (dj + len <= b1 && len < 255 && cls[(size_t)(di - di_lo) * (dj_hi - dj_lo + 1) + (dj + len - dj_lo)] == 1) ++len;
      else
        ++tb.n_uncertain;
      stream.push_back((unsigned)(dj - b0) | ((unsigned)len << 8) | (c == 2 ? 0x80000000u : 0u));
      ++items;
      dj += len;
    }
    if (items == 0) {
      stream.pop_back();
      continue;
    }
    stream[hdr] = (unsigned)(di - a0) | (items << 8);
    ++tb.n_rows;
  }
  return true;
}

hipError_t launch_polygon_footprint_table(const Geo& g, const PolygonArgs& a, const PolygonTables& tabs, const unsigned* d_stream,
                                          const float* trav, const uint8_t* untrav, float* out_x, float* out_rot,
                                          hipStream_t stream) {
  size_t lds = 0;
  for (int w = 0; w < 2; ++w) lds = std::max(lds, (size_t)(255 + tabs.t[w].di_span) * tabs.t[w].dj_span * sizeof(double));
  hipLaunchKernelGGL(k_polygon_footprint_table, dim3((unsigned)((g.rows + 255) / 256), (unsigned)g.cols, (unsigned)g.batch),
                     dim3(256), lds, stream, g, a, tabs, d_stream, trav, untrav, out_x, out_rot);
  return hipGetLastError();
}

hipError_t launch_polygon_footprint(const Geo& g, const PolygonArgs& a, const float* trav, const uint8_t* untrav, float* out_x,
                                    float* out_rot, hipStream_t stream) {
  hipLaunchKernelGGL(k_polygon_footprint, dim3((unsigned)((g.rows + 255) / 256), (unsigned)g.cols, (unsigned)g.batch), dim3(256),
                     0, stream, g, a, trav, untrav, out_x, out_rot);
  return hipGetLastError();
}

hipError_t launch_polygon_untraversable_rows(const Geo& g, int n, const double* vertex_xy, const uint8_t* untrav, double* rows5,
                                             hipStream_t stream) {
  hipLaunchKernelGGL(k_polygon_untraversable_rows, dim3((unsigned)((g.rows + 3) / 4)), dim3(256), 0, stream, g, n, vertex_xy,
                     untrav, rows5);
  return hipGetLastError();
}

// host half: the hull over what k_polygon_untraversable_rows left (rows5, g.rows entries)
void untraversable_hull_from_rows(int rows, const double* rows5, std::vector<double>& hull_xy) {
  std::vector<P2> pts, hull;
  long total = 0;
  for (int a = 0; a < rows; ++a) total += (long)rows5[5 * (size_t)a + 4];
  for (int a = 0; a < rows; ++a) {  // PolygonIterator order: row index outer, column index ascending
    const double* r = rows5 + 5 * (size_t)a;
    const int cnt = (int)r[4];
    if (cnt <= 0) continue;
    pts.push_back(P2{r[0], r[1]});
    if (cnt == 3 && total <= 3) pts.push_back(P2{r[0], r[3]});
    if (cnt >= 2) pts.push_back(P2{r[0], r[2]});
  }
  if (total <= 3)
    hull = pts;  // monotoneChainConvexHullOfPoints' "points.size() <= 3": Polygon(points) as collected
  else
    monotone_chain(pts, hull);  // more than three cells, even when they sit in one or two rows: sorted, collinear ones dropped
  hull_xy.clear();
  for (const P2& q : hull) {
    hull_xy.push_back(q.x);
    hull_xy.push_back(q.y);
  }
}

hipError_t launch_polygons_traversable(const Geo& g, double def, int n_polygons, const int* vertex_offset, const double* vertex_xy,
                                       const float* trav, const uint8_t* untrav, unsigned char* is_traversable,
                                       double* traversability, hipStream_t stream) {
  if (n_polygons <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_polygons_traversable, dim3((unsigned)((n_polygons + 3) / 4)), dim3(256), 0, stream, g, def, n_polygons,
                     vertex_offset, vertex_xy, trav, untrav, is_traversable, traversability);
  return hipGetLastError();
}

}  // namespace te
