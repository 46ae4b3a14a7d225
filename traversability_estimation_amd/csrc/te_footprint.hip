// te_footprint.hip -- the circular footprint pass on gfx950.
//
//   TraversabilityMap::traversabilityFootprint(radius, offset)   traversability_estimation/src/TraversabilityMap.cpp:307-318
//     -> isTraversable(center, radiusMax, traversability, radiusMin)              :654-746
//     -> isTraversableForFilters / checkForSlope / checkForStep / checkForRoughness   :774-921
//
// k_fp_mask   one thread per cell: the pure per-cell predicate isTraversableForFilters (the reference
//             memoises it in slope_footprint / step_footprint / roughness_footprint).  Only cells whose
//             slope/step/roughness score is exactly 0 do any work; the branchy checkForStep geometry
//             (submap, ray extension, Bresenham line) is evaluated with the reference's own double
//             formulas (this file is built with -ffp-contract=off).
// k_fp_slide  sliding-disc kernel (see te_slide_normals.hip): per lane the sum of the traversability
//             values and the number of untraversable cells of its disc, moved one row per step.  A disc
//             without untraversable cell gives the mean directly (sums of float values are exact in
//             double, so the result equals the reference's spiral-order sum bit for bit); otherwise the
//             lane walks the host-built spiral table (SpiralIterator order) over the rows already staged
//             in LDS until the first untraversable cell, exactly like isTraversable().
#include "te_internal.h"

#include <cstdlib>
#include "te_geom.h"
#include "te_march.h"

namespace te {

namespace {

constexpr int kLanes = 64;

__device__ __forceinline__ float qnanf() { return __builtin_nanf(""); }
// GridMap::getSubmap of the 2.5*res window around a candidate on a map border: the window corner beyond the
// border is clamped by boundPositionToRange to `length - eps` (eps = 10 ulp(1), scaled by |position| only
// above 1 m) and looked up again; depending on the map position that value can round onto the border itself,
// getIndexFromPosition fails, getSubmap reports failure and checkForStep returns false for every cell that has
// such a candidate (TraversabilityMap.cpp:817-822).  A property of the geometry and of the border side only:
// bit 0 candidates with i == 0, bit 1 i == rows-1, bit 2 j == 0, bit 3 j == cols-1.
inline int submap_edge_failures(const Geo& g) {
  auto bound = [](double position, double len, double mappos) {  // boundPositionToRange, one axis
    double shifted = position - mappos + 0.5 * len;
    double eps = 10.0 * 2.220446049250313e-16;
    if (fabs(position) > 1.0) eps *= fabs(position);
    if (shifted <= 0)
      shifted = eps;
    else if (shifted >= len)
      shifted = len - eps;
    return shifted + mappos - 0.5 * len;
  };
  auto ok = [](double x, double len, double mappos, double res, int n) {  // one axis of getIndexFromPosition
    const double t = -((x - mappos) - 0.5 * len);
    const int idx = (int)(-(((x - 0.5 * len) - mappos) / res));
    return t >= 0.0 && t < len && idx >= 0 && idx < n;
  };
  const double half = 0.5 * (2.5 * g.res);
  const double x0 = g.ax, x1 = g.ax + g.res * (double)(-(g.rows - 1));
  const double y0 = g.ay, y1 = g.ay + g.res * (double)(-(g.cols - 1));
  int m = 0;
  if (!ok(bound(x0 + half, g.len_x, g.pos_x), g.len_x, g.pos_x, g.res, g.rows)) m |= 1;
  if (!ok(bound(x1 - half, g.len_x, g.pos_x), g.len_x, g.pos_x, g.res, g.rows)) m |= 2;
  if (!ok(bound(y0 + half, g.len_y, g.pos_y), g.len_y, g.pos_y, g.res, g.cols)) m |= 4;
  if (!ok(bound(y1 - half, g.len_y, g.pos_y), g.len_y, g.pos_y, g.res, g.cols)) m |= 8;
  return m;
}
__device__ __forceinline__ bool submap_fails(const Geo& g, int edge_fail, int a, int b) {
  return ((edge_fail & 1) && a == 0) || ((edge_fail & 2) && a == g.rows - 1) || ((edge_fail & 4) && b == 0) ||
         ((edge_fail & 8) && b == g.cols - 1);
}

// A layer seen through the LDS tile of the block (64 x MY cells + halo, MY = 32, 8 or 4); cells outside the tile are
// read from global memory (only the rare long Bresenham walks of checkForStep leave the tile).
constexpr int MX = 64, MBY = 4, MH = 3;  // tile width, threads along j, halo (>= reach of both windows + 1)
constexpr int MTW = MX + 2 * MH;

struct TileView {
  const float* lds;
  const float* glob;
  int i0, j0, rows, th;  // th: rows of the tile (MY + 2 * MH)
  __device__ __forceinline__ float at(int a, int b) const {
    const int la = a - i0 + MH, lb = b - j0 + MH;
    if (lds && (unsigned)la < (unsigned)MTW && (unsigned)lb < (unsigned)th) return lds[lb * MTW + la];
    return glob[(size_t)b * rows + a];
  }
};


// Visit the in-map cells of CircleIterator(center (i,j), disc d): run table + per-cell test of the tie offsets.
// (The visiting order differs from CircleIterator's; the predicates below do not depend on it.)
template <typename F>
__device__ __forceinline__ void for_disc(const Geo& g, const Disc& d, int i, int j, F&& body) {
  for (int dj = -d.R; dj <= d.R; ++dj) {
    const int b = j + dj;
    if (b < 0 || b >= g.cols) continue;
    const int hw = d.hw[dj < 0 ? -dj : dj];
    for (int di = -hw; di <= hw; ++di) {
      const int a = i + di;
      if (a < 0 || a >= g.rows) continue;
      body(a, b);
    }
  }
  for (int t = 0; t < d.n_ties; ++t) {
    const int a = i + d.tie_di[t], b = j + d.tie_dj[t];
    if (a < 0 || a >= g.rows || b < 0 || b >= g.cols) continue;
    const double dx = cell_x(g, a) - cell_x(g, i), dy = cell_y(g, b) - cell_y(g, j);
    if (dx * dx + dy * dy <= d.r2) body(a, b);
  }
}

// checkForSlope :867-893 / checkForRoughness :895-921: more than ncrit zeros of `layer` in circle(3*res)?
__device__ __forceinline__ bool count_zero_ok(const Geo& g, const Disc& d, const TileView& layer, int i, int j, int ncrit) {
  // circle(3 * res) away from the border -- what nearly every call sees: the 5 x 5 block (di^2 + dj^2 <= 8) and the four axis
  // cells at distance 3, which lie ON the circle (isInside decides per centre; an axis cell's test without its exactly-zero
  // term).  All 29 loads are issued together; the general loop below takes them one at a time, and where the slope score is 0
  // in a fifth of the cells (the default 0.05 m normals radius on a 0.05 m map: discs whose cells are collinear have a
  // horizontal normal) the slow cells' list is a tenth of the tile: k_fp_mask 132 -> 105 us on 4096^2.  (The 29 live values
  // cost registers: k_fp_mask is held at 128 by amdgpu_waves_per_eu, see there.)
  if (d.R == 2 && d.hw[0] == 2 && d.hw[1] == 2 && d.hw[2] == 2 && d.n_ties == 4 && d.reach == 3 && layer.lds == nullptr && i >= 3 && j >= 3 &&
      i < g.rows - 3 && j < g.cols - 3) {
    const float* const c = layer.glob + ((size_t)j * layer.rows + i);
    const ptrdiff_t rs = (ptrdiff_t)layer.rows;
    float v[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) v[k] = c[(ptrdiff_t)(k / 5 - 2) * rs + (k % 5 - 2)];
    const float wxp = c[3], wxm = c[-3], wyp = c[3 * rs], wym = c[-3 * rs];
    int n = 0;
#pragma unroll
    for (int k = 0; k < 25; ++k) n += (v[k] == 0.0f) ? 1 : 0;
    const double xi = cell_x(g, i), yj = cell_y(g, j);
    const double dxp = cell_x(g, i + 3) - xi, dxm = cell_x(g, i - 3) - xi, dyp = cell_y(g, j + 3) - yj, dym = cell_y(g, j - 3) - yj;
    n += (dxp * dxp <= d.r2 && wxp == 0.0f) ? 1 : 0;
    n += (dxm * dxm <= d.r2 && wxm == 0.0f) ? 1 : 0;
    n += (dyp * dyp <= d.r2 && wyp == 0.0f) ? 1 : 0;
    n += (dym * dym <= d.r2 && wym == 0.0f) ? 1 : 0;
    return !(n > ncrit);
  }
  int n = 0;
  for_disc(g, d, i, j, [&](int a, int b) { n += (layer.at(a, b) == 0.0f) ? 1 : 0; });
  return !(n > ncrit);
}

// largest float <= T: for a float x,  (double)x > T <=> x > ffloor(T)
__device__ __forceinline__ float ffloor_d(double T) {
  float f = (float)T;
  if ((double)f > T)  // step one float down
    f = f > 0.0f ? __uint_as_float(__float_as_uint(f) - 1u)
                 : (f < 0.0f ? __uint_as_float(__float_as_uint(f) + 1u) : __uint_as_float(0x80000001u));
  return f;
}

// Screening pass of checkForStep entirely on the LDS tiles (cells outside the map are NaN there, which
// fails every comparison exactly like being skipped).  The window of circle(2.5*res) and the 3x3
// submaps around its candidates lie within 3 cells of the centre.
//   tkey[n] = elevation of n where its step score is 0, NaN elsewhere: n is a candidate of centre c iff
//             tkey[n] > elev[c] + crit_step (:807-809)
//   tkl[n]  = tkey[n] where additionally some cell of the 3x3 block around n has step 0 and lies more than
//             crit_step below n (the hit condition of :825, a property of n alone), NaN elsewhere
// so "some candidate has a lower step neighbour" is  max over the window of tkl > thr  and "there is a
// candidate" is  max over the window of tkey > thr  (NaN-ignoring maxima).  Without a candidate the
// centre itself is examined (:811); its step score is 0 here, so its flag is "tkl[c] is not NaN".
// Returns true when no submap cell satisfies :825, in which case checkForStep passes without any of its
// ray/line geometry; otherwise the full function decides.  General disc shapes: plain loop.
__device__ __forceinline__ bool check_step_screen(const Disc& d, const float* __restrict__ te,
                                                  const float* __restrict__ tkey, const float* __restrict__ tkl,
                                                  int ctr, double crit_step) {
  if (d.n_ties) return false;
  const float thr = ffloor_d(crit_step + (double)te[ctr]);
  bool any_cand = false, hit = false;
  for (int dj = -d.R; dj <= d.R; ++dj) {
    const int hw = d.hw[dj < 0 ? -dj : dj];
    for (int di = -hw; di <= hw; ++di) {
      const int idx = ctr + dj * MTW + di;
      any_cand |= tkey[idx] > thr;
      hit |= tkl[idx] > thr;
    }
  }
  if (!any_cand) hit = tkl[ctr] == tkl[ctr];
  return !hit;
}

// ---- checkForStep :794-865, in pieces ---------------------------------------------------------------------------------
// GridMap::getSubmap -> getSubmapInformation (grid_map_core) for a 2.5*res square around a cell centre: the corners lie
// 1.25 cells from the centre, i.e. 0.25 cells inside the neighbouring cells, or are clamped into the border cell by
// boundPositionToRange: the submap is exactly the 3x3 block clipped to the map (rounding of 1e-13 cells cannot move a
// corner across a cell boundary a quarter cell away) -- unless the clamped corner rounds onto the map border and the
// lookup fails (:818-822), which submap_edge_failures() decides per border side with the reference's own arithmetic.
struct Submap {
  int ti, tj, sr, sc;  // first cell and size; GridMapIterator runs over it with the row index fastest: a = lin % sr, b = lin / sr
};
__device__ __forceinline__ Submap submap_of(const Geo& g, int ii, int ij) {
  Submap s;
  s.ti = ii > 0 ? ii - 1 : 0;
  s.tj = ij > 0 ? ij - 1 : 0;
  const int bi = ii < g.rows - 1 ? ii + 1 : g.rows - 1, bj = ij < g.cols - 1 ? ij + 1 : g.cols - 1;
  s.sr = bi - s.ti + 1;
  s.sc = bj - s.tj + 1;
  return s;
}
// v = position of submap cell (a, b) - position of the candidate (ii, ij), with the submap's own geometry (length
// re-derived by setGeometry) as the reference computes it (:826-828)
__device__ __forceinline__ void pair_vector(const Geo& g, int ii, int ij, const Submap& s, int a, int b, double& vx, double& vy) {
  const double sx = cell_x(g, ii), sy = cell_y(g, ij);  // subMapPos
  const double tcornx = cell_x(g, s.ti) + 0.5 * g.res, tcorny = cell_y(g, s.tj) + 0.5 * g.res;
  const double slx = (double)s.sr * g.res, sly = (double)s.sc * g.res;
  const double spx = tcornx - 0.5 * slx, spy = tcorny - 0.5 * sly;
  const double px = (spx + (0.5 * slx - 0.5 * g.res)) + g.res * (double)(-a);
  const double py = (spy + (0.5 * sly - 0.5 * g.res)) + g.res * (double)(-b);
  vx = px - sx;
  vy = py - sy;
}
// :833-856 for a pair (candidate (ii, ij) of elevation `height`, direction v) that passed :825 and :829-832: the ray is
// extended up to max_gap_width, the Bresenham line to its end is walked.  true: checkForStep returns false because of
// this pair (an obstacle on the line, :840-843, or a gap that does not end, :853-856).  A property of the candidate and
// the direction alone -- not of the centre cell whose check runs into it.
__device__ bool pair_blocks(const Geo& g, const TileView& elev, int ii, int ij, double vx, double vy, double height, double crit_step,
                            double max_gap) {
  const double sx = cell_x(g, ii), sy = cell_y(g, ij);
  const double lowest = height - crit_step;
  double qx = sx + vx, qy = sy + vy;
  for (int guard = 0; guard < 100000; ++guard) {  // :834
    const double ex = (qx - sx) + vx, ey = (qy - sy) + vy;
    if (!(sqrt(ex * ex + ey * ey) < max_gap && pos_inside(g, qx + vx, qy + vy))) break;
    qx += vx;
    qy += vy;
  }
  int ei, ej;
  pos_to_index(g, qx, qy, ei, ej);
  ei = ei < 0 ? 0 : (ei > g.rows - 1 ? g.rows - 1 : ei);
  ej = ej < 0 ? 0 : (ej > g.cols - 1 ? g.cols - 1 : ej);
  // LineIterator (Bresenham, grid_map_core) from `index` to `endIndex` :839-852
  const int dx = ei > ii ? ei - ii : ii - ei, dy = ej > ij ? ej - ij : ij - ej;
  int inc1i = (ei >= ii) ? 1 : -1, inc2i = inc1i, inc1j = (ej >= ij) ? 1 : -1, inc2j = inc1j;
  int den, num, numadd, ncells;
  if (dx >= dy) {
    inc1i = 0; inc2j = 0; den = dx; num = dx / 2; numadd = dy; ncells = dx + 1;
  } else {
    inc2i = 0; inc1j = 0; den = dy; num = dy / 2; numadd = dx; ncells = dy + 1;
  }
  int li = ii, lj = ij;
  bool gap_start = false, gap_end = false;
  for (int icell = 0; icell < ncells; ++icell) {
    const float ef = elev.at(li, lj);
    if ((double)ef > height + crit_step) return true;  // :840-843
    if ((double)ef < lowest || !__builtin_isfinite(ef)) {
      gap_start = true;
    } else if (gap_start) {
      gap_end = true;
      break;
    }
    num += numadd;
    if (num >= den) {
      num -= den;
      li += inc1i;
      lj += inc1j;
    }
    li += inc2i;
    lj += inc2j;
  }
  return gap_start && !gap_end;  // :853-856
}

// checkForStep :794-865, straight: every (candidate, submap cell) pair evaluated where the centre cell meets it.  Serves the
// general disc shapes and the cells next to a border whose submap lookups fail.
__device__ bool check_step(const Geo& g, const Disc& d, const TileView& elev, const TileView& step, int ci, int cj,
                           double crit_step, double max_gap, int edge_fail) {
  const double cx = cell_x(g, ci), cy = cell_y(g, cj);
  double height = (double)elev.at(ci, cj);
  int cand[32];
  int ncand = 0;
  for_disc(g, d, ci, cj, [&](int a, int b) {
    if ((double)elev.at(a, b) > crit_step + height && step.at(a, b) == 0.0f && ncand < 32)  // :807-809
      cand[ncand++] = (a << 16) | b;
  });
  if (ncand == 0) cand[ncand++] = (ci << 16) | cj;  // :811
  for (int c = 0; c < ncand; ++c) {
    const int ii = cand[c] >> 16, ij = cand[c] & 0xffff;
    if (submap_fails(g, edge_fail, ii, ij)) return false;  // :817-822
    const Submap sm = submap_of(g, ii, ij);
    height = (double)elev.at(ii, ij);  // :823
    const double lowest = height - crit_step;
    const double tcx = cx - cell_x(g, ii), tcy = cy - cell_y(g, ij);  // toCenter :816
    for (int lin = 0; lin < sm.sr * sm.sc; ++lin) {  // GridMapIterator over the submap: row index fastest
      const int a = lin % sm.sr, b = lin / sm.sr;
      if (!(step.at(sm.ti + a, sm.tj + b) == 0.0f && (double)elev.at(sm.ti + a, sm.tj + b) < lowest)) continue;  // :825
      double vx, vy;
      pair_vector(g, ii, ij, sm, a, b, vx, vy);
      if (sqrt(vx * vx + vy * vy) < 0.025) continue;  // :829
      if (sqrt(tcx * tcx + tcy * tcy) > 0.025) {      // :830-832
        if (tcx * vx + tcy * vy < 0.0) continue;
      }
      if (pair_blocks(g, elev, ii, ij, vx, vy, height, crit_step, max_gap)) return false;
    }
  }
  return true;
}

// ---- the same with the pairs' ray / line tests MEMOISED per candidate (k_fp_mask, tiles that hold a vertical face) ----
// Whether a pair (candidate n, submap cell m) makes checkForStep return false -- :825 holds for m, m is not n itself, and
// pair_blocks() -- does not depend on the centre cell; only the direction filter :830-832 does.  Next to a kerb every
// candidate is met by up to 21 centres (circle(2.5 res)), each of which used to walk the candidate's rays again: one cell
// beside a tall edge ran 189 pairs, ~1500 instructions each, 70 us in its thread, and the tile waited for it.  Now a tile
// evaluates the pairs of its candidates ONCE (pair_bit: one bit per submap cell, for every tile cell that can be a
// candidate and has a lower step neighbour -- the cells of t_kl; one PAIR per thread: a candidate on a straight edge has
// three pairs to walk, and a thread that took them all was the tile's critical path), and a centre's check is a look at 21
// masks plus the direction filter for the set bits.
// one pair: submap cell `lin` of candidate (ii, ij) (tile index idx); true: its bit belongs into the candidate's mask
__device__ bool pair_bit(const Geo& g, const TileView& elev, const float* __restrict__ t_elev, const float* __restrict__ t_key, int idx,
                         int ii, int ij, int lin, double crit_step, double max_gap) {
  const Submap sm = submap_of(g, ii, ij);
  if (lin >= sm.sr * sm.sc) return false;
  const double height = (double)t_elev[idx];  // :823
  const double lowest = height - crit_step;
  const int a = lin % sm.sr, b = lin / sm.sr;
  const int m = idx + (sm.tj + b - ij) * MTW + (sm.ti + a - ii);
  const float km = t_key[m];  // elevation where the step score is 0 (NaN elsewhere): :825
  if (!((double)km < lowest)) return false;
  double vx, vy;
  pair_vector(g, ii, ij, sm, a, b, vx, vy);
  if (sqrt(vx * vx + vy * vy) < 0.025) return false;  // :829
  return pair_blocks(g, elev, ii, ij, vx, vy, height, crit_step, max_gap);
}
// circle(2.5 res) is the tie-free shape Q = 5 (rows dj = 0, +-1 span |di| <= 2, rows dj = +-2 span |di| <= 1); cells
// outside the map are NaN in t_key and fail the candidate test like being skipped.
__device__ bool check_step_memo(const Geo& g, const float* __restrict__ t_elev, const float* __restrict__ t_key,
                                const unsigned short* __restrict__ fmask, int ctr, int ci, int cj, double crit_step, int edge_fail) {
  const double cx = cell_x(g, ci), cy = cell_y(g, cj);
  const double thr = crit_step + (double)t_elev[ctr];
  int ncand = 0;
  for (int dj = -2; dj <= 2; ++dj) {
    const int hw = (dj == -2 || dj == 2) ? 1 : 2;
    for (int di = -hw; di <= hw; ++di) {
      const int idx = ctr + dj * MTW + di;
      if (!((double)t_key[idx] > thr)) continue;  // :807-809 (t_key: the elevation where the step score is 0)
      ++ncand;
      const int ii = ci + di, ij = cj + dj;
      if (submap_fails(g, edge_fail, ii, ij)) return false;  // :817-822
      unsigned f = fmask[idx];
      if (f == 0u) continue;
      const Submap sm = submap_of(g, ii, ij);
      const double tcx = cx - cell_x(g, ii), tcy = cy - cell_y(g, ij);  // toCenter :816
      const bool filter = sqrt(tcx * tcx + tcy * tcy) > 0.025;           // :830
      while (f) {
        const int lin = __ffs((int)f) - 1;
        f &= f - 1u;
        if (filter) {
          double vx, vy;
          pair_vector(g, ii, ij, sm, lin % sm.sr, lin / sm.sr, vx, vy);
          if (tcx * vx + tcy * vy < 0.0) continue;  // :831
        }
        return false;
      }
    }
  }
  if (ncand == 0) {  // :811 the centre is its own candidate; toCenter = 0: no direction filter
    if (submap_fails(g, edge_fail, ci, cj)) return false;
    if (fmask[ctr] != 0) return false;
  }
  return true;
}

struct MaskArgs {
  Disc slope_disc;  // circle(3*res)
  Disc step_disc;   // circle(2.5*res)
  int ncrit_slope, ncrit_rough, check_rough, write_memo;
  double crit_step, max_gap;
  int edge_fail;  // submap_edge_failures(): border sides on which the 2.5*res submap lookup fails
  int combine;  // also write traversability = w_scale*((w_slope*slope + w_step*step) + w_rough*roughness) (float32)
  float w_scale, w_slope, w_step, w_rough;
  int ti0, tj0, map;  // first tile of the launch (in tiles of MX x MY cells) and the map (< 0: blockIdx.z): region runs
  unsigned* blocked_count;  // the first mask launch of a pass empties k_fp_slide4's list (nullptr: not this launch)
  // one byte per 64 x 4 cells (all maps): "holds an untraversable cell".  A block clears the flags of its tile before its
  // first barrier and a thread that finds a cell untraversable -- behind the barriers -- sets the flag of the cell's rows:
  // after the kernel a clear flag says that none of the 256 mask bytes is 1.  (k_fp_slide5 reads no mask bytes where
  // every flag of its strip is clear.)
  uint8_t* untrav_flags;
  int flag_ntx, flag_nfy;
  int whatif;  // lab library only (TE_MASK_WHATIF; 0 in the product): 1 no pair masks are evaluated, 2 the slow cells are not decided
};

// isTraversableForFilters :774-792 for every cell of a 64 x MY tile; every thread owns MY / 4 cells of a column.  MY = 8
// (4 for very small maps) for maps too small to fill the GPU with 64 x 32 tiles: a thread's cells that need the full checkForStep are serial,
// and on the reference's own 100 x 133 map (where half of the cells do) ten workgroups took 0.32 ms.
// (amdgpu_waves_per_eu(4, 4): four waves per SIMD, 128 registers.  Left to itself the compiler takes 132-136 once
// count_zero_ok holds its 29 values, and every tile -- with or without such cells -- runs at three waves: 67 -> 81 us on the
// bench map.  Held at 128 it spills 144 bytes on the slow cells' path and the straight-line pass keeps its time.)
template <int MY>
__global__ __launch_bounds__(MX* MBY) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_fp_mask(Geo g, MaskArgs a, const float* __restrict__ elev,
                                                     const float* __restrict__ slope, const float* __restrict__ step,
                                                     const float* __restrict__ rough, uint8_t* __restrict__ untrav,
                                                     float* __restrict__ slope_fp, float* __restrict__ step_fp,
                                                     float* __restrict__ rough_fp, float* __restrict__ trav) {
  // t_elev = elevation; t_key / t_kl: see check_step_screen (NaN outside the map).  Slope / roughness scores
  // are only needed at the centre cell (plus, for the rare zero scores, their window): they are read
  // straight from global memory.
  constexpr int MTH = MY + 2 * MH;
  __shared__ float t_elev[MTW * MTH], t_key[MTW * MTH], t_kl[MTW * MTH];
  // the slow cells' list (below): entries, wavefronts past the screening pass, wavefronts with slow cells, of those the ones
  // whose entries are in the list -- zeroed here, before the first barrier
  __shared__ int ntodo, arrived, members, compacted, nkl;
  if (threadIdx.x == 0 && threadIdx.y == 0) ntodo = arrived = members = compacted = nkl = 0;
  const size_t mo = (size_t)(a.map >= 0 ? a.map : (int)blockIdx.z) * g.rows * g.cols;
  const int i0 = ((int)blockIdx.x + a.ti0) * MX, j0 = ((int)blockIdx.y + a.tj0) * MY;
  // (the footprint pass's list of blocked cells starts empty: k_fp_slide4 / k_fp_blocked run after this kernel)
  if (a.blocked_count && threadIdx.x == 0 && threadIdx.y == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
    a.blocked_count[0] = a.blocked_count[1] = a.blocked_count[4] = 0u;
  // my tile's flags (flag rows j0 / 4 .. of flag column i0 / 64: i0 and j0 are multiples of 64 and of MY)
  uint8_t* const my_flags = a.untrav_flags + ((size_t)(a.map >= 0 ? a.map : (int)blockIdx.z) * a.flag_nfy) * a.flag_ntx + (i0 >> 6);
  {
    constexpr int NF = MY / 4;
    const int t = threadIdx.y * MX + threadIdx.x;
    if (t < NF && (j0 >> 2) + t < a.flag_nfy) my_flags[(size_t)((j0 >> 2) + t) * a.flag_ntx] = 0;
  }
  // The three scores of this thread's MY/MBY cells: issued together with the tile loads, so that they
  // are in flight during the staging and the two LDS passes (clamped rows; a thread beyond the last column has nothing to do but must reach the barrier).
  constexpr int NC = MY / MBY;
  const int i = i0 + threadIdx.x;
  const int ic = i < g.rows ? i : g.rows - 1;
  // (MEASURED: making the row index a scalar -- readfirstlane of threadIdx.y, uniform per wavefront -- and hoisting the
  // per-cell argument reads lowered the static instruction count of the straight-line pass by 5 % and RAISED the executed
  // one by 13 % (1723 instead of 1522 per wavefront, SQ_INSTS_*): the extra scalar values spill to VGPR lanes,
  // v_readlane / v_writelane around every use.  The kernel time did not move either way; the simple form stays.)
  const int jb = threadIdx.y * NC;  // first tile row of this thread
  float s_slope[NC], s_step[NC], s_rough[NC];
  fast::static_for<NC>([&](auto cc) __attribute__((always_inline)) {
    constexpr int c = decltype(cc)::value;
    int j = j0 + jb + c;
    j = j < g.cols ? j : g.cols - 1;
    const size_t o = mo + (size_t)j * g.rows + ic;
    s_slope[c] = slope[o];
    s_step[c] = step[o];
    s_rough[c] = (a.check_rough || a.combine) ? rough[o] : 1.0f;
  });
  {
    // all loads of the tile in flight at once (clamped addresses), then the LDS writes.  A thread stages its own column
    // (tile column threadIdx.x + MH) in the tile rows threadIdx.y, + MBY, ... -- one clamped column, row offsets that are
    // multiples of MBY map rows, LDS addresses that differ by constants -- and the first 2 * MH * MTH threads one cell of
    // the 2 * MH halo columns each.  (Round 1-4 flattened the tile over the threads: a division by MTW, two clamps and a
    // 64-bit address per cell and pass, 40 % of the kernel's vector instructions.)
    constexpr int KM = (MTH + MBY - 1) / MBY;
    const int tid = threadIdx.y * MX + threadIdx.x;
    const int ac = i0 + (int)threadIdx.x;
    const bool col_in = ac < g.rows;
    const float* const pe = elev + mo + (col_in ? ac : g.rows - 1);
    const float* const ps = step + mo + (col_in ? ac : g.rows - 1);
    float le[KM], ls[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      int bb = j0 - MH + (int)threadIdx.y + k * MBY;
      bb = bb < 0 ? 0 : (bb >= g.cols ? g.cols - 1 : bb);
      const size_t o = (size_t)bb * g.rows;
      le[k] = pe[o];
      ls[k] = ps[o];
    }
    // halo: tile columns 0 .. MH-1 and MX+MH .. MTW-1
    constexpr int NH = 2 * MH * MTH;
    static_assert(NH <= MX * MBY, "one halo cell per thread");
    const int hj = tid / (2 * MH), hc = tid - hj * (2 * MH);
    const int hti = hc < MH ? hc : MX + hc;  // tile column
    const int ha = i0 - MH + hti, hb = j0 - MH + hj;
    float he = 0.0f, hs = 0.0f;
    if (tid < NH) {
      const int aa = ha < 0 ? 0 : (ha >= g.rows ? g.rows - 1 : ha), bb = hb < 0 ? 0 : (hb >= g.cols ? g.cols - 1 : hb);
      const size_t o = mo + (size_t)bb * g.rows + aa;
      he = elev[o];
      hs = step[o];
    }
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      const int tj = (int)threadIdx.y + k * MBY, bb = j0 - MH + tj;
      const bool in = col_in && bb >= 0 && bb < g.cols;
      if (tj < MTH) {
        const int idx = tj * MTW + (int)threadIdx.x + MH;
        t_elev[idx] = in ? le[k] : qnanf();
        t_key[idx] = (in && ls[k] == 0.0f) ? le[k] : qnanf();
      }
    }
    if (tid < NH) {
      const bool in = ha >= 0 && ha < g.rows && hb >= 0 && hb < g.cols;
      const int idx = hj * MTW + hti;
      t_elev[idx] = in ? he : qnanf();
      t_key[idx] = (in && hs == 0.0f) ? he : qnanf();
    }
  }
  __syncthreads();
  bool any_kl = false;  // some cell of my share has a lower step neighbour (see below: without one in the whole tile the windows have nothing to find)
  {
    // t_kl for the tile cells the windows can reach (rows 1..MTH-2, columns 1..MTW-2): the 3x3 minimum of
    // t_key slides down a column (row minimum of 3 cells, then the minimum of 3 consecutive rows).
    // Columns 2..65 are walked by the 64 lanes (MTH-2 = 36 rows in MBY segments of 9); the four remaining
    // columns are done cell by cell.
    static_assert((MTH - 2) % MBY == 0, "row segments");
    constexpr int SEG = (MTH - 2) / MBY;
    const int la = threadIdx.x + 2, r0 = 1 + threadIdx.y * SEG;
    float rm[3];
    auto row_min = [&](int r) {
      const float* k = t_key + r * MTW + la;
      return fast::vmin3(k[-1], k[0], k[1]);
    };
    rm[0] = row_min(r0 - 1);
    rm[1] = row_min(r0);
    fast::static_for<SEG>([&](auto rc) __attribute__((always_inline)) {
      constexpr int q = decltype(rc)::value;
      const int r = r0 + q;
      const float next = row_min(r + 1);  // slots rotate with q: rows r-1, r, r+1 in rm[q%3], rm[(q+1)%3], rm[(q+2)%3]
      const float m = fast::vmin3(rm[q % 3], rm[(q + 1) % 3], next);
      rm[(q + 2) % 3] = next;
      const int idx = r * MTW + la;
      const bool hit = (double)m < (double)t_elev[idx] - a.crit_step;  // :825 in the reference's double arithmetic
      t_kl[idx] = hit ? t_key[idx] : qnanf();
      any_kl |= hit;
    });
    const int tid = threadIdx.y * MX + threadIdx.x;
    if (tid < 4 * (MTH - 2)) {
      const int col = (tid & 3) == 0 ? 1 : MTW - 5 + (tid & 3);  // 1, MTW-4, MTW-3, MTW-2
      const int idx = (1 + (tid >> 2)) * MTW + col;
      const float* k = t_key + idx;
      const float m = fast::vmin3(fast::vmin3(k[-MTW - 1], k[-MTW], k[-MTW + 1]), fast::vmin3(k[-1], k[0], k[1]),
                                  fast::vmin3(k[MTW - 1], k[MTW], k[MTW + 1]));
      const bool hit = (double)m < (double)t_elev[idx] - a.crit_step;
      t_kl[idx] = hit ? t_key[idx] : qnanf();
      any_kl |= hit;
    }
  }
  // (the barrier that publishes t_kl also tells whether the tile holds ANY lower step neighbour: on terrain without
  // vertical faces -- a drop of more than crit_step between adjacent cells -- no tile does, t_kl is NaN throughout, both
  // branches of the screen's test are false for every cell (NaN > thr, NaN == NaN) and the window maxima need not be
  // formed at all: a third of the kernel's instructions on the bench map, where the step score is 0 in 98 % of the cells)
  const bool tile_has_kl = __syncthreads_or(any_kl ? 1 : 0) != 0;
  const TileView ve = {t_elev, elev + mo, i0, j0, g.rows, MTH}, vs = {nullptr, step + mo, i0, j0, g.rows, MTH},
                 vl = {nullptr, slope + mo, i0, j0, g.rows, MTH}, vr = {nullptr, rough + mo, i0, j0, g.rows, MTH};
  const bool in_map = i < g.rows;  // (a thread beyond the last column takes no cells of its own but helps with the list below)
  // Every thread walks MY/MBY consecutive rows of its column.  For the 21-cell window of circle(2.5*res)
  // (di^2+dj^2 <= 5: rows dj=0,+-1 span |di|<=2, rows dj=+-2 span |di|<=1) the two window maxima slide:
  // each tile row is reduced once along i (H1 = max over |di|<=1, H2 = max over |di|<=2: 5 LDS reads and
  // 2 v_max3 per array) and an output combines the run values of its 5 rows with 2 more v_max3.
  const bool q5 = a.step_disc.Q == 5 && a.step_disc.n_ties == 0;
  float h1k[5], h2k[5], h1l[5], h2l[5];  // run maxima of the last 5 tile rows (slot = row mod 5)
  auto reduce_row = [&](int row, int slot) {  // tile row jb + row (window rows start 2 above the outputs)
    const int base = (jb + row + MH) * MTW + (threadIdx.x + MH);
    const float* k = t_key + base;
    const float* l = t_kl + base;
    h1k[slot] = fast::vmax3(k[-1], k[0], k[1]);
    h2k[slot] = fast::vmax3(h1k[slot], k[-2], k[2]);
    h1l[slot] = fast::vmax3(l[-1], l[0], l[1]);
    h2l[slot] = fast::vmax3(h1l[slot], l[-2], l[2]);
  };
  unsigned screen_mask = 0;  // bit c: the screening pass clears my c-th cell
  if (q5 && in_map && !tile_has_kl) screen_mask = (1u << NC) - 1u;
  if (q5 && in_map && tile_has_kl) {
    fast::static_for<4>([&](auto rc) __attribute__((always_inline)) {
      constexpr int r = decltype(rc)::value;
      reduce_row(r - 2, r);  // rows -2 .. 1
    });
    fast::static_for<NC>([&](auto cc) __attribute__((always_inline)) {
      constexpr int c = decltype(cc)::value;
      constexpr int s2 = (c + 4) % 5;  // slot of row c+2
      reduce_row(c + 2, s2);
      constexpr int sm2 = c % 5, sm1 = (c + 1) % 5, s0 = (c + 2) % 5, s1 = (c + 3) % 5;
      const float mk = fast::vmax3(fast::vmax3(h1k[sm2], h2k[sm1], h2k[s0]), h2k[s1], h1k[s2]);
      const float ml = fast::vmax3(fast::vmax3(h1l[sm2], h2l[sm1], h2l[s0]), h2l[s1], h1l[s2]);
      const int ctr = (jb + c + MH) * MTW + (threadIdx.x + MH);
      const double thr = a.crit_step + (double)t_elev[ctr];  // :807 (double comparison, like the reference)
      const float lc = t_kl[ctr];
      const bool hit = ((double)mk > thr) ? ((double)ml > thr) : (lc == lc);
      screen_mask |= hit ? 0u : (1u << c);
    });
  }
  // Straight-line pass over my cells: almost every cell is decided by its scores and the screening bit alone
  // (traversable; combined layer written); the few that need a window count or the full checkForStep are
  // collected in a bit mask and handled by the rolled loop below.
  unsigned slow_mask = 0;
  fast::static_for<NC>([&](auto cc) __attribute__((always_inline)) {
    constexpr int c = decltype(cc)::value;
    const int j = j0 + jb + c;
    const float c_slope = s_slope[c], c_step = s_step[c], c_rough = s_rough[c];
    bool near_bad_edge = false;
    if (__builtin_expect(a.edge_fail != 0, 0))  // (uniform; a map whose border makes the submap lookup fail is the exception)
      near_bad_edge = ((a.edge_fail & 1) && i <= 2) || ((a.edge_fail & 2) && i >= g.rows - 3) || ((a.edge_fail & 4) && j <= 2) ||
                      ((a.edge_fail & 8) && j >= g.cols - 3);
    const bool step_fast = !(c_step == 0.0f) || (q5 && ((screen_mask >> c) & 1u) != 0 && !near_bad_edge);
    const bool slow = (c_slope == 0.0f) || !step_fast || (a.check_rough && c_rough == 0.0f);
    slow_mask |= (slow && j < g.cols && in_map) ? (1u << c) : 0u;
    if (j < g.cols && in_map) {
      const size_t o = mo + (size_t)j * g.rows + i;
      if (!slow) untrav[o] = 0;
      if (a.combine) {  // MathExpressionFilter, fixed form, float32, left to right
        const float ta = a.w_slope * c_slope, tb = a.w_step * c_step, tc = a.w_rough * c_rough;
        const float tab = ta + tb;
        const float tabc = tab + tc;
        trav[o] = a.w_scale * tabc;
      }
      if (a.write_memo && !slow) {  // a step check that the screen clears is memoised as passed (:857)
        slope_fp[o] = qnanf();
        step_fp[o] = (c_step == 0.0f) ? 1.0f : qnanf();
        rough_fp[o] = qnanf();
      }
    }
  });
  // The cells that need a window count or the full checkForStep.  A thread's own such cells used to be its own serial
  // loop -- next to a kerb a few threads of a tile carry hundreds of them while the other lanes idle: a 4096^2 map with
  // 3000 boxes took 2.2 ms in this kernel against 0.07 ms without.  They go to a tile-wide list instead (order-
  // preserving compaction: neighbouring cells land in neighbouring lanes and meet similar work) and all 256 threads
  // take them in turn.  An entry is  tile row << 7 | screening bit << 6 | tile column.  The list lives in t_kl, which
  // nothing reads any more once the screening pass is through (circle(2.5 res) is always the tie-free shape Q = 5:
  // 6.25 is not a sum of two squares; the general-shape screen below is kept for completeness and reads t_kl -- it
  // then takes the cells thread by thread as before).
  static_assert(sizeof(t_kl) >= MX * MY * sizeof(unsigned short), "the list fits the tile it replaces");
  unsigned short* const todo = reinterpret_cast<unsigned short*>(t_kl);
  if (!q5) {
#pragma unroll 1
    while (slow_mask) {
      const int c = __ffs((int)slow_mask) - 1;
      slow_mask &= slow_mask - 1;
      const int j = j0 + jb + c;
      const size_t o = mo + (size_t)j * g.rows + i;
      const float c_slope = slope[o], c_step = step[o], c_rough = (a.check_rough || a.combine) ? rough[o] : 1.0f;
      float m_slope = qnanf(), m_step = qnanf(), m_rough = qnanf();
      bool ok = true;
      const int ctr = (jb + c + MH) * MTW + (threadIdx.x + MH);
      if (c_slope == 0.0f) {
        ok = count_zero_ok(g, a.slope_disc, vl, i, j, a.ncrit_slope);
        m_slope = ok ? 1.0f : 0.0f;
      }
      if (ok && c_step == 0.0f) {
        const bool screen_ok = check_step_screen(a.step_disc, t_elev, t_key, t_kl, ctr, a.crit_step);
        const bool near_bad_edge = a.edge_fail && (((a.edge_fail & 1) && i <= 2) || ((a.edge_fail & 2) && i >= g.rows - 3) ||
                                                    ((a.edge_fail & 4) && j <= 2) || ((a.edge_fail & 8) && j >= g.cols - 3));
        ok = (screen_ok && !near_bad_edge) || check_step(g, a.step_disc, ve, vs, i, j, a.crit_step, a.max_gap, a.edge_fail);
        m_step = ok ? 1.0f : 0.0f;
      }
      if (ok && a.check_rough && c_rough == 0.0f) {
        ok = count_zero_ok(g, a.slope_disc, vr, i, j, a.ncrit_rough);
        m_rough = ok ? 1.0f : 0.0f;
      }
      untrav[o] = ok ? 0 : 1;
      if (!ok) my_flags[(size_t)(j >> 2) * a.flag_ntx] = 1;
      if (a.write_memo) {
        slope_fp[o] = m_slope;
        step_fp[o] = m_step;
        rough_fp[o] = m_rough;
      }
    }
    return;  // (uniform)
  }
  if (tile_has_kl) {  // (uniform)
    // The tile holds a vertical face.  Every wavefront stays (real barriers: such a tile has slow cells to share):
    //   1. the candidates with a lower step neighbour -- the non-NaN cells of t_kl -- are collected from the whole tile
    //      (own cells and halo), by the threads that computed them;
    //   2. their pair masks are evaluated, 256 (candidate, submap cell) pairs at a time (pair_bit: the reference's ray /
    //      line geometry, once per candidate instead of once per centre that meets it);
    //   3. the slow cells are listed and shared as below, their step check being check_step_memo.
    // t_kl is dead once every thread is through the screening pass: its memory holds the candidate list (later the slow
    // cells' list) and the pair masks, one unsigned short each per tile cell.
    constexpr int NCELL = MTW * MTH;
    static_assert(sizeof(t_kl) >= 2 * NCELL * sizeof(unsigned short), "candidate list + pair masks fit the tile they replace");
    static_assert(NCELL % 2 == 0, "pair masks are cleared word by word");
    unsigned short* const klist = reinterpret_cast<unsigned short*>(t_kl);
    unsigned short* const fmask = klist + NCELL;
    const int tid = threadIdx.y * MX + threadIdx.x;
    // (1) my share of t_kl, read before anything overwrites it: the column segment of the 3x3-minimum pass + one edge cell
    constexpr int SEG = (MTH - 2) / MBY;
    unsigned klbits = 0;
    {
      const int la = threadIdx.x + 2, r0 = 1 + threadIdx.y * SEG;
#pragma unroll
      for (int q = 0; q < SEG; ++q) {
        const float v = t_kl[(r0 + q) * MTW + la];
        klbits |= (v == v) ? (1u << q) : 0u;
      }
      if (tid < 4 * (MTH - 2)) {
        const int col = (tid & 3) == 0 ? 1 : MTW - 5 + (tid & 3);
        const float v = t_kl[(1 + (tid >> 2)) * MTW + col];
        klbits |= (v == v) ? 0x80000000u : 0u;
      }
    }
    __syncthreads();
    // (Staging the zero-score flags of the slope / roughness layers here as well, so that checkForSlope's window count reads
    // LDS instead of 29 + 4 cells from L2 per slow cell, was built and measured in round 5 like in round 3: no gain --
    // 3 boxes 106 -> 112 us, 300 boxes 142 -> 150 us for this kernel, profiles/r05_experiments.json.)
    for (int k = tid; k < NCELL / 2; k += MX * MBY) reinterpret_cast<unsigned*>(fmask)[k] = 0u;
    if (klbits != 0u) {
      int at = atomicAdd(&nkl, __popc(klbits));
      const int la = threadIdx.x + 2, r0 = 1 + threadIdx.y * SEG;
      unsigned bits = klbits & 0x7fffffffu;
      while (bits) {
        const int q = __ffs((int)bits) - 1;
        bits &= bits - 1u;
        klist[at++] = (unsigned short)((r0 + q) * MTW + la);
      }
      if (klbits & 0x80000000u) klist[at++] = (unsigned short)((1 + (tid >> 2)) * MTW + ((tid & 3) == 0 ? 1 : MTW - 5 + (tid & 3)));
    }
    __syncthreads();
    // (2) one (candidate, submap cell) pair per thread; the few that block OR their bit into the candidate's mask
    {
      const int n_jobs = (a.whatif & 1) ? 0 : nkl * 9;
#pragma unroll 1
      for (int k = tid; k < n_jobs; k += MX * MBY) {
        const int idx = klist[k / 9], lin = k - (k / 9) * 9;
        const int lb = idx / MTW, la2 = idx - lb * MTW;
        if (pair_bit(g, ve, t_elev, t_key, idx, i0 - MH + la2, j0 - MH + lb, lin, a.crit_step, a.max_gap))
          atomicOr(reinterpret_cast<unsigned*>(fmask) + (idx >> 1), (1u << lin) << ((idx & 1) * 16));
      }
    }
    __syncthreads();
    // (3) the slow cells, into the candidate list's memory
    unsigned short* const todo2 = klist;
    fast::static_for<NC>([&](auto cc) __attribute__((always_inline)) {
      constexpr int c = decltype(cc)::value;
      const bool need = ((slow_mask >> c) & 1u) != 0;
      const unsigned long long bm = __ballot(need);
      if (bm != 0ull) {  // uniform
        int base = 0;
        if (threadIdx.x == 0) base = atomicAdd(&ntodo, __popcll(bm));
        base = __shfl(base, 0);
        if (need)
          todo2[base + __popcll(bm & ((1ull << threadIdx.x) - 1ull))] =
              (unsigned short)(((jb + c) << 7) | ((((screen_mask >> c) & 1u) != 0 ? 1 : 0) << 6) | (int)threadIdx.x);
      }
    });
    __syncthreads();
    const int n_todo2 = (a.whatif & 2) ? 0 : ntodo;
#pragma unroll 1
    for (int k = tid; k < n_todo2; k += MX * MBY) {
      const int e = todo2[k];
      const int li = e & 63, lj = e >> 7;
      const bool screened = ((e >> 6) & 1) != 0;
      const int ci = i0 + li, j = j0 + lj;
      const size_t o = mo + (size_t)j * g.rows + ci;
      const float c_slope = slope[o], c_step = step[o], c_rough = (a.check_rough || a.combine) ? rough[o] : 1.0f;
      float m_slope = qnanf(), m_step = qnanf(), m_rough = qnanf();
      bool ok = true;
      if (c_slope == 0.0f) {  // checkForSlope
        ok = count_zero_ok(g, a.slope_disc, vl, ci, j, a.ncrit_slope);
        m_slope = ok ? 1.0f : 0.0f;
      }
      if (ok && c_step == 0.0f) {  // checkForStep
        const bool near_bad_edge = a.edge_fail && (((a.edge_fail & 1) && ci <= 2) || ((a.edge_fail & 2) && ci >= g.rows - 3) ||
                                                    ((a.edge_fail & 4) && j <= 2) || ((a.edge_fail & 8) && j >= g.cols - 3));
        ok = (screened && !near_bad_edge) ||
             check_step_memo(g, t_elev, t_key, fmask, (lj + MH) * MTW + (li + MH), ci, j, a.crit_step, a.edge_fail);
        m_step = ok ? 1.0f : 0.0f;
      }
      if (ok && a.check_rough && c_rough == 0.0f) {  // checkForRoughness
        ok = count_zero_ok(g, a.slope_disc, vr, ci, j, a.ncrit_rough);
        m_rough = ok ? 1.0f : 0.0f;
      }
      untrav[o] = ok ? 0 : 1;
      if (!ok) my_flags[(size_t)(j >> 2) * a.flag_ntx] = 1;
      if (a.write_memo) {
        slope_fp[o] = m_slope;
        step_fp[o] = m_step;
        rough_fp[o] = m_rough;
      }
    }
    return;  // (uniform)
  }
  // No barrier: on a map without obstacles no tile has a slow cell, and two barriers per tile cost the mask kernel 5 of
  // its 72 us.  A wavefront (one tile row of threads) without slow cells signs off and leaves; the others wait -- LDS
  // counters, all wavefronts of a workgroup are resident together -- until every wavefront is through the screening pass
  // (the list lives in t_kl), put their cells into the list, wait for one another's entries and share the list.
  const bool mine = __ballot(slow_mask != 0u) != 0ull;  // (uniform: a wavefront is the MX threads of one threadIdx.y)
  int rank = 0;
  // (release / acquire at workgroup scope around the counters: the LDS executes a wavefront's operations in order, the
  // fences keep the COMPILER from moving the plain accesses to t_kl / todo across the relaxed atomics and volatile reads)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (threadIdx.x == 0) {
    if (mine) rank = atomicAdd(&members, 1);
    atomicAdd(&arrived, 1);  // (after my last read of t_kl and after `members`: LDS operations of a wavefront execute in order)
  }
  if (!mine) return;
  rank = __builtin_amdgcn_readfirstlane(rank);
  while (*(volatile int*)&arrived < MBY) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const int n_members = *(volatile int*)&members;
  fast::static_for<NC>([&](auto cc) __attribute__((always_inline)) {
    constexpr int c = decltype(cc)::value;
    const bool need = ((slow_mask >> c) & 1u) != 0;
    const unsigned long long bm = __ballot(need);
    if (bm != 0ull) {  // uniform
      int base = 0;
      if (threadIdx.x == 0) base = atomicAdd(&ntodo, __popcll(bm));
      base = __shfl(base, 0);
      if (need)
        todo[base + __popcll(bm & ((1ull << threadIdx.x) - 1ull))] =
            (unsigned short)(((jb + c) << 7) | ((((screen_mask >> c) & 1u) != 0 ? 1 : 0) << 6) | (int)threadIdx.x);
    }
  });
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (threadIdx.x == 0) atomicAdd(&compacted, 1);  // (behind my entries)
  while (*(volatile int*)&compacted < n_members) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const int n_todo = *(volatile int*)&ntodo;
#pragma unroll 1
  for (int k = rank * MX + (int)threadIdx.x; k < n_todo; k += n_members * MX) {
    const int e = todo[k];
    const int li = e & 63, lj = e >> 7;
    const bool screened = ((e >> 6) & 1) != 0;
    const int ci = i0 + li, j = j0 + lj;
    const size_t o = mo + (size_t)j * g.rows + ci;
    const float c_slope = slope[o], c_step = step[o], c_rough = (a.check_rough || a.combine) ? rough[o] : 1.0f;
    float m_slope = qnanf(), m_step = qnanf(), m_rough = qnanf();
    bool ok = true;
    if (c_slope == 0.0f) {  // checkForSlope
      ok = count_zero_ok(g, a.slope_disc, vl, ci, j, a.ncrit_slope);
      m_slope = ok ? 1.0f : 0.0f;
    }
    if (ok && c_step == 0.0f) {  // checkForStep
      const bool screen_ok = screened;
      // (the screen knows nothing about failing submap lookups: next to such a border the full function decides)
      const bool near_bad_edge = a.edge_fail && (((a.edge_fail & 1) && ci <= 2) || ((a.edge_fail & 2) && ci >= g.rows - 3) ||
                                                  ((a.edge_fail & 4) && j <= 2) || ((a.edge_fail & 8) && j >= g.cols - 3));
      ok = (screen_ok && !near_bad_edge) || check_step(g, a.step_disc, ve, vs, ci, j, a.crit_step, a.max_gap, a.edge_fail);
      m_step = ok ? 1.0f : 0.0f;
    }
    if (ok && a.check_rough && c_rough == 0.0f) {  // checkForRoughness
      ok = count_zero_ok(g, a.slope_disc, vr, ci, j, a.ncrit_rough);
      m_rough = ok ? 1.0f : 0.0f;
    }
    untrav[o] = ok ? 0 : 1;
    if (!ok) my_flags[(size_t)(j >> 2) * a.flag_ntx] = 1;
    if (a.write_memo) {
      slope_fp[o] = m_slope;
      step_fp[o] = m_step;
      rough_fp[o] = m_rough;
    }
  }
}

struct SpiralArgs {
  int h[kMaxRadiusCells + 1];  // half-heights of the tie-free part of the footprint disc
  int n_ties;
  int8_t tie_di[kMaxTies], tie_dj[kMaxTies];
  double r2;
  int n_spiral;               // entries of the ordered offset table
  const int16_t* table;       // [n_spiral][4]: di, dj, ring (integer norm), tie flag
  const int* gtab;            // clip table of the tie-free part: {n, ...} per (ky, kx)
  double rmin, rmax, def;
  int out_rows;
  int inner_q;  // see k_fp_slide, step (0)
  int map0;     // first map of the launch (a region run covers its own map only)
};

// Ring encoding: one double per cell, T' + kUOff * U with T' = traversability (NaN -> default) and
// U = 1 for an untraversable cell.  T' is a float in (-kUOff/2, kUOff/2) (scores live in [0, 1]), so
// both the per-cell value and the disc sum  sum(T') + kUOff * sum(U)  are exact in double and split
// back exactly:  sum(U) = floor((S + kUOff/2) / kUOff).
constexpr double kUOff = 4096.0;
constexpr int kFpHead = 24;  // spiral entries every lane walks on its own before the wavefront takes the long walks over
constexpr int kFpWaves = 2;  // waves per SIMD k_fp_slide is compiled for; the launcher fills exactly these slots

// Q >= 0: the (tie-free) disc shape is the compile-time shape fast::Shape<Q> (R == Shape<Q>::R): the ring
// rows of every column are then fixed positions of a rotating table of row offsets and the per-row
// wrap-around bookkeeping (3 scalar instructions per column and row) disappears.  Q < 0: run table from
// the arguments (any radius up to 20 cells, tie radii).
template <int R, int Q>
__global__ __launch_bounds__(kLanes, kFpWaves) void k_fp_slide(Geo g, SpiralArgs a, const float* __restrict__ trav,
                                                     const uint8_t* __restrict__ untrav,
                                                     float* __restrict__ footprint) {
  constexpr int W = kLanes + 2 * R;
  constexpr int NR = 2 * R + 3;  // rows j-R .. j+2+R: the reads of step j+1 are issued during step j
  constexpr int NX = (W + kLanes - 1) / kLanes;
  constexpr int kAhead = 4;
  constexpr bool kPipe = R <= 10;  // software-pipelined ring reads (needs 8(R+1) more VGPRs)
  __shared__ double ring[NR * W];
  const int lane = threadIdx.x;
  const size_t mo = (size_t)(a.map0 + (int)blockIdx.z) * g.rows * g.cols;
  const int i0 = blockIdx.x * kLanes;
  const int js = blockIdx.y * a.out_rows;
  const int jend = js + a.out_rows < g.cols ? js + a.out_rows : g.cols;
  const int i = i0 + lane;
  const int c = lane + R;
  const int kx = (i >= g.rows) ? 0 : (i < R) ? (R - i) : ((g.rows - 1 - i < R) ? -(R - (g.rows - 1 - i)) : 0);

  for (int idx = lane; idx < NR * W; idx += kLanes) ring[idx] = 0.0;
  __syncthreads();
  // clamped (always valid) addresses; whether the cell exists is decided when it is staged
  auto load_row = [&](int r, float (&pt)[NX], int (&pu)[NX]) {
    const int rc = r < 0 ? 0 : (r >= g.cols ? g.cols - 1 : r);
    const size_t base = mo + (size_t)rc * g.rows;
#pragma unroll
    for (int x = 0; x < NX; ++x) {
      int ci = i0 - R + lane + x * kLanes;
      ci = ci < 0 ? 0 : (ci >= g.rows ? g.rows - 1 : ci);
      pt[x] = trav[base + ci];
      pu[x] = untrav[base + ci];
    }
  };
  auto store_row = [&](int r, int slot, const float (&pt)[NX], const int (&pu)[NX]) {
    const bool rin = r >= 0 && r < g.cols;
    double* dst = ring + slot * W;
    double v0 = 0.0;
#pragma unroll
    for (int x = 0; x < NX; ++x) {
      const int cc = lane + x * kLanes;
      const int ci = i0 - R + cc;
      const bool inmap = rin && cc < W && ci >= 0 && ci < g.rows;
      const double t = __builtin_isfinite(pt[x]) ? (double)pt[x] : a.def;  // :719-724
      const double v = inmap ? t + (pu[x] ? kUOff : 0.0) : 0.0;             // cells outside the map: nothing
      if (x == 0) v0 = v;
      dst[cc < W ? cc : lane] = cc < W ? v : v0;
    }
  };

  // the spiral table, entry ch * 64 + lane in register ch of lane `lane` (the wavefront-wide walk below)
  constexpr bool kRegTab = R <= 16;
  constexpr int NTAB = kRegTab ? (int)(3.2 * (R + 1) * (R + 1) / kLanes) + 1 : 1;  // >= cells of a disc of radius R + 1
  unsigned tabreg[NTAB];
  if (kRegTab) {
    const unsigned* __restrict__ ptab0 = reinterpret_cast<const unsigned*>(a.table + 4 * kMaxSpiral);
#pragma unroll
    for (int ch = 0; ch < NTAB; ++ch) tabreg[ch] = ch * kLanes + lane < a.n_spiral ? ptab0[ch * kLanes + lane] : 0u;
  }

  double S = 0.0;
  // The strip starts with its first disc summed directly (rows js-R .. js+1+R staged first): sliding in from an empty
  // disc cost 2R+1 full steps per strip, which on a small map is most of the launch.
  const int jstart = js;
  int slot_j = 0;
  // ring offsets (in doubles) of the leading (j+1+h) / trailing (j-h) row of disc column |di| = d for the
  // NEXT fetch; a column outside the tie-free disc (h < 0: only tie offsets reach it) reads the same row
  // twice, so it contributes exactly 0
  constexpr bool kStatic = Q >= 0;
  int lead[R + 1], trail[R + 1];
  int rowoff[NR];  // kStatic: ring offset (in doubles) of row j-R+k of the step the next fetch belongs to
  if (kStatic) {
#pragma unroll
    for (int k = 0; k < NR; ++k) rowoff[k] = ((k - R + NR) % NR) * W;
  } else {
#pragma unroll
    for (int d = 0; d <= R; ++d) {
      const int h = a.h[d];
      lead[d] = h >= 0 ? ((1 + h) % NR) * W : 0;
      trail[d] = h >= 0 ? ((NR - h) % NR) * W : 0;
    }
  }
  float ptq[kAhead][NX];
  int puq[kAhead][NX];
#pragma unroll
  for (int d = 0; d < kAhead; ++d) load_row(jstart + 2 + R + d, ptq[d], puq[d]);
  // Software pipeline: step j (disc row j -> j+1) consumes the 4(R+1)-2 ring values that were read during
  // step j-1 and, before that, issues the reads of step j+1 -- the LDS latency is covered by the adds.
  struct Vals {
    double lp[R + 1], lm[R + 1], tp[R + 1], tm[R + 1];
  };
  auto fetch = [&](Vals& v) {
    if constexpr (kStatic) {
      fast::static_for<R + 1>([&](auto dc) __attribute__((always_inline)) {
        constexpr int d = decltype(dc)::value;
        constexpr int h = fast::Shape<Q>::hw(d);
        const double* rl = ring + rowoff[R + 1 + h] + c;  // row j+1+h
        const double* rt = ring + rowoff[R - h] + c;      // row j-h
        v.lp[d] = rl[d];
        v.tp[d] = rt[d];
        v.lm[d] = d ? rl[-d] : 0.0;
        v.tm[d] = d ? rt[-d] : 0.0;
      });
    } else {
#pragma unroll
      for (int d = 0; d <= R; ++d) {
        const double* rl = ring + lead[d] + c;
        const double* rt = ring + trail[d] + c;
        v.lp[d] = rl[d];
        v.tp[d] = rt[d];
        v.lm[d] = d ? rl[-d] : 0.0;
        v.tm[d] = d ? rt[-d] : 0.0;
        lead[d] = lead[d] + W >= NR * W ? 0 : lead[d] + W;
        trail[d] = trail[d] + W >= NR * W ? 0 : trail[d] + W;
      }
    }
  };
  auto consume = [&](const Vals& v) {
    double acc = v.lp[0] - v.tp[0];
#pragma unroll
    for (int d = 1; d <= R; ++d) acc += (v.lp[d] - v.tp[d]) + (v.lm[d] - v.tm[d]);
    S += acc;  // all terms are exact, so is the order
  };
  // step j: stage row j+2+R, refill the load queue, read for step j+1, slide with the values read earlier
  auto advance = [&](int j, float (&pt)[NX], int (&pu)[NX], Vals& next, const Vals& cur) {
    int sr = slot_j + 2 + R;
    sr = sr >= NR ? sr - NR : sr;
    store_row(j + 2 + R, sr, pt, pu);
    load_row(j + 2 + R + kAhead, pt, pu);
    if constexpr (kStatic) {  // the table now describes step j+1: the oldest row's slot is the newest row's
      const int oldest = rowoff[0];
      fast::static_for<NR - 1>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        rowoff[k] = rowoff[k + 1];
      });
      rowoff[NR - 1] = oldest;
    }
    if (kPipe) {
      fetch(next);
      consume(cur);
    } else {  // large discs: two value buffers do not fit the register file, read and add in the same step
      fetch(next);
      consume(next);
    }
    slot_j = slot_j + 1 >= NR ? 0 : slot_j + 1;
  };
  Vals va, vb;
  {  // rows js-R .. js+1+R into the slots (row - js) mod NR (slot_j == 0 belongs to row js), kAhead loads in flight
    float pt0[kAhead][NX];
    int pu0[kAhead][NX];
#pragma unroll 1
    for (int b = 0; b < 2 * R + 2; b += kAhead) {
#pragma unroll
      for (int k = 0; k < kAhead; ++k)
        if (b + k < 2 * R + 2) load_row(js - R + b + k, pt0[k], pu0[k]);
#pragma unroll
      for (int k = 0; k < kAhead; ++k)
        if (b + k < 2 * R + 2) {
          const int sl = b + k - R;
          store_row(js - R + b + k, sl < 0 ? sl + NR : sl, pt0[k], pu0[k]);
        }
    }
    // the tie-free disc of row js, column by column (every term is exact: the same number the slide would reach)
    auto add_column = [&](int d, int h) __attribute__((always_inline)) {
      if (h < 0) return;
      double col = 0.0;
      const int dm = d ? -d : d;  // column 0 once
#pragma unroll 8
      for (int dj = -h; dj <= h; ++dj) {  // (unrolled: the reads of eight rows in flight, not one round trip per cell)
        const double* row = ring + (dj < 0 ? dj + NR : dj) * W + c;
        col += row[d];
        col += d ? row[dm] : 0.0;
      }
      S += col;
    };
    if constexpr (kStatic) {
      fast::static_for<R + 1>([&](auto dc) __attribute__((always_inline)) {
        constexpr int d = decltype(dc)::value;
        add_column(d, fast::Shape<Q>::hw(d));
      });
    } else {
      for (int d = 0; d <= R; ++d) add_column(d, a.h[d]);
    }
    if (kPipe) fetch(va);
  }
  static_assert(kAhead % 2 == 0, "the value buffers alternate with the queue slots");

  int nt_next = a.gtab[(((js < R) ? (R - js) : ((g.cols - 1 - js < R) ? -(R - (g.cols - 1 - js)) : 0)) + R) * (2 * R + 1) * 6 +
                       (kx + R) * 6];
  int nt_cached = 0;
  double rnt = 0.0;
#pragma unroll 1
  for (int j0 = js; j0 < jend; j0 += kAhead) {
  // (a lambda per queue slot: a rolled loop would index the load queue dynamically and put it into scratch)
  fast::static_for<kAhead>([&](auto qc) __attribute__((always_inline)) {
    constexpr int qs = decltype(qc)::value;
    const int j = j0 + qs;
    if (j >= jend) return;
    double St = S;
    int nt = nt_next;
    if (g.rows < 2 * R + 1 || g.cols < 2 * R + 1) {
      // map narrower than the disc: both borders can clip it at once, which the one-sided clip codes of the
      // table cannot express -- count the cells of the clipped tie-free runs directly (tiny maps only)
      nt = 0;
      for (int dj = -R; dj <= R; ++dj) {
        const int hw = a.h[dj < 0 ? -dj : dj];
        if (hw < 0 || j + dj < 0 || j + dj >= g.cols) continue;
        const int lo = i - hw > 0 ? i - hw : 0, hi = i + hw < g.rows - 1 ? i + hw : g.rows - 1;
        nt += hi >= lo ? hi - lo + 1 : 0;
      }
    }
    {  // cell count of the clipped tie-free disc for the next row (only changes near the map border)
      const int jn = j + 1 < g.cols ? j + 1 : j;
      const int kyn = (jn < R) ? (R - jn) : ((g.cols - 1 - jn < R) ? -(R - (g.cols - 1 - jn)) : 0);
      nt_next = a.gtab[((kyn + R) * (2 * R + 1) + (kx + R)) * 6];
    }
    // cells on the circle itself (tie radii): SpiralIterator::isInside per cell
    for (int t = 0; t < a.n_ties; ++t) {
      const int di = a.tie_di[t], dj = a.tie_dj[t];
      const int ii = i + di, jj = j + dj;
      if (ii < 0 || ii >= g.rows || jj < 0 || jj >= g.cols) continue;
      const double dx = cell_x(g, ii) - cell_x(g, i), dy = cell_y(g, jj) - cell_y(g, j);
      if (dx * dx + dy * dy <= a.r2) {
        int sl = slot_j + dj;
        sl = sl >= NR ? sl - NR : (sl < 0 ? sl + NR : sl);
        St += ring[sl * W + c + di];
        nt += 1;
      }
    }
    const int Ut = (int)floor((St + 0.5 * kUOff) * (1.0 / kUOff));
    float out = qnanf();
    if (Ut == 0) {
      // :732-735 no untraversable cell in the footprint: mean = St / nt.  nt is constant away from the map
      // border, so the division is a multiplication by the cached RN(1/nt) plus two residual corrections
      // (the second one makes the quotient correctly rounded)
      if (nt != nt_cached) {
        nt_cached = nt;
        rnt = 1.0 / (double)nt;
      }
      const double dn = (double)nt;
      const double q0 = St * rnt;
      const double q1 = fma(fma(-q0, dn, St), rnt, q0);
      out = (float)fma(fma(-q1, dn, St), rnt, q1);
    }
    if (__builtin_expect(__any(Ut != 0), 0)) {
      // walk the spiral until the first untraversable cell :687-717
      const unsigned* __restrict__ ptab = reinterpret_cast<const unsigned*>(a.table + 4 * kMaxSpiral);  // packed entries
      auto slot_of = [&](int dj) __attribute__((always_inline)) {
        const int sl = slot_j + dj;
        return sl >= NR ? sl - NR : (sl < 0 ? sl + NR : sl);
      };
      auto value_at = [&](int ring_no, double t, int ncells) __attribute__((always_inline)) {
        const double ru = (double)ring_no * g.res;  // getCurrentRadius()
        if (a.rmin == 0.0 || ru <= a.rmin) return 0.0f;  // :694-704
        const double factor = ((ru - a.rmin) / (a.rmax - a.rmin) + 1.0) / 2.0;  // :705-711
        t *= factor / ncells;
        return (float)t;
      };
      bool found = Ut == 0;
      // (0) An untraversable cell within the inner radius makes the footprint 0 whatever comes before it (:694-704; the
      // spiral visits the rings in order), so a disc is first searched for one directly: all lanes at once, a few hundred
      // ring reads, no table.  inner_q: largest di^2 + dj^2 of the rings that lie within the inner radius and are taken
      // whole by the SpiralIterator (-1: none).  On a map full of obstacles most discs end here.
      if (!found && a.rmin == 0.0) {
        out = 0.0f;
        found = true;
      }
      if (!found && a.inner_q >= 0) {
        const int dm = (int)__builtin_sqrtf((float)a.inner_q);
        int hits = 0;
        for (int dj = -dm; dj <= dm; ++dj) {
          const int hwi = (int)__builtin_sqrtf((float)(a.inner_q - dj * dj));
          const double* row = ring + slot_of(dj) * W + c;
#pragma unroll 8
          for (int di = -hwi; di <= hwi; ++di) hits += row[di] >= 0.5 * kUOff ? 1 : 0;
        }
        if (hits > 0) {
          out = 0.0f;
          found = true;
        }
      }
      // (1) every lane walks the head of its own spiral: eight table entries per trip, their ring cells fetched
      // together (one entry per trip made the lane wait for a table load and an LDS read in turn).  Pointless after (0).
      if (!found && a.inner_q < 8) {
        double t = 0.0;
        int ncells = 0;
        const int n_head = a.n_spiral < kFpHead ? a.n_spiral : kFpHead;
        for (int k0 = 0; k0 < n_head && !found; k0 += 8) {
          double v[8];
          bool in[8];
          int ring_no[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int kk = k0 + q < n_head ? k0 + q : n_head - 1;
            const unsigned w = ptab[kk];  // uniform: a scalar load
            const int di = (int)(signed char)(w & 0xffu), dj = (int)(signed char)((w >> 8) & 0xffu);
            ring_no[q] = (int)((w >> 16) & 0xffu);
            const int ii = i + di, jj = j + dj;
            in[q] = k0 + q < n_head && ii >= 0 && ii < g.rows && jj >= 0 && jj < g.cols;
            if (in[q] && (w >> 24)) {  // a cell on the circle itself: SpiralIterator::isInside
              const double dx = cell_x(g, ii) - cell_x(g, i), dy = cell_y(g, jj) - cell_y(g, j);
              in[q] = dx * dx + dy * dy <= a.r2;
            }
            v[q] = ring[slot_of(dj) * W + c + (in[q] ? di : 0)];
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            if (found || !in[q]) continue;
            if (v[q] >= 0.5 * kUOff) {
              out = value_at(ring_no[q], t, ncells);
              found = true;
            } else {
              ncells++;
              t += v[q];
            }
          }
        }
      }
      // (2) the discs whose first untraversable cell lies further out, one at a time with the whole wavefront: lane q
      // takes entry 64 ch + q (held in registers since the kernel started for radii up to 16 cells: a table load per
      // chunk was a memory round trip on the critical path), the first untraversable entry comes from a ballot, the sum
      // of the cells before it from one reduction (a lane walking 700 entries on its own kept the other 63 waiting; the
      // reference's own 100 x 133 map at 0.03 m spent 1.1 ms in this kernel)
      unsigned long long rest = __ballot(!found);
      while (rest != 0ull) {
        const int l = __builtin_ctzll(rest);
        rest &= rest - 1ull;
        const int ic = i0 + l;
        double acc = 0.0;
        int cnt = 0;
        float oc = qnanf();
        bool done = false;
        auto chunk = [&](unsigned w, int k0) __attribute__((always_inline)) {
          const bool valid = k0 + lane < a.n_spiral;
          const int di = (int)(signed char)(w & 0xffu), dj = (int)(signed char)((w >> 8) & 0xffu);
          const int ii = ic + di, jj = j + dj;
          bool in = valid && ii >= 0 && ii < g.rows && jj >= 0 && jj < g.cols;
          if (in && (w >> 24)) {  // a cell on the circle itself: SpiralIterator::isInside
            const double dx = cell_x(g, ii) - cell_x(g, ic), dy = cell_y(g, jj) - cell_y(g, j);
            in = dx * dx + dy * dy <= a.r2;
          }
          const double v = ring[slot_of(dj) * W + l + R + di];
          const unsigned long long bm = __ballot(in && v >= 0.5 * kUOff);
          if (bm != 0ull) {
            const int first = __builtin_ctzll(bm);
            const int ring_first = __builtin_amdgcn_readlane((int)((w >> 16) & 0xffu), first);
            done = true;
            if (a.rmin == 0.0 || (double)ring_first * g.res <= a.rmin) {  // :694-704: no sum needed
              oc = 0.0f;
              return;
            }
            const bool before = in && lane < first;
            acc += before ? v : 0.0;
            cnt += __popcll(__ballot(before));
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);  // exact terms: any order
            oc = value_at(ring_first, acc, cnt);
            return;
          }
          acc += in ? v : 0.0;
          cnt += __popcll(__ballot(in));
        };
        if (kRegTab && a.n_spiral <= NTAB * kLanes) {
          fast::static_for<NTAB>([&](auto chc) __attribute__((always_inline)) {
            constexpr int ch = decltype(chc)::value;
            if (done || ch * kLanes >= a.n_spiral) return;  // uniform
            chunk(tabreg[ch], ch * kLanes);
          });
        } else {
          for (int k0 = 0; k0 < a.n_spiral && !done; k0 += kLanes) chunk(ptab[k0 + lane < a.n_spiral ? k0 + lane : 0], k0);
        }
        if (lane == l) out = oc;  // an untraversable cell is in the disc (Ut > 0), so oc was set
      }
    }
    if (qs % 2 == 0)
      advance(j, ptq[qs], puq[qs], vb, va);
    else
      advance(j, ptq[qs], puq[qs], va, vb);
    if (i < g.rows) footprint[mo + (size_t)j * g.rows + i] = out;
  });
  }
}

}  // namespace

hipError_t launch_footprint(const Geo& g, const FootprintParams& p, const Layers& L, const int16_t* spiral_table,
                            const int* clip_table, bool write_memo, const ChainParams* combine, double trav_cap, hipStream_t stream,
                            const Region* region) {
  MaskArgs m;
  m.slope_disc = p.slope_disc;
  m.step_disc = p.step_disc;
  m.ncrit_slope = p.ncrit_slope;
  m.ncrit_rough = p.ncrit_rough;
  m.check_rough = p.check_rough;
  m.write_memo = write_memo ? 1 : 0;
  m.crit_step = p.crit_step;
  m.max_gap = p.max_gap;
  m.combine = combine ? 1 : 0;
  m.edge_fail = submap_edge_failures(g);
  m.w_scale = combine ? combine->w_scale : 0.0f;
  m.w_slope = combine ? combine->w_slope : 0.0f;
  m.w_step = combine ? combine->w_step : 0.0f;
  m.w_rough = combine ? combine->w_rough : 0.0f;
  m.ti0 = m.tj0 = 0;
  m.map = -1;
  m.blocked_count = L.fp_blocked_count;
  m.untrav_flags = L.untrav_flags;
  m.flag_ntx = untrav_flag_ntx(g.rows);
  m.flag_nfy = untrav_flag_nfy(g.cols);
  static const int mask_whatif = lab_int("TE_MASK_WHATIF", 0);  // (timing experiments: wrong results by construction)
  m.whatif = mask_whatif;
  // A region run (te_run_chain_region with the footprint flag): isTraversableForFilters of a cell reads scores within
  // 3 cells (circle(3 res), circle(2.5 res) and the 3x3 blocks around its cells), so the mask is recomputed on the
  // region grown by MH; the footprint of a cell reads the mask and the traversability within the footprint's reach.
  Region rm = {-1, 0, 0, g.rows, g.cols}, rf = rm;
  if (region) {
    auto grow = [&](const Region& r, int k) {
      Region o = r;
      o.i0 = r.i0 - k < 0 ? 0 : r.i0 - k;
      o.j0 = r.j0 - k < 0 ? 0 : r.j0 - k;
      o.i1 = r.i1 + k > g.rows ? g.rows : r.i1 + k;
      o.j1 = r.j1 + k > g.cols ? g.cols : r.j1 + k;
      return o;
    };
    rm = grow(*region, MH);
    rf = grow(rm, p.reach);
    m.map = region->map;
  }
  const long tiles32 = (long)((rm.i1 - rm.i0 + MX - 1) / MX) * ((rm.j1 - rm.j0 + 31) / 32) * (region ? 1 : (g.batch > 0 ? g.batch : 1));
  static const int small_env = lab_int("TE_MASK_SMALL_TILES", -1);
  const bool small = small_env >= 0 ? small_env != 0 : tiles32 < 1024;  // fewer than 4 workgroups per CU
  static const int tile_env = lab_int("TE_MASK_TILE", 0);  // measurement aid: 16-row tiles on large maps (8 blocks per CU instead of 5)
  const int my = (small && tiles32 < 128 && small_env != 8) ? 4 : (small ? 8 : (tile_env == 16 ? 16 : 32));  // a very small map: one cell per thread
  // the mask kernel on the tile rows [t0, t1) (tiles of my cells) of the region
  auto launch_mask = [&](int t0, int t1, hipStream_t st) {
    if (t1 <= t0) return;
    m.ti0 = rm.i0 / MX;
    m.tj0 = t0;
    const dim3 grid((unsigned)((rm.i1 - 1) / MX - m.ti0 + 1), (unsigned)(t1 - t0), (unsigned)(region ? 1 : g.batch));
    if (my == 4)
      hipLaunchKernelGGL(k_fp_mask<4>, grid, dim3(MX, MBY), 0, st, g, m, L.elev, L.slope, L.step, L.rough, L.untrav, L.slope_fp, L.step_fp, L.rough_fp, L.trav);
    else if (my == 8)
      hipLaunchKernelGGL(k_fp_mask<8>, grid, dim3(MX, MBY), 0, st, g, m, L.elev, L.slope, L.step, L.rough, L.untrav, L.slope_fp, L.step_fp, L.rough_fp, L.trav);
#ifdef TE_LAB
    else if (my == 16)
      hipLaunchKernelGGL(k_fp_mask<16>, grid, dim3(MX, MBY), 0, st, g, m, L.elev, L.slope, L.step, L.rough, L.untrav, L.slope_fp, L.step_fp, L.rough_fp, L.trav);
#endif
    else
      hipLaunchKernelGGL(k_fp_mask<32>, grid, dim3(MX, MBY), 0, st, g, m, L.elev, L.slope, L.step, L.rough, L.untrav, L.slope_fp, L.step_fp, L.rough_fp, L.trav);
  };
  const int t_lo = rm.j0 / my, t_hi = (rm.j1 - 1) / my + 1;
  // (A pass in two bands -- the upper half's sliding sum on the second stream beside the lower half's mask kernel -- was
  // measured in round 3 with k_fp_slide4 (0.418 ms per launch against 0.385) and again in round 4 with k_fp_slide5 and
  // 2 / 3 / 4 / 6 bands on two streams (0.393 / 0.425 / 0.462 / 0.589 against 0.375, profiles/r04_experiments.json): the two
  // kernels slow each other down by more than they overlap, and every band pays the strips' 2R lead-in rows again.)
  // (Round 5 what-ifs, recorded in profiles/r05_experiments.json and kept in the tree of git tag round5-record as tools/lab/attic/r05_whatif_footprint.patch:
  // the sum kernel on the second stream BESIDE the mask kernel -- 2-3 % slower than behind it -- and ONE kernel that stages
  // elevation and the three scores itself: 264 us alone against 67 + 51.)
  {
    TraceRange tr(combine ? "footprint: isTraversableForFilters mask + weighted combine" : "footprint: isTraversableForFilters mask");
    launch_mask(t_lo, t_hi, stream);
  }
  TraceRange tr_sum("footprint: disc sums (+ blocked discs)");
  const Region* rfp = region ? &rf : nullptr;
  SpiralArgs a;
  const Disc& d = p.fp_disc;
  for (int k = 0; k <= kMaxRadiusCells; ++k) a.h[k] = (k <= d.R) ? d.hw[k] : -1;
  a.n_ties = d.n_ties;
  for (int t = 0; t < kMaxTies; ++t) {
    a.tie_di[t] = d.tie_di[t];
    a.tie_dj[t] = d.tie_dj[t];
  }
  a.r2 = d.r2;
  a.n_spiral = p.n_spiral;
  a.table = spiral_table;
  a.gtab = clip_table;
  a.rmin = p.rmin;
  a.rmax = p.rmax;
  a.def = p.def;
  a.inner_q = fast::footprint_inner_q(g.res, p.rmin, p.rmax);
  // tie-free disc of an instantiated shape on a map at least one block wide: the scatter-form sum on fixed point, ...
  {
    bool needs_blocked = false;
    if (fast::footprint_slide5(g, p, L, clip_table, trav_cap, stream, rfp, &needs_blocked)) {
      if (needs_blocked) fast::footprint_blocked4(g, p, L, spiral_table, stream);
      return hipGetLastError();
    }
  }
  // ... the sliding sum on fixed point (tie radii; radii of 16 cells), the sliding sum in double (unbounded layers)
  if (fast::footprint_slide4(g, p, L, spiral_table, clip_table, trav_cap, stream, rfp) ||
      fast::footprint_slide3(g, p, L, spiral_table, clip_table, stream, rfp))
    return hipGetLastError();
  // (the kernel below always covers every cell of a map: a region run falls back to it for the region's map -- the cells
  // outside the region are recomputed from unchanged inputs.  Where an earlier whole-map pass wrote them with the
  // fixed-point kernels they move by that kernel's rounding, below 1e-6; include/travgpu.h says so at te_run_chain_region)
  {  // one round of resident waves (kFpWaves per SIMD): as many strips as fit
    const int nbx = (g.rows + kLanes - 1) / kLanes;
    const int Rk = p.reach;
    const long ring_bytes = (long)(2 * Rk + 3) * (kLanes + 2 * Rk) * 8;
    long per_cu = 160 * 1024 / ring_bytes;  // blocks (= waves) per CU the LDS allows
    if (per_cu > kFpWaves * 4) per_cu = kFpWaves * 4;
    int strips = (int)((per_cu * 256) / (nbx * (region ? 1 : (g.batch > 0 ? g.batch : 1))));
    strips = strips < 1 ? 1 : strips;
    int rows_per = (g.cols + strips - 1) / strips;
    // (a small map cannot fill the wave slots anyway: every block is resident at once and the launch takes one warm-up
    // plus the rows of one strip, so the shortest strips win -- each wave's spiral walks are serial)
    static const int min_rows = lab_int("TE_FP_MIN_STRIP", 1);
    a.out_rows = rows_per < min_rows ? min_rows : (rows_per > 512 ? 512 : rows_per);
    a.out_rows = a.out_rows < 1 ? 1 : a.out_rows;
  }
  a.map0 = region ? region->map : 0;
  const dim3 grid((unsigned)((g.rows + kLanes - 1) / kLanes), (unsigned)((g.cols + a.out_rows - 1) / a.out_rows),
                  (unsigned)(region ? 1 : g.batch));
  // (the compile-time run tables of this kernel, Q >= 0, are no longer instantiated: tie-free discs of the instantiated
  // shapes go to k_fp_slide3 above unless the map is narrower than a wavefront, where speed is not a concern)
  switch (p.reach) {
#define X(q) \
  case q:    \
    hipLaunchKernelGGL((k_fp_slide<q, -1>), grid, dim3(kLanes), 0, stream, g, a, L.trav, L.untrav, L.footprint); \
    break;
    X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20)
#undef X
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace te
