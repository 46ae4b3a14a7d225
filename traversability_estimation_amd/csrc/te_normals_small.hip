// te_normals_small.hip -- NormalVectorsFilter + SlopeFilter + RoughnessFilter for discs that reach at most TWO cells: one
// cell per thread, the neighbourhood gathered directly.
//
//   NormalVectorsFilter (area method; un-vendored grid_map_filters, call site
//                        traversability_estimation/config/robot_filter_parameter.yaml:3-9)
//   SlopeFilter::update      traversability_estimation_filters/src/SlopeFilter.cpp:59-88
//   RoughnessFilter::update  traversability_estimation_filters/src/RoughnessFilter.cpp:73-132
//
// Why a kernel of its own.  The reference's DEFAULT radius is 0.05 m (robot_filter_parameter.yaml:8, :28): on a 0.05 m map
// that is a TIE radius of exactly one cell -- the runs of the disc hold the centre alone and the four edge neighbours lie
// on the circle, kept or dropped centre by centre by the rounding of the cell positions (CircleIterator::isInside).  The
// sliding kernels of te_normals3.hip exist to make a 253-point disc cost 38 cells per step; with at most 12 neighbours
// there is nothing to slide, and their strips (a serial march per wavefront, a ring, strip starts, the map frame left to the
// fix-up pass) are pure overhead: the TIES march took 0.47 ms for a 4096^2 map at one cell, 31 us for a 256^2 one.
// Here every thread owns one cell (workgroups of 64 x 4), reads its neighbours straight from the layer
// (coalesced along i, the rows above and below come from L2 / the vector cache), decides its tie cells with the
// reference's own position arithmetic, and runs the general tail of te_eig3.h.  Discs clipped by the map border and
// invalid neighbours are just fewer points: no frame, no hole march.  Cells the fast tail does not resolve (a nearly
// horizontal normal, an ambiguous middle eigenvalue) and scores within their error of the clip at 0 (te_internal.h) take the
// generic arithmetic in place: no fix-up pass behind this kernel.
//
// Algorithmic bytes: 4 B read + 8 B written per cell (+ 12 B when the normals are kept).
#include "te_cell.h"
#include "te_eig.h"
#include "te_eig3.h"
#include "te_geom.h"
#include "te_internal.h"
#include "te_march.h"

namespace te {
namespace fast {

namespace {

constexpr int kSmallMaxOffsets = 12;  // the 13-point disc (reach 2) without its centre
constexpr int kSmallBY = 4;           // rows of a workgroup (64 x 4 threads, one cell each)

struct SmallArgs {
  const float* elev;
  float* slope;
  float* rough;
  float* nx;
  float* ny;
  float* nz;
  int rows, cols;
  long long map_cells;
  int map;  // >= 0: this map only; < 0: blockIdx.z
  int i_lo, i_hi, j_lo, j_hi;
  double res, r2, ax, ay;
  int n_off;
  signed char di[kSmallMaxOffsets], dj[kSmallMaxOffsets];
  unsigned tie_mask;  // bit k: offset k lies on the circle -- isInside decides it for every centre
  float inv_slope_crit, inv_rough_crit, band_slope, band_rough;
  // SINGLE-CELL STEP WINDOWS (both radii below one cell: the default 0.04 m on any map coarser than 0.04 m).  StepFilter then
  // needs no neighbour: pass 1 gives max - min = 0 for a valid centre (StepFilter.cpp:113-143), pass 2 finds stepMax = 0 and
  // nCells = 0 (0 > critical is false for any critical >= 0, :165), so step = 0 and the score is 1 - 0 / critical = 1
  // (critical > 0; 0 otherwise, "0 < 0" fails at :172) -- and NaN where the elevation is invalid (no valid step height in the
  // window, :161).  The kernel that holds the centre's elevation anyway writes the layer, and with it the weighted sum
  // (MathExpressionFilter, float32, left to right): the chain of such a map is this kernel and its fix-up pass.
  float* step;
  float* trav;
  float* step_height;  // k_chain_window: StepFilter's temp layer is kept complete -- a later region run reads it around its region
  int write_step, combine;
  double slope_crit, rough_crit;  // (the exact tail of the cells the fast one does not settle)
  float step_valid;  // the step score of a valid cell: 1, or 0 for a critical value of 0
  // k_chain_window: the 5 x 5 window around a cell, bit (dj + 2) * 5 + (di + 2) -- the cells of the normals disc, those of
  // them that lie on its circle (isInside decides per centre), and the two step windows as 3 x 3 masks, bit (dj + 1) * 3 + (di + 1)
  unsigned disc25, tie25, need25, win1_9, win2_9;
  double step_crit;
  float step_crit_lo;  // largest float <= step_crit
  int step_ncrit;
  float w_scale, w_slope, w_step, w_rough;
};

// The valid cells of a disc all lie on ONE grid line (with a one-cell tie radius: the centre and its two neighbours along
// one axis, the pair across the other axis rejected by isInside -- on a 0.05 m map at the origin a third of all rows): the
// covariance has the exact eigenvalue 0 with the horizontal unit vector across the line as eigenvector.  general_tail3
// leaves a (nearly) horizontal normal unresolved, and in round 6's first measurement these cells -- a fifth of the map --
// made the fix-up pass the longest kernel of the launch (0.35 of 0.48 ms on 4096^2).  They need no eigen-solver:
//   * the normal is that horizontal vector if the middle eigenvalue -- the smaller one of the 2 x 2 (line coordinate, z)
//     block -- is > 1e-8 (NormalVectorsFilter), else UnitZ; decided from the sign of (A - t)(F - t) - D^2, t = 1e-8 n^2;
//   * horizontal: nz = 0, slope pi/2, n^T C n = 0, roughness 0 (the SIGN of such a normal is rounding noise in the
//     reference itself, tests/helpers.py: orient_horizontal_normals); UnitZ: n^T C n = var(z).
// Returns 0: done; 1: not this case's business after all (a diagonal line, the threshold within its rounding): unresolved.
__device__ __forceinline__ int collinear_tail(double res, int n, long long Ai, long long Ci, int si, int sj, double Sz, double Siz, double Sjz, double Szz,
                                              float& nx, float& ny, float& nz, double& q_scaled) {
  const double dn = (double)n;
  const double F = fma(dn, Szz, -(Sz * Sz));  // n^2 var(z)
  const bool along_i = Ci == 0;               // every point in one map row j: the line runs along i (x)
  if (!along_i && Ai != 0) return 1;          // a diagonal line (13-point discs with many invalid cells): the fix-up pass
  const double Au = res * res * (double)(along_i ? Ai : Ci);
  const double Du = -res * (along_i ? fma(dn, Siz, -((double)si * Sz)) : fma(dn, Sjz, -((double)sj * Sz)));
  const double thr = 1e-8 * dn * dn;
  const double P = fma(Au - thr, F - thr, -(Du * Du));
  if (fabs(P) <= 1e-6 * Au * thr) return 1;
  if (P > 0.0 && F > thr) {
    nx = along_i ? 0.0f : 1.0f;
    ny = along_i ? 1.0f : 0.0f;
    nz = 0.0f;
    q_scaled = 0.0;
  } else {
    nx = ny = 0.0f;
    nz = 1.0f;
    q_scaled = F > 0.0 ? F : 0.0;
  }
  return 0;
}

// The tail of one cell from its moments (centre-local coordinates): normal, slope score, roughness score.  The fast tails
// (collinear_tail, general_tail3) settle nearly every cell; what they leave open takes the generic arithmetic in place.
__device__ __forceinline__ void small_tail(const SmallArgs& a, int n, int si, int sj, int sii, int sij, int sjj, double Sz, double Siz, double Sjz,
                                           double Szz, float& o_slope, float& o_rough, float& fx, float& fy, float& fz) {
  double qs = 0.0;
  int unresolved;
  const long long Ai = (long long)n * sii - (long long)si * si, Bi = (long long)n * sij - (long long)si * sj, Ci = (long long)n * sjj - (long long)sj * sj;
  if (n >= 3 && Ai * Ci - Bi * Bi == 0)
    unresolved = collinear_tail(a.res, n, Ai, Ci, si, sj, Sz, Siz, Sjz, Szz, fx, fy, fz, qs);
  else
    unresolved = general_tail3<true>(a.res, n, si, sj, sii, sij, sjj, Sz, Siz, Sjz, Szz, fx, fy, fz, qs);
  // slope = acos(float32 nz) (SlopeFilter.cpp:74); roughness^2 = q / (n (n - 1)) (RoughnessFilter.cpp:105-117)
  const float sl = acosf_poly(fz);
  const float rs = fmaf(-sl, a.inv_slope_crit, 1.0f);
  o_slope = fmaxf(rs, 0.0f);
  float rq = (float)(qs * rcp_fast((double)n * (double)(n > 1 ? n - 1 : 1)));
  rq = rq > 0.0f ? rq : 0.0f;
  const float rgh = __builtin_amdgcn_sqrtf(rq);
  const float rr = fmaf(-rgh, a.inv_rough_crit, 1.0f);
  o_rough = n > 1 ? fmaxf(rr, 0.0f) : 0.0f;  // n == 1: 0/0 -> "roughness < crit" false -> 0
  const bool near = near_clip(rs, a.band_slope) || (n > 1 && near_clip(rr, a.band_rough));
  if (unresolved != 0 || near) {
    // What the fast tail does not settle -- a nearly horizontal normal, an ambiguous middle eigenvalue, a score within its
    // error of the clip at 0 -- is settled HERE with the generic arithmetic (te_cell.h: cyclic Jacobi, double acos and
    // square root, as k_normals_fixup would): the moments are at hand, the cells are rare (a handful per map on terrain,
    // box edges on maps with steps), and the fix-up pass -- 4 us of launch, flag scan and drain, half of the chain on the
    // maps the reference's node actually filters (4 m x 4 m: 80 x 80 cells) -- is not launched behind this kernel at all.
    Mom m;
    m.n = n; m.si = si; m.sj = sj; m.sii = sii; m.sij = sij; m.sjj = sjj;
    m.sz = Sz; m.siz = Siz; m.sjz = Sjz; m.szz = Szz;
    double cov[6];
    float nf[3];
    covariance(m, a.res, cov);
    normal_from_cov(m, cov, 2, nf);
    o_slope = slope_score(nf[2], a.slope_crit);
    o_rough = roughness_score(m, cov, nf, a.rough_crit);
    fx = nf[0];
    fy = nf[1];
    fz = nf[2];
  }
}

// CROSS: the disc is the centre and its four edge neighbours, all four ON the circle (a one-cell tie radius: the default
// 0.05 m on a 0.05 m map) -- the offsets are constants, the sums lose their zero terms and their int -> double conversions
template <bool KEEP, bool CROSS>
__global__ __launch_bounds__(kLanes* kSmallBY) void k_normals_small(SmallArgs a) {
  constexpr int NOFF = CROSS ? 4 : kSmallMaxOffsets;
  constexpr int cdi[4] = {1, -1, 0, 0}, cdj[4] = {0, 0, 1, -1};
  const int lane = (int)threadIdx.x, ty = (int)threadIdx.y;
  const int mz = a.map >= 0 ? 0 : (int)blockIdx.z;
  const size_t mo = (size_t)(a.map >= 0 ? a.map : mz) * (size_t)a.map_cells;
  const int i = a.i_lo + (int)blockIdx.x * kLanes + lane;
  const int j = a.j_lo + (int)blockIdx.y * kSmallBY + ty;
  if (i < a.i_hi && j < a.j_hi) {
    const size_t o = mo + (size_t)j * a.rows + i;
    const float zcf = a.elev[o];
    // every neighbour's load is issued before the first is used (a loop over n_off waits for one load per turn: on a
    // 256^2 map the kernel then took 18 us against 11 for the sliding kernel it replaces)
    const double xi = a.ax + a.res * (double)(-i), yj = a.ay + a.res * (double)(-j);  // cell_x, cell_y (te_geom.h)
    float zn[NOFF];
#pragma unroll
    for (int k = 0; k < NOFF; ++k) {
      zn[k] = qnan();
      if (CROSS || k < a.n_off) {  // (uniform)
        const int odi = CROSS ? cdi[k & 3] : (int)a.di[k], odj = CROSS ? cdj[k & 3] : (int)a.dj[k];
        const int ii = i + odi, jj = j + odj;
        bool in = (unsigned)ii < (unsigned)a.rows && (unsigned)jj < (unsigned)a.cols;
        if (CROSS || ((a.tie_mask >> k) & 1u)) {  // CircleIterator::isInside with the reference's rounded positions
          // (a cell on an axis: its other coordinate is the centre's own, that difference and its square are 0 exactly and
          // x + 0 = x -- the term is left out, the result is the same bit for bit; uniform branches)
          double sq;
          if (odi == 0) {
            const double dy = (a.ay + a.res * (double)(-jj)) - yj;
            sq = dy * dy;
          } else if (odj == 0) {
            const double dx = (a.ax + a.res * (double)(-ii)) - xi;
            sq = dx * dx;
          } else {
            const double dx = (a.ax + a.res * (double)(-ii)) - xi, dy = (a.ay + a.res * (double)(-jj)) - yj;
            sq = dx * dx + dy * dy;
          }
          in = in && (sq <= a.r2);
        }
        if (in) zn[k] = a.elev[mo + (size_t)jj * a.rows + ii];
      }
    }
    float o_slope = qnan(), o_rough = qnan(), fx = qnan(), fy = qnan(), fz = qnan();
    if (__builtin_isfinite(zcf)) {  // normals only where the input layer is valid; slope / roughness follow (SlopeFilter.cpp:71, RoughnessFilter.cpp:84)
      const double zc = (double)zcf;
      // moments in centre-local coordinates (te_cell.h): the centre itself is (0, 0, 0)
      int n = 1, si = 0, sj = 0, sii = 0, sij = 0, sjj = 0;
      double Sz = 0.0, Siz = 0.0, Sjz = 0.0, Szz = 0.0;
#pragma unroll
      for (int k = 0; k < NOFF; ++k) {
        if (CROSS || k < a.n_off) {
          const int di = CROSS ? cdi[k & 3] : (int)a.di[k], dj = CROSS ? cdj[k & 3] : (int)a.dj[k];
          const bool v = __builtin_isfinite(zn[k]);
          const double dz = v ? (double)zn[k] - zc : 0.0;
          const int w = v ? 1 : 0;
          n += w;
          si += w * di;
          sj += w * dj;
          sii += w * di * di;
          sij += w * di * dj;
          sjj += w * dj * dj;
          Sz += dz;
          if constexpr (CROSS) {  // offsets of +-1 and 0: fma(+-1, dz, S) = S +- dz, fma(0, dz, S) = S (dz is finite)
            if (di == 1) Siz += dz;
            if (di == -1) Siz -= dz;
            if (dj == 1) Sjz += dz;
            if (dj == -1) Sjz -= dz;
          } else {
            Siz = fma((double)di, dz, Siz);
            Sjz = fma((double)dj, dz, Sjz);
          }
          Szz = fma(dz, dz, Szz);
        }
      }
      small_tail(a, n, si, sj, sii, sij, sjj, Sz, Siz, Sjz, Szz, o_slope, o_rough, fx, fy, fz);
    }
    a.slope[o] = o_slope;
    a.rough[o] = o_rough;
    if (a.write_step || a.combine) {  // (uniform)
      float st;
      if (a.write_step) {
        st = __builtin_isfinite(zcf) ? a.step_valid : qnan();
        a.step[o] = st;
      } else {
        st = a.step[o];
      }
      if (a.combine) {  // a cell left to the fix-up pass is NaN here and combined again there (NormalsArgs::combine)
        const float ta = a.w_slope * o_slope, tb = a.w_step * st, tc = a.w_rough * o_rough;
        const float tab = ta + tb;
        const float tabc = tab + tc;
        a.trav[o] = a.w_scale * tabc;
      }
    }
    if (KEEP) {
      a.nx[o] = fx;
      a.ny[o] = fy;
      a.nz[o] = fz;
    }
  }
}

// THE WHOLE CHAIN OF A SMALL MAP IN ONE KERNEL.  On a small launch every kernel is its launch latency (the reference's bag
// map: normals 6 + step 4 + 4 + combine 2 us), and with discs and windows this small one thread can hold everything a cell
// needs: the 5 x 5 window of elevations around it (24 loads, all in flight together).  From it: the normals disc (reach <=
// 2, tie cells decided per centre) -> normal, slope, roughness as k_normals_small; StepFilter's first pass at the up to nine
// cells of the second window (each a max - min over its own window of up to nine cells: windows of at most 3 x 3, tie-free),
// the second pass over them (StepFilter.cpp:112-178); the weighted sum.  A cell outside the map is NaN in the window, which
// is what "not in the window" means to every stage.  Launches of at most 2^18 cells (beyond that the gathers cost more
// than the marching kernels: te_fast_step.hip).
template <bool KEEP>
__global__ __launch_bounds__(kLanes* kSmallBY) void k_chain_window(SmallArgs a) {
  const int lane = (int)threadIdx.x, ty = (int)threadIdx.y;
  const int mz = a.map >= 0 ? 0 : (int)blockIdx.z;
  const size_t mo = (size_t)(a.map >= 0 ? a.map : mz) * (size_t)a.map_cells;
  const int i = a.i_lo + (int)blockIdx.x * kLanes + lane;
  const int j = a.j_lo + (int)blockIdx.y * kSmallBY + ty;
  if (!(i < a.i_hi && j < a.j_hi)) return;
  const size_t o = mo + (size_t)j * a.rows + i;
  float z[25];
  const unsigned need = a.need25;  // the cells any stage reads
  static_for<25>([&](auto kc) __attribute__((always_inline)) {
    constexpr int k = decltype(kc)::value, di = k % 5 - 2, dj = k / 5 - 2;
    z[k] = qnan();
    if ((need >> k) & 1u) {  // (uniform)
      const int ii = i + di, jj = j + dj;
      if ((unsigned)ii < (unsigned)a.rows && (unsigned)jj < (unsigned)a.cols) z[k] = a.elev[mo + (size_t)jj * a.rows + ii];
    }
  });
  const float zcf = z[12];
  float o_slope = qnan(), o_rough = qnan(), fx = qnan(), fy = qnan(), fz = qnan();
  if (__builtin_isfinite(zcf)) {
    const double zc = (double)zcf;
    const double xi = a.ax + a.res * (double)(-i), yj = a.ay + a.res * (double)(-j);  // cell_x, cell_y (te_geom.h)
    int n = 1, si = 0, sj = 0, sii = 0, sij = 0, sjj = 0;
    double Sz = 0.0, Siz = 0.0, Sjz = 0.0, Szz = 0.0;
    static_for<25>([&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value, di = k % 5 - 2, dj = k / 5 - 2;
      if (k != 12 && (((a.disc25 | a.tie25) >> k) & 1u)) {  // (uniform)
        bool in = true;
        if ((a.tie25 >> k) & 1u) {  // CircleIterator::isInside with the reference's rounded positions (axis cells: see k_normals_small)
          if constexpr (di == 0) {
            const double dy = (a.ay + a.res * (double)(-(j + dj))) - yj;
            in = dy * dy <= a.r2;
          } else if constexpr (dj == 0) {
            const double dx = (a.ax + a.res * (double)(-(i + di))) - xi;
            in = dx * dx <= a.r2;
          } else {
            const double dx = (a.ax + a.res * (double)(-(i + di))) - xi, dy = (a.ay + a.res * (double)(-(j + dj))) - yj;
            in = dx * dx + dy * dy <= a.r2;
          }
        }
        const bool v = in && __builtin_isfinite(z[k]);
        const double dz = v ? (double)z[k] - zc : 0.0;
        const int w = v ? 1 : 0;
        n += w;
        si += w * di;
        sj += w * dj;
        sii += w * di * di;
        sij += w * di * dj;
        sjj += w * dj * dj;
        Sz += dz;
        Siz = fma((double)di, dz, Siz);
        Sjz = fma((double)dj, dz, Sjz);
        Szz = fma(dz, dz, Szz);
      }
    });
    small_tail(a, n, si, sj, sii, sij, sjj, Sz, Siz, Sjz, Szz, o_slope, o_rough, fx, fy, fz);
  }
  // StepFilter: the first pass at every cell c of the second window, the second pass over them
  float m = qnan(), sh_own = qnan();
  int count = 0;
  static_for<9>([&](auto cc) __attribute__((always_inline)) {
    constexpr int c = decltype(cc)::value, ci = c % 3 - 1, cj = c / 3 - 1;
    if ((a.win2_9 >> c) & 1u) {  // (uniform)
      const float ec = z[(cj + 2) * 5 + (ci + 2)];
      float vmx = ec, vmn = ec;  // (:113 only cells with a valid elevation get a step height; NaN: not in the map either)
      static_for<9>([&](auto wc) __attribute__((always_inline)) {
        constexpr int w = decltype(wc)::value, wi = w % 3 - 1, wj = w / 3 - 1;
        if (w != 4 && ((a.win1_9 >> w) & 1u)) {
          const float e = z[(cj + wj + 2) * 5 + (ci + wi + 2)];
          if (__builtin_isfinite(e)) {
            vmx = fmaxf(vmx, e);
            vmn = fminf(vmn, e);
          }
        }
      });
      if (__builtin_isfinite(ec)) {
        const float sh = __fsub_rn(vmx, vmn);  // :143 (float)(max - min)
        if (c == 4) sh_own = sh;
        m = fmaxf(m, sh);                      // (NaN: no valid step height so far)
        count += sh > a.step_crit_lo ? 1 : 0;
      }
    }
  });
  float st = count == 0 ? (0.0 < a.step_crit ? 1.0f : 0.0f) : 0.0f;  // (k_step_score5's emit)
  if (count > 0 && count < a.step_ncrit) {
    const double sm = (double)(m > 0.0f ? m : 0.0f);                // stepMax starts at 0.0 (:149)
    const double a1 = ((double)count / (double)a.step_ncrit) * sm;  // nCells / nCellCritical_ * stepMax (:169)
    const double step = sm < a1 ? sm : a1;                          // :170
    st = step < a.step_crit ? (float)(1.0 - step / a.step_crit) : 0.0f;
  }
  st = (m == m) ? st : qnan();  // no valid step height in the window: the cell stays NaN (:161)
  a.slope[o] = o_slope;
  a.rough[o] = o_rough;
  a.step[o] = st;
  a.step_height[o] = sh_own;
  if (a.combine) {  // MathExpressionFilter, fixed form, float32, left to right
    const float ta = a.w_slope * o_slope, tb = a.w_step * st, tc = a.w_rough * o_rough;
    const float tab = ta + tb;
    const float tabc = tab + tc;
    a.trav[o] = a.w_scale * tabc;
  }
  if (KEEP) {
    a.nx[o] = fx;
    a.ny[o] = fy;
    a.nz[o] = fz;
  }
}

}  // namespace

namespace {
// the arguments both kernels share
void fill_common(SmallArgs& a, const Geo& g, const ChainParams& p, const Layers& L, const Region& r) {
  a.elev = L.elev;
  a.slope = L.slope;
  a.rough = L.rough;
  a.nx = L.nx;
  a.ny = L.ny;
  a.nz = L.nz;
  a.rows = g.rows;
  a.cols = g.cols;
  a.map_cells = (long long)g.rows * g.cols;
  a.map = r.map;
  a.i_lo = r.i0;
  a.i_hi = r.i1;
  a.j_lo = r.j0;
  a.j_hi = r.j1;
  a.res = g.res;
  a.r2 = p.normals.r2;
  a.ax = g.ax;
  a.ay = g.ay;
  a.inv_slope_crit = (float)(1.0 / p.slope_crit);
  a.inv_rough_crit = (float)(1.0 / p.rough_crit);
  a.band_slope = clip_band_slope(p.slope_crit);
  a.band_rough = clip_band_rough(p.rough_crit);
  a.step = L.step;
  a.trav = L.trav;
  a.step_height = L.step_height;
  a.step_valid = 0.0 < p.step_crit ? 1.0f : 0.0f;
  a.slope_crit = p.slope_crit;
  a.rough_crit = p.rough_crit;
  a.w_scale = p.w_scale;
  a.w_slope = p.w_slope;
  a.w_step = p.w_step;
  a.w_rough = p.w_rough;
  a.disc25 = a.tie25 = a.need25 = 0;
  a.win1_9 = a.win2_9 = 0x010u;
  a.step_crit = p.step_crit;
  a.step_crit_lo = (float)p.step_crit;
  if ((double)a.step_crit_lo > p.step_crit) a.step_crit_lo = nextafterf(a.step_crit_lo, -INFINITY);
  a.step_ncrit = p.step_ncrit;
  a.write_step = a.combine = 0;
}

}  // namespace

// Discs that reach at most two cells, tie radii included: the one-cell tie radius of the default parameters on a 0.05 m
// map first of all.  Tie-free discs of that size only on small launches (the sliding kernels are faster from about 2^18
// cells on: 4096^2 at 1.67 cells 0.09 ms).  False: not taken.
bool normals_small(const Geo& g, const ChainParams& p, const Layers& L, bool keep_normals, const Region& r, int* flags, FastGrid* fg,
                   hipStream_t s, bool write_step, bool combine) {
  const Disc& d = p.normals;
  static const bool off = lab_flag("TE_NO_SMALL");  // measurement aid
  static const int max_cells_env = lab_int("TE_SMALL_MAX_CELLS", 0);
  (void)flags;  // (no cell is left to the fix-up pass)
  if (off || d.reach < 1 || d.reach > 2) return false;
  if ((double)g.rows * (double)g.cols * 4.0 >= 4294967296.0) return false;
  const long long cells = (long long)(r.i1 - r.i0) * (r.j1 - r.j0) * (r.map >= 0 ? 1 : g.batch);
  const long long small_launch = max_cells_env > 0 ? max_cells_env : (1ll << 18);
  if (d.n_ties == 0 && cells > small_launch) return false;
  SmallArgs a;
  a.n_off = 0;
  a.tie_mask = 0;
  // the generic order: the runs row by row, then the circle cells (the sums are order-dependent only below the tolerance)
  for (int dj = -d.R; dj <= d.R; ++dj) {
    const int hw = d.hw[dj < 0 ? -dj : dj];
    for (int di = -hw; di <= hw; ++di) {
      if (di == 0 && dj == 0) continue;
      if (a.n_off >= kSmallMaxOffsets) return false;
      a.di[a.n_off] = (signed char)di;
      a.dj[a.n_off] = (signed char)dj;
      ++a.n_off;
    }
  }
  for (int t = 0; t < d.n_ties; ++t) {
    if (a.n_off >= kSmallMaxOffsets) return false;
    a.di[a.n_off] = d.tie_di[t];
    a.dj[a.n_off] = d.tie_dj[t];
    a.tie_mask |= 1u << a.n_off;
    ++a.n_off;
  }
  if (a.n_off < 2) return false;  // fewer than three points in every disc: UnitZ everywhere, the generic kernel's business
  for (int k = a.n_off; k < kSmallMaxOffsets; ++k) a.di[k] = a.dj[k] = 0;
  fill_common(a, g, p, L, r);
  a.write_step = write_step ? 1 : 0;
  a.combine = combine ? 1 : 0;
  fg->ntx = (r.i1 - r.i0 + kLanes - 1) / kLanes;
  fg->nty = (r.j1 - r.j0 + 15) / 16;
  fg->nbz = r.map >= 0 ? 1 : g.batch;
  fg->frame = -1;  // nothing is left to the fix-up pass: the caller does not launch it
  const dim3 grid((unsigned)fg->ntx, (unsigned)((r.j1 - r.j0 + kSmallBY - 1) / kSmallBY), (unsigned)fg->nbz);
  // the one-cell tie radius: the four edge neighbours, every one of them on the circle (any order in the table)
  bool cross = a.n_off == 4 && a.tie_mask == 0xfu;
  for (int k = 0; k < 4 && cross; ++k) cross = (a.di[k] == 0) != (a.dj[k] == 0) && a.di[k] * a.di[k] + a.dj[k] * a.dj[k] == 1;
  for (int k = 0; k < 4 && cross; ++k)
    for (int q = 0; q < k; ++q) cross = cross && !(a.di[k] == a.di[q] && a.dj[k] == a.dj[q]);
  if (cross) {
    if (keep_normals)
      hipLaunchKernelGGL((k_normals_small<true, true>), grid, dim3(kLanes, kSmallBY), 0, s, a);
    else
      hipLaunchKernelGGL((k_normals_small<false, true>), grid, dim3(kLanes, kSmallBY), 0, s, a);
    return true;
  }
  if (keep_normals)
    hipLaunchKernelGGL((k_normals_small<true, false>), grid, dim3(kLanes, kSmallBY), 0, s, a);
  else
    hipLaunchKernelGGL((k_normals_small<false, false>), grid, dim3(kLanes, kSmallBY), 0, s, a);
  return true;
}

// The whole chain of a small whole-map launch in one kernel (k_chain_window); false: not taken.  Conditions: at most 2^18
// cells, a normals disc that reaches at most two cells (tie cells included), both step windows tie-free and at most 3 x 3.
bool chain_window(const Geo& g, const ChainParams& p, const Layers& L, bool keep_normals, const Region& r, bool combine, hipStream_t s) {
  static const bool off = lab_flag("TE_NO_CHAIN_WINDOW");  // measurement aid
  const Disc& d = p.normals;
  if (off || d.reach < 1 || d.reach > 2 || p.step1.n_ties != 0 || p.step2.n_ties != 0 || p.step1.Q < 0 || p.step1.Q > 2 || p.step2.Q < 0 || p.step2.Q > 2) return false;
  const long long cells = (long long)(r.i1 - r.i0) * (r.j1 - r.j0) * (r.map >= 0 ? 1 : g.batch);
  if (cells > (1ll << 18) || (double)g.rows * (double)g.cols * 4.0 >= 4294967296.0) return false;
  SmallArgs a;
  a.n_off = 0;
  a.tie_mask = 0;
  for (int k = 0; k < kSmallMaxOffsets; ++k) a.di[k] = a.dj[k] = 0;
  fill_common(a, g, p, L, r);
  int points = 1;
  for (int dj = -d.R; dj <= d.R; ++dj) {
    const int hw = d.hw[dj < 0 ? -dj : dj];
    for (int di = -hw; di <= hw; ++di)
      if (di || dj) {
        a.disc25 |= 1u << ((dj + 2) * 5 + (di + 2));
        ++points;
      }
  }
  for (int t = 0; t < d.n_ties; ++t) {
    if (d.tie_di[t] == 0 && d.tie_dj[t] == 0) return false;
    a.tie25 |= 1u << ((d.tie_dj[t] + 2) * 5 + (d.tie_di[t] + 2));
    ++points;
  }
  if (points < 3) return false;  // UnitZ everywhere: the generic kernel's business
  auto window = [](int Q) {
    unsigned m = 0;
    for (int dj = -1; dj <= 1; ++dj)
      for (int di = -1; di <= 1; ++di)
        if (di * di + dj * dj <= Q) m |= 1u << ((dj + 1) * 3 + (di + 1));
    return m;
  };
  a.win1_9 = window(p.step1.Q);
  a.win2_9 = window(p.step2.Q);
  a.need25 = a.disc25 | a.tie25 | (1u << 12);
  for (int c = 0; c < 9; ++c)      // a cell of the second window ...
    for (int w = 0; w < 9; ++w)    // ... and a cell of ITS first window
      if (((a.win2_9 >> c) & 1u) && ((a.win1_9 >> w) & 1u)) a.need25 |= 1u << ((c / 3 - 1 + w / 3 - 1 + 2) * 5 + (c % 3 - 1 + w % 3 - 1 + 2));
  a.combine = combine ? 1 : 0;
  const dim3 grid((unsigned)((r.i1 - r.i0 + kLanes - 1) / kLanes), (unsigned)((r.j1 - r.j0 + kSmallBY - 1) / kSmallBY), (unsigned)(r.map >= 0 ? 1 : g.batch));
  if (keep_normals)
    hipLaunchKernelGGL(k_chain_window<true>, grid, dim3(kLanes, kSmallBY), 0, s, a);
  else
    hipLaunchKernelGGL(k_chain_window<false>, grid, dim3(kLanes, kSmallBY), 0, s, a);
  return true;
}

}  // namespace fast
}  // namespace te
