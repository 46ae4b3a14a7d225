// te_fast_step.hip -- StepFilter at the shapes the marching kernels serve: dispatch, and the tie-radius folds.
//
//   StepFilter::update first pass   traversability_estimation_filters/src/StepFilter.cpp:112-144
//   StepFilter::update second pass  StepFilter.cpp:147-178
//
// The marching kernels themselves are k_step_height5 / k_step_score5 (te_step5.hip, on te_march5.h); the round-1
// kernels that lived here (PeriodLoader, a period of rows in LDS, 168 / 216 registers) were retired in round 4 after the
// new ones passed the whole GPU suite bit for bit and measured faster (60 -> 46 us, 77 -> 64 us on the 4096^2 bench map).
// TIE RADII (radius a whole number R of cells): CircleIterator::isInside decides the cells exactly on the circle from
// rounded positions, centre by centre.  Maximum, minimum and count are folds, so the marching kernels run with the
// shape WITHOUT its circle (RAW: they store the running maximum / minimum, or maximum / count, instead of the
// result) and k_step_height_ties / k_step_score_ties fold the accepted circle cells in, cell by cell, and finish with
// the kernels' own arithmetic.  One scratch layer (the minimum, then the count).
// SMALL WINDOWS (round 6): a window that holds the centre alone (the default 0.04 m on a 0.05 m map) and a tie radius of
// one or two cells (0.04 m on a 0.04 m map) have nothing to march over: k_step_small gathers them directly, one cell per
// thread, every load issued before the first is used, tie cells decided per centre (the single-cell window, for which the
// marching kernel was instantiated too: 150 -> 25 us per pass on 4096^2; tie radii of one cell went to the generic kernels).
#include "te_geom.h"
#include "te_march.h"

namespace te {
namespace fast {

namespace {

// ---- tie radii: the accepted circle cells folded in, one cell per thread ------------------------------------------
struct TieArgs {
  int n_ties;
  int8_t di[kMaxTies], dj[kMaxTies];
  double r2;
};

// CircleIterator::isInside for the cell (i + di, j + dj) of the circle around (i, j), cell centres as te_geom.h has them
__device__ __forceinline__ bool tie_inside(const Geo& g, const TieArgs& t, int i, int j, int k) {
  const double dx = cell_x(g, i + t.di[k]) - cell_x(g, i), dy = cell_y(g, j + t.dj[k]) - cell_y(g, j);
  return dx * dx + dy * dy <= t.r2;
}

// The values of the circle cells a centre accepts (NaN for the others), every load issued before the first value is used:
// written as a loop that tests, loads and folds one cell per turn the two kernels below waited for one load per tie cell --
// 19 us per cell on a 4096^2 layer against 5 (tools/lab/step_small_ubench.hip); 100 -> 45 us per fold at 4 tie cells.
constexpr int kTieBatch = 12;  // tie cells of a whole-cell radius up to 24 cells (25 cells: 20 -- the remainder takes the loop)
__device__ __forceinline__ void gather_ties(const Geo& g, const TieArgs& t, const float* __restrict__ layer, size_t mo, int i, int j,
                                            float (&v)[kTieBatch]) {
#pragma unroll
  for (int k = 0; k < kTieBatch; ++k) {
    v[k] = qnan();
    if (k < t.n_ties) {  // (uniform)
      const int ii = i + t.di[k], jj = j + t.dj[k];
      if ((unsigned)ii < (unsigned)g.rows && (unsigned)jj < (unsigned)g.cols && tie_inside(g, t, i, j, k)) v[k] = layer[mo + (size_t)jj * g.rows + ii];
    }
  }
}

// StepFilter.cpp:112-144 finished: sh holds the maximum, sh_min the minimum over the valid cells of the disc without its circle
__global__ __launch_bounds__(256) void k_step_height_ties(Geo g, TieArgs t, const float* __restrict__ elev, float* __restrict__ sh,
                                                          const float* __restrict__ sh_min, Region rg) {
  const int i = rg.i0 + (int)(blockIdx.x * blockDim.x + threadIdx.x), j = rg.j0 + (int)blockIdx.y;
  if (i >= rg.i1) return;
  const size_t mo = (size_t)(rg.map >= 0 ? rg.map : (int)blockIdx.z) * g.rows * g.cols;
  const size_t o = mo + (size_t)j * g.rows + i;
  float vmx = sh[o], vmn = sh_min[o];
  float v[kTieBatch];
  gather_ties(g, t, elev, mo, i, j, v);
#pragma unroll
  for (int k = 0; k < kTieBatch; ++k)
    if (__builtin_isfinite(v[k])) {
      vmx = fmaxf(vmx, v[k]);  // (NaN: no valid cell so far)
      vmn = fminf(vmn, v[k]);
    }
  for (int k = kTieBatch; k < t.n_ties; ++k) {
    const int ii = i + t.di[k], jj = j + t.dj[k];
    if ((unsigned)ii >= (unsigned)g.rows || (unsigned)jj >= (unsigned)g.cols || !tie_inside(g, t, i, j, k)) continue;
    const float z = elev[mo + (size_t)jj * g.rows + ii];
    if (!__builtin_isfinite(z)) continue;
    vmx = fmaxf(vmx, z);  // (NaN: no valid cell so far)
    vmn = fminf(vmn, z);
  }
  const float z0 = elev[o];
  sh[o] = __builtin_isfinite(z0) ? __fsub_rn(vmx, vmn) : qnan();  // as k_step_height_fast's emit
}

// StepFilter.cpp:147-178 finished: out holds the maximum of the valid step heights, cnt how many exceed the critical value
__global__ __launch_bounds__(256) void k_step_score_ties(Geo g, TieArgs t, double crit, float crit_lo, int ncrit, const float* __restrict__ shl,
                                                         float* __restrict__ out, const float* __restrict__ cnt, Region rg) {
  const int i = rg.i0 + (int)(blockIdx.x * blockDim.x + threadIdx.x), j = rg.j0 + (int)blockIdx.y;
  if (i >= rg.i1) return;
  const size_t mo = (size_t)(rg.map >= 0 ? rg.map : (int)blockIdx.z) * g.rows * g.cols;
  const size_t o = mo + (size_t)j * g.rows + i;
  float m = out[o];
  int count = __float_as_int(cnt[o]);
  float v[kTieBatch];
  gather_ties(g, t, shl, mo, i, j, v);
#pragma unroll
  for (int k = 0; k < kTieBatch; ++k)
    if (__builtin_isfinite(v[k])) {
      m = fmaxf(m, v[k]);
      count += v[k] > crit_lo ? 1 : 0;
    }
  for (int k = kTieBatch; k < t.n_ties; ++k) {
    const int ii = i + t.di[k], jj = j + t.dj[k];
    if ((unsigned)ii >= (unsigned)g.rows || (unsigned)jj >= (unsigned)g.cols || !tie_inside(g, t, i, j, k)) continue;
    const float h = shl[mo + (size_t)jj * g.rows + ii];
    if (!__builtin_isfinite(h)) continue;
    m = fmaxf(m, h);
    count += h > crit_lo ? 1 : 0;
  }
  // (k_step_score_fast's emit, with the division it takes from a table)
  float res = count == 0 ? (0.0 < crit ? 1.0f : 0.0f) : 0.0f;
  if (count > 0 && count < ncrit) {
    const double sm = (double)(m > 0.0f ? m : 0.0f);               // stepMax starts at 0.0 (:149)
    const double a1 = ((double)count / (double)ncrit) * sm;         // nCells / nCellCritical_ * stepMax (:169)
    const double step = sm < a1 ? sm : a1;                          // :170
    res = step < crit ? (float)(1.0 - step / crit) : 0.0f;
  }
  out[o] = (m == m) ? res : qnan();
}

// ---- small windows: at most 13 cells, gathered directly ---------------------------------------------------------------
constexpr int kSmallWin = 12;  // offsets besides the centre (the 13-point disc: reach 2)
struct SmallWin {
  int n;
  signed char di[kSmallWin], dj[kSmallWin];
  unsigned test_mask;  // bit k: offset k lies on the circle -- CircleIterator::isInside decides it for every centre
  double r2;
};

// SCORE = false: StepFilter.cpp:112-144 (in: elevation, out: step_height); true: :147-178 (in: step_height, out: the score)
template <bool SCORE>
__global__ __launch_bounds__(256) void k_step_small(Geo g, SmallWin w, double crit, float crit_lo, int ncrit, const float* __restrict__ in,
                                                    float* __restrict__ out, Region rg) {
  const int i = rg.i0 + (int)blockIdx.x * kLanes + (int)threadIdx.x, j = rg.j0 + (int)blockIdx.y * 4 + (int)threadIdx.y;
  if (i >= rg.i1 || j >= rg.j1) return;
  const size_t mo = (size_t)(rg.map >= 0 ? rg.map : (int)blockIdx.z) * g.rows * g.cols;
  const size_t o = mo + (size_t)j * g.rows + i;
  const float c = in[o];
  const double xi = cell_x(g, i), yj = cell_y(g, j);
  float v[kSmallWin];
#pragma unroll
  for (int k = 0; k < kSmallWin; ++k) {
    v[k] = qnan();
    if (k < w.n) {  // (uniform)
      const int ii = i + w.di[k], jj = j + w.dj[k];
      bool inside = (unsigned)ii < (unsigned)g.rows && (unsigned)jj < (unsigned)g.cols;
      if ((w.test_mask >> k) & 1u) {
        const double dx = cell_x(g, ii) - xi, dy = cell_y(g, jj) - yj;
        inside = inside && (dx * dx + dy * dy <= w.r2);
      }
      if (inside) v[k] = in[mo + (size_t)jj * g.rows + ii];
    }
  }
  if constexpr (!SCORE) {
    const bool vc = __builtin_isfinite(c);
    float vmx = vc ? c : qnan(), vmn = vmx;
#pragma unroll
    for (int k = 0; k < kSmallWin; ++k)
      if (k < w.n && __builtin_isfinite(v[k])) {
        vmx = fmaxf(vmx, v[k]);  // (NaN: no valid cell so far)
        vmn = fminf(vmn, v[k]);
      }
    out[o] = vc ? __fsub_rn(vmx, vmn) : qnan();  // :113 only cells with a valid centre; :143 (float)(max - min)
  } else {
    const bool vc = __builtin_isfinite(c);
    float m = vc ? c : qnan();
    int count = (vc && c > crit_lo) ? 1 : 0;
#pragma unroll
    for (int k = 0; k < kSmallWin; ++k)
      if (k < w.n && __builtin_isfinite(v[k])) {
        m = fmaxf(m, v[k]);
        count += v[k] > crit_lo ? 1 : 0;
      }
    // (k_step_score5's emit)
    float res = count == 0 ? (0.0 < crit ? 1.0f : 0.0f) : 0.0f;
    if (count > 0 && count < ncrit) {
      const double sm = (double)(m > 0.0f ? m : 0.0f);               // stepMax starts at 0.0 (:149)
      const double a1 = ((double)count / (double)ncrit) * sm;         // nCells / nCellCritical_ * stepMax (:169)
      const double step = sm < a1 ? sm : a1;                          // :170
      res = step < crit ? (float)(1.0 - step / crit) : 0.0f;
    }
    out[o] = (m == m) ? res : qnan();  // no valid step height in the window: the cell stays NaN (:161)
  }
}

// the window of a tie-free shape Q, or of a disc with tie cells; false: more than kSmallWin cells besides the centre
bool small_window(int Q, const Disc* d, SmallWin* w) {
  w->n = 0;
  w->test_mask = 0;
  w->r2 = d ? d->r2 : 0.0;
  for (int k = 0; k < kSmallWin; ++k) w->di[k] = w->dj[k] = 0;
  auto push = [&](int di, int dj, bool test) {
    if (w->n >= kSmallWin) return false;
    w->di[w->n] = (signed char)di;
    w->dj[w->n] = (signed char)dj;
    if (test) w->test_mask |= 1u << w->n;
    ++w->n;
    return true;
  };
  if (d) {
    if (d->reach > 2) return false;
    for (int dj = -d->R; dj <= d->R; ++dj) {
      const int hw = d->hw[dj < 0 ? -dj : dj];
      for (int di = -hw; di <= hw; ++di)
        if ((di || dj) && !push(di, dj, false)) return false;
    }
    for (int t = 0; t < d->n_ties; ++t) {
      // (a radius of exactly 0: the circle IS the centre -- isInside accepts it, 0 <= 0 -- and the kernels count the centre
      // themselves: listed again it was counted twice, nCells 2 instead of 1; the sweep's seeds 20477 ... 23804, round 6)
      if (d->tie_di[t] == 0 && d->tie_dj[t] == 0) continue;
      if (!push(d->tie_di[t], d->tie_dj[t], true)) return false;
    }
    return true;
  }
  // tie-free windows: the centre alone.  (A gather through the vector cache costs 5 us per neighbour and pass on a 4096^2
  // layer -- tools/lab/step_small_ubench.hip: 0 / 4 / 8 neighbours 25 / 51 / 72 us -- so from five cells on the marching
  // kernel beside the normals kernel wins: default windows at res 0.03, chain 0.219 ms against 0.267 with this kernel
  // beside the normals kernel and 0.294 with it ahead of it.)
  return Q == 0;
}

dim3 small_grid(const Geo& g, const Region& r) {
  return dim3((unsigned)((r.i1 - r.i0 + kLanes - 1) / kLanes), (unsigned)((r.j1 - r.j0 + 3) / 4), (unsigned)(r.map >= 0 ? 1 : g.batch));
}

float largest_float_below(double crit) {  // largest float <= crit: "h > crit" in double == "h > this" in float
  float lo = (float)crit;
  if ((double)lo > crit) lo = nextafterf(lo, -INFINITY);
  return lo;
}

// the shapes a whole-cell radius of 3 .. 10 cells leaves without its circle (2 cells: k_step_small) (te_march.h has them all): only these exist as RAW kernels
constexpr bool tie_free_part(int Q) { return Q == 8 || Q == 13 || Q == 20 || Q == 34 || Q == 45 || Q == 61 || Q == 80 || Q == 98; }

// the shape of a tie disc without its circle (largest norm in its runs), its ties as kernel arguments; false: not a
// whole-cell radius this file serves
bool tie_disc(const Disc& d, int* q_free, TieArgs* t) {
  if (d.n_ties == 0 || d.n_ties > kMaxTies || d.R < 1) return false;
  int q = 0;
  for (int b = 0; b <= d.R; ++b)
    if (d.hw[b] >= 0 && d.hw[b] * d.hw[b] + b * b > q) q = d.hw[b] * d.hw[b] + b * b;
  const int n2 = d.reach * d.reach;
  for (int k = 0; k < d.n_ties; ++k)
    if ((int)d.tie_di[k] * d.tie_di[k] + (int)d.tie_dj[k] * d.tie_dj[k] != n2) return false;
  if (!tie_free_part(q)) return false;
  *q_free = q;
  t->n_ties = d.n_ties;
  for (int k = 0; k < kMaxTies; ++k) {
    t->di[k] = k < d.n_ties ? d.tie_di[k] : 0;
    t->dj[k] = k < d.n_ties ? d.tie_dj[k] : 0;
  }
  t->r2 = d.r2;
  return true;
}

dim3 cell_grid(const Geo& g, const Region& r) {
  return dim3((unsigned)((r.i1 - r.i0 + 255) / 256), (unsigned)(r.j1 - r.j0), (unsigned)(r.map >= 0 ? 1 : g.batch));
}

}  // namespace

bool step_height_fast(int Q, const Geo& g, const float* elev, float* sh, const Region& r, hipStream_t s) {
  SmallWin w;
  static const bool no_small = lab_flag("TE_STEP_NO_SMALL");  // measurement aid
  if (!no_small && small_window(Q, nullptr, &w)) {
    hipLaunchKernelGGL(k_step_small<false>, small_grid(g, r), dim3(kLanes, 4), 0, s, g, w, 0.0, 0.0f, 1, elev, sh, r);
    return true;
  }
  return step_height5(Q, g, elev, sh, nullptr, r, s);
}

// a tie radius (see the header); scratch: one float per cell of the layer
bool step_height_ties(const Disc& d, const Geo& g, const float* elev, float* sh, float* scratch, const Region& r, hipStream_t s) {
  static const bool off = lab_flag("TE_STEP_NO_TIES");  // measurement aid: tie radii to the generic kernels as before
  int q = 0;
  TieArgs t;
  SmallWin w;
  static const bool no_small = lab_flag("TE_STEP_NO_SMALL");
  if (!off && !no_small && d.n_ties != 0 && small_window(-1, &d, &w)) {  // a tie radius of one or two cells
    hipLaunchKernelGGL(k_step_small<false>, small_grid(g, r), dim3(kLanes, 4), 0, s, g, w, 0.0, 0.0f, 1, elev, sh, r);
    return true;
  }
  if (off || !scratch || r.i1 - r.i0 < kLanes || !tie_disc(d, &q, &t)) return false;
  if (!step_height5(q, g, elev, sh, scratch, r, s)) return false;
  hipLaunchKernelGGL(k_step_height_ties, cell_grid(g, r), dim3(256), 0, s, g, t, elev, sh, (const float*)scratch, r);
  return true;
}

bool step_score_fast(int Q, const Geo& g, double crit, int ncrit, const float* sh, float* out, const Region& r,
                     hipStream_t s) {
  SmallWin w;
  static const bool no_small = lab_flag("TE_STEP_NO_SMALL");
  if (!no_small && small_window(Q, nullptr, &w)) {
    hipLaunchKernelGGL(k_step_small<true>, small_grid(g, r), dim3(kLanes, 4), 0, s, g, w, crit, largest_float_below(crit), ncrit, sh, out, r);
    return true;
  }
  return step_score5(Q, g, crit, ncrit, sh, out, nullptr, r, s);
}

bool step_score_ties(const Disc& d, const Geo& g, double crit, int ncrit, const float* sh, float* out, float* scratch, const Region& r,
                     hipStream_t s) {
  static const bool off = lab_flag("TE_STEP_NO_TIES");
  int q = 0;
  TieArgs t;
  SmallWin w;
  static const bool no_small = lab_flag("TE_STEP_NO_SMALL");
  if (!off && !no_small && d.n_ties != 0 && small_window(-1, &d, &w)) {
    hipLaunchKernelGGL(k_step_small<true>, small_grid(g, r), dim3(kLanes, 4), 0, s, g, w, crit, largest_float_below(crit), ncrit, sh, out, r);
    return true;
  }
  if (off || !scratch || r.i1 - r.i0 < kLanes || !tie_disc(d, &q, &t)) return false;
  const float lo = largest_float_below(crit);
  if (!step_score5(q, g, crit, ncrit, sh, out, scratch, r, s)) return false;
  hipLaunchKernelGGL(k_step_score_ties, cell_grid(g, r), dim3(256), 0, s, g, t, crit, lo, ncrit, sh, out, (const float*)scratch, r);
  return true;
}

}  // namespace fast
}  // namespace te
