// te_fast_step.hip -- shape-specialised StepFilter kernels (marching wavefront, see te_march.h).
//
//   k_step_height_fast<Q>  StepFilter::update first pass   traversability_estimation_filters/src/StepFilter.cpp:112-144
//   k_step_score_fast<Q>   StepFilter::update second pass  StepFilter.cpp:147-178
//
// Pure compare/select arithmetic on float32 (plus one double subtraction), so the results are
// bit-identical to the reference.  Invalid cells are staged as quiet NaN and v_max/v_min ignore them,
// which is exactly the reference's isValid() skip; cells outside the map are staged as NaN too
// (CircleIterator clamps at the border).
// TIE RADII (radius a whole number R of cells): CircleIterator::isInside decides the cells exactly on the circle from
// rounded positions, centre by centre.  Maximum, minimum and count are folds, so the marching kernels run with the
// shape WITHOUT its circle (RAW = true: they store the running maximum / minimum, or maximum / count, instead of the
// result) and k_step_height_ties / k_step_score_ties fold the accepted circle cells in, cell by cell, and finish with
// the kernels' own arithmetic.  One scratch layer (the minimum, then the count).
#include "te_geom.h"
#include "te_march.h"

#include <cstdlib>

namespace te {
namespace fast {

namespace {

// waves per SIMD the kernels are compiled for (register budget 512 / waves); the launchers size the
// strips so that the whole grid is resident at this occupancy
#ifndef TE_SCORE_WAVES
#define TE_SCORE_WAVES 2  // (3: 168 VGPRs and a few spills, but a wave of it then fits beside two k_normals3 waves -- tools/build_variant.sh)
#endif
constexpr int kHeightWaves = 3, kScoreWaves = TE_SCORE_WAVES;

template <int Q, bool RAW = false>
__global__ __launch_bounds__(kLanes, kHeightWaves) void k_step_height_fast(Geo g, const float* __restrict__ elev,
                                                                           float* __restrict__ sh, Region rg,
                                                                           int periods, float* __restrict__ sh_min = nullptr) {
  using S = Shape<Q>;
  using T = Strip<Q>;
  constexpr int R = S::R, P = S::P, W = T::W;
  __shared__ float rowbuf[P * W];
  const int lane = threadIdx.x;
  const int map = rg.map >= 0 ? rg.map : blockIdx.z;
  const size_t mo = (size_t)map * g.rows * g.cols;
  // (the last block of a row of blocks is shifted left to end at the region's edge: no lane is ever masked; the columns
  // it shares with its neighbour are written twice with the same bits)
  const int i0 = rg.i0 + (int)blockIdx.x * kLanes + kLanes > rg.i1 ? rg.i1 - kLanes : rg.i0 + (int)blockIdx.x * kLanes;
  const int out_rows = T::out_rows(periods);
  const int js = rg.j0 + blockIdx.y * out_rows;
  const int jstop = js + out_rows < rg.j1 ? js + out_rows : rg.j1;  // one past the last output row of this strip
  typedef float __attribute__((address_space(1))) gfloat;
  float amax[P], amin[P], zc[P];
  static_for<P>([&](auto kc) __attribute__((always_inline)) {
    constexpr int k = decltype(kc)::value;
    amax[k] = amin[k] = zc[k] = qnan();
  });

  PeriodLoader<Q> loader;
  float stage[PeriodLoader<Q>::NLD];
  loader.init(g, lane);
  loader.load(stage, elev + mo, g, js - R, i0 - R, lane);
#pragma unroll 1
  for (int per = 0; per < periods; ++per) {
    const int rbase = js - R + per * P;
    if (rbase - R >= rg.j1) break;  // nothing left to emit (uniform)
    __syncthreads();
    loader.store(rowbuf, stage, lane, [](float t) { return __builtin_isfinite(t) ? t : qnan(); });
    __syncthreads();
    if (per + 1 < periods) loader.load(stage, elev + mo, g, rbase + P, i0 - R, lane);  // in flight during the period
    // Row p of the period completes output row rbase + p - R.  Which of the P rows emit is one bit mask per period
    // (uniform: a scalar bit test per emit instead of three compares), and the output address is a scalar row pointer
    // that advances by two rows per pass + one of two constant lane offsets (first / second row of the pass).
    const int p_lo = js - (rbase - R) > 0 ? js - (rbase - R) : 0, p_hi = jstop - (rbase - R) < P ? jstop - (rbase - R) : P;
    const unsigned emask = p_hi > p_lo ? (p_hi >= 32 ? ~0u : (1u << p_hi) - 1u) & ~((1u << p_lo) - 1u) : 0u;
    gfloat* op = (gfloat*)(sh + mo + ((long long)(rbase - R) * g.rows + i0));  // output row of period row 0 (may lie above the strip: never stored)
    gfloat* op_min = RAW ? (gfloat*)(sh_min + mo + ((long long)(rbase - R) * g.rows + i0)) : nullptr;
    // Two rows per pass: every pending output takes the run values of both rows with ONE v_max3/v_min3.
    // Row p is at offset e1 (slot = (p+e1) mod P) and row p+1 at e1-1 of the same output; the output that
    // completes with row p (e1 == -R) is emitted in between and its slot restarts with row p+1 (offset +R).
    auto horiz = [&](int p, float (&mx)[R + 1], float (&mn)[R + 1]) __attribute__((always_inline)) {
      const float* row = rowbuf + p * W + lane + R;
      mx[0] = mn[0] = row[0];
      static_for<R>([&](auto dc) __attribute__((always_inline)) {
        constexpr int d = decltype(dc)::value + 1;
        const float a = row[-d], b = row[d];
        vmax3_min3(mx[d], mn[d], mx[d - 1], mn[d - 1], a, b);
      });
      zc[p] = row[0];
    };
    auto emit = [&](int p, float vmx, float vmn) __attribute__((always_inline)) {  // the output row completed by row p of this period
      const int so = (p + R + 1) % P;
      if ((emask >> p) & 1u) {
        const float z0 = zc[so];
        // StepFilter.cpp:113 only valid centres; :143 double difference stored as float
        // (float)((double)vmx - (double)vmn) == vmx - vmn in float32: the double difference of two floats rounded to
        // float is the correctly rounded float difference (53 >= 2 * 24 + 2 bits: double rounding is innocuous)
        if constexpr (RAW) {  // the fold over the circle cells comes first (k_step_height_ties)
          op[(p & 1) ? g.rows + lane : lane] = vmx;
          op_min[(p & 1) ? g.rows + lane : lane] = vmn;
        } else {
          const float out = (z0 == z0) ? __fsub_rn(vmx, vmn) : qnan();
          op[(p & 1) ? g.rows + lane : lane] = out;
        }
      }
    };
    static_for<(P + 1) / 2>([&](auto pc) __attribute__((always_inline)) {
      constexpr int p = 2 * decltype(pc)::value;
      // keep the passes of the period apart: otherwise the scheduler hoists the LDS reads of many passes
      // and the run arrays of all of them are live at once
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (p + 1 < P) {
        float mx1[R + 1], mn1[R + 1], mx2[R + 1], mn2[R + 1];
        horiz(p, mx1, mn1);
        horiz(p + 1, mx2, mn2);
        static_for<P>([&](auto sc) __attribute__((always_inline)) {
          constexpr int sl = decltype(sc)::value;
          constexpr int e0 = ((sl - p) % P + P) % P;
          constexpr int e1 = e0 > R ? e0 - P : e0;
          if constexpr (e1 == -R) {
            emit(p, vmax2(amax[sl], mx1[S::hw(R)]), vmin2(amin[sl], mn1[S::hw(R)]));
            amax[sl] = mx2[S::hw(R)];
            amin[sl] = mn2[S::hw(R)];
          } else {
            constexpr int w1 = S::hw(e1 < 0 ? -e1 : e1), w2 = S::hw(e1 - 1 < 0 ? 1 - e1 : e1 - 1);
            vmax3_min3(amax[sl], amin[sl], amax[sl], amin[sl], mx1[w1], mx2[w2], mn1[w1], mn2[w2]);
            if constexpr (e1 - 1 == -R) {
              emit(p + 1, amax[sl], amin[sl]);
              amax[sl] = amin[sl] = qnan();
            }
          }
        });
      } else {  // last (odd) row of the period on its own
        float mx[R + 1], mn[R + 1];
        horiz(p, mx, mn);
        static_for<P>([&](auto ec) __attribute__((always_inline)) {
          constexpr int e = decltype(ec)::value - R;
          constexpr int slot = (p + e + P) % P;
          constexpr int w = S::hw(e < 0 ? -e : e);
          amax[slot] = vmax2(amax[slot], mx[w]);
          amin[slot] = vmin2(amin[slot], mn[w]);
        });
        constexpr int so = (p + R + 1) % P;
        emit(p, amax[so], amin[so]);
        amax[so] = amin[so] = qnan();
      }
      op += 2 * (long long)g.rows;
      if constexpr (RAW) op_min += 2 * (long long)g.rows;
    });
  }
}

// crit_lo = largest float <= critical_value, so that for a float s:  (double)s > crit  <=>  s > crit_lo.
template <int Q, bool RAW = false>
__global__ __launch_bounds__(kLanes, kScoreWaves) void k_step_score_fast(Geo g, double crit, double rcrit, float crit_lo, int ncrit,
                                                                         const float* __restrict__ shl,
                                                                         float* __restrict__ out, Region rg,
                                                                         int periods, float* __restrict__ out_count = nullptr) {
  using S = Shape<Q>;
  using T = Strip<Q>;
  constexpr int R = S::R, P = S::P, W = T::W;
  __shared__ float2 rowbuf[P * W];  // {step_height, (step_height > crit) as integer bits}
  const int lane = threadIdx.x;
  const int map = rg.map >= 0 ? rg.map : blockIdx.z;
  const size_t mo = (size_t)map * g.rows * g.cols;
  const int i0 = rg.i0 + (int)blockIdx.x * kLanes + kLanes > rg.i1 ? rg.i1 - kLanes : rg.i0 + (int)blockIdx.x * kLanes;  // see k_step_height_fast
  const int out_rows = T::out_rows(periods);
  const int js = rg.j0 + blockIdx.y * out_rows;
  const int jstop = js + out_rows < rg.j1 ? js + out_rows : rg.j1;
  typedef float __attribute__((address_space(1))) gfloat;
  // nCells / nCellCritical_ for every possible count, divided once per block (exactly the reference's
  // double division); first read after the barriers of the first period
  __shared__ double ratio[S::npoints() + 1];
  for (int k = lane; k <= S::npoints(); k += kLanes) ratio[k] = (double)k / (double)ncrit;
  const float one_if_crit = 0.0 < crit ? 1.0f : 0.0f;
  float vm[P];  // NaN-ignoring max of the valid step heights (NaN == no valid cell yet)
  int cnt[P];
  static_for<P>([&](auto kc) __attribute__((always_inline)) {
    constexpr int k = decltype(kc)::value;
    vm[k] = qnan();
    cnt[k] = 0;
  });

  PeriodLoader<Q> loader;
  float stage[PeriodLoader<Q>::NLD];
  loader.init(g, lane);
  loader.load(stage, shl + mo, g, js - R, i0 - R, lane);
#pragma unroll 1
  for (int per = 0; per < periods; ++per) {
    const int rbase = js - R + per * P;
    if (rbase - R >= rg.j1) break;
    __syncthreads();
    // staged step heights are finite or NaN
    loader.store(rowbuf, stage, lane, [&](float v) { return make_float2(v, __int_as_float(v > crit_lo ? 1 : 0)); });
    __syncthreads();
    if (per + 1 < periods) loader.load(stage, shl + mo, g, rbase + P, i0 - R, lane);
    // emit mask and running output pointer of the period: see k_step_height_fast
    const int p_lo = js - (rbase - R) > 0 ? js - (rbase - R) : 0, p_hi = jstop - (rbase - R) < P ? jstop - (rbase - R) : P;
    const unsigned emask = p_hi > p_lo ? (p_hi >= 32 ? ~0u : (1u << p_hi) - 1u) & ~((1u << p_lo) - 1u) : 0u;
    gfloat* op = (gfloat*)(out + mo + ((long long)(rbase - R) * g.rows + i0));
    gfloat* op_cnt = RAW ? (gfloat*)(out_count + mo + ((long long)(rbase - R) * g.rows + i0)) : nullptr;
    // One row = 2R+1 staged cells {value, flag}; the reads of the NEXT row are issued before the current row is
    // reduced (two row buffers alternate), so the LDS latency is covered by the reduction and the scatter.
    auto read_row = [&](int p, float2 (&raw)[2 * R + 1]) __attribute__((always_inline)) {
      const float2* row = rowbuf + p * W + lane + R;
      static_for<2 * R + 1>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        raw[k] = row[k - R];
      });
    };
    auto horiz = [&](const float2 (&raw)[2 * R + 1], float (&mx)[R + 1], int (&cn)[R + 1]) __attribute__((always_inline)) {
      mx[0] = raw[R].x;
      cn[0] = __float_as_int(raw[R].y);
      static_for<R>([&](auto dc) __attribute__((always_inline)) {
        constexpr int d = decltype(dc)::value + 1;
        vmax3_add3(mx[d], cn[d], mx[d - 1], raw[R - d].x, raw[R + d].x, cn[d - 1], __float_as_int(raw[R - d].y),
                   __float_as_int(raw[R + d].y));
      });
    };
    auto emit = [&](int p, float m, int count) __attribute__((always_inline)) {
      if constexpr (RAW) {  // the fold over the circle cells comes first (k_step_score_ties)
        if ((emask >> p) & 1u) {
          op[(p & 1) ? g.rows + lane : lane] = m;
          op_cnt[(p & 1) ? g.rows + lane : lane] = __int_as_float(count);
        }
        return;
      }
      if ((emask >> p) & 1u) {
        // isValid: at least one valid step_height in the window (StepFilter.cpp:161), else the cell stays NaN.
        // nCells == 0: step = min(stepMax, 0 * stepMax) = 0 (:169-170) -> 1 - 0 / crit = 1 (0 if crit == 0: "0 < 0" fails);
        // nCells >= nCellCritical: the ratio is >= 1, so step = stepMax, and a counted cell means stepMax > crit -> 0.
        // Only 0 < nCells < nCellCritical needs the arithmetic, and a wavefront rarely holds such a cell.
        float o = count == 0 ? one_if_crit : 0.0f;
        if (__builtin_expect(__any(count > 0 && count < ncrit), 0)) {
          const double sm = (double)vmax2_zero(m);  // stepMax starts at 0.0 (:149)
          const double a1 = ratio[count] * sm;       // nCells / nCellCritical_ * stepMax (:169)
          const double step = sm < a1 ? sm : a1;     // :170
          // step / crit without the division sequence: q0 = step * RN(1/crit), then two residual corrections
          // (Markstein: the first makes q faithful, the second correctly rounded), all branch-free
          const double q0 = step * rcrit;
          const double q1 = fma(fma(-q0, crit, step), rcrit, q0);
          const double q = fma(fma(-q1, crit, step), rcrit, q1);
          o = step < crit ? (float)(1.0 - q) : 0.0f;
        }
        o = (m == m) ? o : qnan();
        op[(p & 1) ? g.rows + lane : lane] = o;
      }
    };
    float2 rawa[2 * R + 1], rawb[2 * R + 1];
    read_row(0, rawa);
    static_for<(P + 1) / 2>([&](auto pc) __attribute__((always_inline)) {
      constexpr int p = 2 * decltype(pc)::value;  // row p is in rawa
      if constexpr (p + 1 < P) {  // two rows per pass
        float mx1[R + 1], mx2[R + 1];
        int cn1[R + 1], cn2[R + 1];
        read_row(p + 1, rawb);
        horiz(rawa, mx1, cn1);
        if constexpr (p + 2 < P) read_row(p + 2, rawa);
        horiz(rawb, mx2, cn2);
        static_for<P>([&](auto sc) __attribute__((always_inline)) {
          constexpr int sl = decltype(sc)::value;
          constexpr int e0 = ((sl - p) % P + P) % P;
          constexpr int e1 = e0 > R ? e0 - P : e0;
          if constexpr (e1 == -R) {
            emit(p, vmax2(vm[sl], mx1[S::hw(R)]), cnt[sl] + cn1[S::hw(R)]);
            vm[sl] = mx2[S::hw(R)];
            cnt[sl] = cn2[S::hw(R)];
          } else {
            constexpr int w1 = S::hw(e1 < 0 ? -e1 : e1), w2 = S::hw(e1 - 1 < 0 ? 1 - e1 : e1 - 1);
            vmax3_add3(vm[sl], cnt[sl], vm[sl], mx1[w1], mx2[w2], cnt[sl], cn1[w1], cn2[w2]);
            if constexpr (e1 - 1 == -R) {
              emit(p + 1, vm[sl], cnt[sl]);
              vm[sl] = qnan();
              cnt[sl] = 0;
            }
          }
        });
      } else {
        float mx[R + 1];
        int cn[R + 1];
        horiz(rawa, mx, cn);
        static_for<P>([&](auto ec) __attribute__((always_inline)) {
          constexpr int e = decltype(ec)::value - R;
          constexpr int slot = (p + e + P) % P;
          constexpr int w = S::hw(e < 0 ? -e : e);
          vm[slot] = vmax2(vm[slot], mx[w]);
          cnt[slot] += cn[w];
        });
        constexpr int so = (p + R + 1) % P;
        emit(p, vm[so], cnt[so]);
        vm[so] = qnan();
        cnt[so] = 0;
      }
      op += 2 * (long long)g.rows;
      if constexpr (RAW) op_cnt += 2 * (long long)g.rows;
    });
  }
}

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

// StepFilter.cpp:112-144 finished: sh holds the maximum, sh_min the minimum over the valid cells of the disc without its circle
__global__ __launch_bounds__(256) void k_step_height_ties(Geo g, TieArgs t, const float* __restrict__ elev, float* __restrict__ sh,
                                                          const float* __restrict__ sh_min, Region rg) {
  const int i = rg.i0 + (int)(blockIdx.x * blockDim.x + threadIdx.x), j = rg.j0 + (int)blockIdx.y;
  if (i >= rg.i1) return;
  const size_t mo = (size_t)(rg.map >= 0 ? rg.map : (int)blockIdx.z) * g.rows * g.cols;
  const size_t o = mo + (size_t)j * g.rows + i;
  float vmx = sh[o], vmn = sh_min[o];
  for (int k = 0; k < t.n_ties; ++k) {
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
  for (int k = 0; k < t.n_ties; ++k) {
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

// resident wave slots of the device for a kernel compiled for `waves` waves per SIMD
long wave_slots(int waves) {
  static const int ov = lab_int("TE_STEP_WAVES", 0);  // measurement aid: strips sized for this many waves per SIMD
  return 4L * device_cus() * (ov > 0 ? ov : waves);
}

// the shapes a whole-cell radius of 2 .. 10 cells leaves without its circle (te_march.h has them all): only these exist as RAW kernels
constexpr bool tie_free_part(int Q) { return Q == 2 || Q == 8 || Q == 13 || Q == 20 || Q == 34 || Q == 45 || Q == 61 || Q == 80 || Q == 98; }

template <int Q>
bool launch_height(const Geo& g, const float* elev, float* sh, const Region& r, hipStream_t s, float* sh_min = nullptr) {
  using T = Strip<Q>;
  const unsigned nx = (unsigned)((r.i1 - r.i0 + kLanes - 1) / kLanes), nz = (unsigned)(r.map >= 0 ? 1 : g.batch);
  const int periods = plan_periods(T::P, T::R, r.j1 - r.j0, (long)nx * nz, wave_slots(kHeightWaves));
  dim3 grid(nx, (unsigned)((r.j1 - r.j0 + T::out_rows(periods) - 1) / T::out_rows(periods)), nz);
  if (sh_min) {
    if constexpr (tie_free_part(Q)) {
      hipLaunchKernelGGL((k_step_height_fast<Q, true>), grid, dim3(kLanes), 0, s, g, elev, sh, r, periods, sh_min);
      return true;
    }
    return false;
  }
  hipLaunchKernelGGL((k_step_height_fast<Q, false>), grid, dim3(kLanes), 0, s, g, elev, sh, r, periods, (float*)nullptr);
  return true;
}

template <int Q>
bool launch_score(const Geo& g, double crit, float crit_lo, int ncrit, const float* sh, float* out, const Region& r,
                  hipStream_t s, float* out_count = nullptr) {
  using T = Strip<Q>;
  const unsigned nx = (unsigned)((r.i1 - r.i0 + kLanes - 1) / kLanes), nz = (unsigned)(r.map >= 0 ? 1 : g.batch);
  const int periods = plan_periods(T::P, T::R, r.j1 - r.j0, (long)nx * nz, wave_slots(kScoreWaves));
  dim3 grid(nx, (unsigned)((r.j1 - r.j0 + T::out_rows(periods) - 1) / T::out_rows(periods)), nz);
  if (out_count) {
    if constexpr (tie_free_part(Q)) {
      hipLaunchKernelGGL((k_step_score_fast<Q, true>), grid, dim3(kLanes), 0, s, g, crit, 1.0 / crit, crit_lo, ncrit, sh, out, r, periods, out_count);
      return true;
    }
    return false;
  }
  hipLaunchKernelGGL((k_step_score_fast<Q, false>), grid, dim3(kLanes), 0, s, g, crit, 1.0 / crit, crit_lo, ncrit, sh, out, r, periods, (float*)nullptr);
  return true;
}

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
  static const bool old_step = lab_flag("TE_OLD_STEP");  // measurement aid: the round-1 marching kernels of this file
  if (!old_step) return step_height5(Q, g, elev, sh, nullptr, r, s);
  if (r.i1 - r.i0 < kLanes) return false;  // the blocks are 64 cells wide and never mask a lane (the last one is shifted)
  switch (Q) {
#define X(q) \
  case q:    \
    return launch_height<q>(g, elev, sh, r, s);
    TE_DISC_SHAPES(X)
#undef X
    default:
      return false;
  }
}

// a tie radius (see the header); scratch: one float per cell of the layer
bool step_height_ties(const Disc& d, const Geo& g, const float* elev, float* sh, float* scratch, const Region& r, hipStream_t s) {
  static const bool off = lab_flag("TE_STEP_NO_TIES");  // measurement aid: tie radii to the generic kernels as before
  int q = 0;
  TieArgs t;
  if (off || !scratch || r.i1 - r.i0 < kLanes || !tie_disc(d, &q, &t)) return false;
  static const bool old_step = lab_flag("TE_OLD_STEP");
  bool ok = false;
  if (!old_step) {
    ok = step_height5(q, g, elev, sh, scratch, r, s);
  } else {
    switch (q) {
#define X(q_) \
  case q_:    \
    ok = launch_height<q_>(g, elev, sh, r, s, scratch); \
    break;
      TE_DISC_SHAPES(X)
#undef X
      default:
        break;
    }
  }
  if (!ok) return false;
  hipLaunchKernelGGL(k_step_height_ties, cell_grid(g, r), dim3(256), 0, s, g, t, elev, sh, (const float*)scratch, r);
  return true;
}

bool step_score_fast(int Q, const Geo& g, double crit, int ncrit, const float* sh, float* out, const Region& r,
                     hipStream_t s) {
  static const bool old_step = lab_flag("TE_OLD_STEP");
  if (!old_step) return step_score5(Q, g, crit, ncrit, sh, out, nullptr, r, s);
  if (r.i1 - r.i0 < kLanes) return false;
  // largest float <= crit
  float lo = (float)crit;
  if ((double)lo > crit) lo = nextafterf(lo, -INFINITY);
  switch (Q) {
#define X(q) \
  case q:    \
    return launch_score<q>(g, crit, lo, ncrit, sh, out, r, s);
    TE_DISC_SHAPES(X)
#undef X
    default:
      return false;
  }
}

bool step_score_ties(const Disc& d, const Geo& g, double crit, int ncrit, const float* sh, float* out, float* scratch, const Region& r,
                     hipStream_t s) {
  static const bool off = lab_flag("TE_STEP_NO_TIES");
  int q = 0;
  TieArgs t;
  if (off || !scratch || r.i1 - r.i0 < kLanes || !tie_disc(d, &q, &t)) return false;
  float lo = (float)crit;
  if ((double)lo > crit) lo = nextafterf(lo, -INFINITY);
  static const bool old_step = lab_flag("TE_OLD_STEP");
  bool ok = false;
  if (!old_step) {
    ok = step_score5(q, g, crit, ncrit, sh, out, scratch, r, s);
  } else {
    switch (q) {
#define X(q_) \
  case q_:    \
    ok = launch_score<q_>(g, crit, lo, ncrit, sh, out, r, s, scratch); \
    break;
      TE_DISC_SHAPES(X)
#undef X
      default:
        break;
    }
  }
  if (!ok) return false;
  hipLaunchKernelGGL(k_step_score_ties, cell_grid(g, r), dim3(256), 0, s, g, t, crit, lo, ncrit, sh, out, (const float*)scratch, r);
  return true;
}

}  // namespace fast
}  // namespace te
