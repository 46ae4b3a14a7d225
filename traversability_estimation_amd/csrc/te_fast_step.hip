// te_fast_step.hip -- shape-specialised StepFilter kernels (marching wavefront, see te_march.h).
//
//   k_step_height_fast<Q>  StepFilter::update first pass   traversability_estimation_filters/src/StepFilter.cpp:112-144
//   k_step_score_fast<Q>   StepFilter::update second pass  StepFilter.cpp:147-178
//
// Pure compare/select arithmetic on float32 (plus one double subtraction), so the results are
// bit-identical to the reference.  Invalid cells are staged as quiet NaN and v_max/v_min ignore them,
// which is exactly the reference's isValid() skip; cells outside the map are staged as NaN too
// (CircleIterator clamps at the border).
#include "te_march.h"

namespace te {
namespace fast {

namespace {

template <int Q>
__global__ __launch_bounds__(kLanes) void k_step_height_fast(Geo g, const float* __restrict__ elev,
                                                             float* __restrict__ sh, Region rg) {
  using S = Shape<Q>;
  using T = Strip<Q>;
  constexpr int R = S::R, P = S::P, W = T::W;
  __shared__ float rowbuf[P * W];
  const int lane = threadIdx.x;
  const int map = rg.map >= 0 ? rg.map : blockIdx.z;
  const size_t mo = (size_t)map * g.rows * g.cols;
  const int i0 = rg.i0 + blockIdx.x * kLanes;
  const int js = rg.j0 + blockIdx.y * T::out_rows;
  const int i = i0 + lane;
  float amax[P], amin[P], zc[P];
#pragma unroll
  for (int k = 0; k < P; ++k) amax[k] = amin[k] = zc[k] = qnan();

  float stage[T::NLD];
  load_period<Q>(stage, elev + mo, g, js - R, i0 - R, lane);
#pragma unroll 1
  for (int per = 0; per < T::periods; ++per) {
    const int rbase = js - R + per * P;
    if (rbase - R >= rg.j1) break;  // nothing left to emit (uniform)
    __syncthreads();
#pragma unroll
    for (int k = 0; k < T::NLD; ++k) {
      const int idx = lane + k * kLanes;
      const float t = stage[k];
      if (idx < P * W) rowbuf[idx] = __builtin_isfinite(t) ? t : qnan();
    }
    __syncthreads();
    if (per + 1 < T::periods) load_period<Q>(stage, elev + mo, g, rbase + P, i0 - R, lane);  // in flight during the period
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float* row = rowbuf + p * W + lane + R;
      float mx[R + 1], mn[R + 1];
      mx[0] = mn[0] = row[0];
#pragma unroll
      for (int d = 1; d <= R; ++d) {
        const float a = row[-d], b = row[d];
        mx[d] = vmax3(mx[d - 1], a, b);
        mn[d] = vmin3(mn[d - 1], a, b);
      }
      zc[p] = row[0];
#pragma unroll
      for (int e = -R; e <= R; ++e) {
        constexpr int dummy = 0;
        (void)dummy;
        const int slot = (p + e + P) % P;
        const int w = S::hw(e < 0 ? -e : e);
        amax[slot] = vmax2(amax[slot], mx[w]);
        amin[slot] = vmin2(amin[slot], mn[w]);
      }
      const int so = (p + R + 1) % P;
      const int j = rbase + p - R;
      if (j >= js && j < js + T::out_rows && j < rg.j1 && i < rg.i1) {
        const float z0 = zc[so];
        // StepFilter.cpp:113 only valid centres; :143 double difference stored as float
        const float out = (z0 == z0) ? (float)((double)amax[so] - (double)amin[so]) : qnan();
        sh[mo + (size_t)j * g.rows + i] = out;
      }
      amax[so] = amin[so] = qnan();
    }
  }
}

// crit_lo = largest float <= critical_value, so that for a float s:  (double)s > crit  <=>  s > crit_lo.
template <int Q>
__global__ __launch_bounds__(kLanes) void k_step_score_fast(Geo g, double crit, float crit_lo, int ncrit,
                                                            const float* __restrict__ shl,
                                                            float* __restrict__ out, Region rg) {
  using S = Shape<Q>;
  using T = Strip<Q>;
  constexpr int R = S::R, P = S::P, W = T::W;
  __shared__ float2 rowbuf[P * W];  // {step_height, (step_height > crit) as integer bits}
  const int lane = threadIdx.x;
  const int map = rg.map >= 0 ? rg.map : blockIdx.z;
  const size_t mo = (size_t)map * g.rows * g.cols;
  const int i0 = rg.i0 + blockIdx.x * kLanes;
  const int js = rg.j0 + blockIdx.y * T::out_rows;
  const int i = i0 + lane;
  float vm[P];  // NaN-ignoring max of the valid step heights (NaN == no valid cell yet)
  int cnt[P];
#pragma unroll
  for (int k = 0; k < P; ++k) {
    vm[k] = qnan();
    cnt[k] = 0;
  }

  float stage[T::NLD];
  load_period<Q>(stage, shl + mo, g, js - R, i0 - R, lane);
#pragma unroll 1
  for (int per = 0; per < T::periods; ++per) {
    const int rbase = js - R + per * P;
    if (rbase - R >= rg.j1) break;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < T::NLD; ++k) {
      const int idx = lane + k * kLanes;
      const float v = stage[k];  // finite or NaN
      if (idx < P * W) rowbuf[idx] = make_float2(v, __int_as_float(v > crit_lo ? 1 : 0));
    }
    __syncthreads();
    if (per + 1 < T::periods) load_period<Q>(stage, shl + mo, g, rbase + P, i0 - R, lane);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float2* row = rowbuf + p * W + lane + R;
      float mx[R + 1];
      int cn[R + 1];
      {
        const float2 c = row[0];
        mx[0] = c.x;
        cn[0] = __float_as_int(c.y);
      }
#pragma unroll
      for (int d = 1; d <= R; ++d) {
        const float2 a = row[-d], b = row[d];
        mx[d] = vmax3(mx[d - 1], a.x, b.x);
        cn[d] = cn[d - 1] + __float_as_int(a.y) + __float_as_int(b.y);
      }
#pragma unroll
      for (int e = -R; e <= R; ++e) {
        const int slot = (p + e + P) % P;
        const int w = S::hw(e < 0 ? -e : e);
        vm[slot] = vmax2(vm[slot], mx[w]);
        cnt[slot] += cn[w];
      }
      const int so = (p + R + 1) % P;
      const int j = rbase + p - R;
      if (j >= js && j < js + T::out_rows && j < rg.j1 && i < rg.i1) {
        const float m = vm[so];
        float o = qnan();
        if (m == m) {  // isValid: at least one valid step_height in the window (StepFilter.cpp:161)
          const double sm = (double)vmax2(m, 0.0f);  // stepMax starts at 0.0 (:149)
          const double a1 = (double)cnt[so] / (double)ncrit * sm;
          const double step = sm < a1 ? sm : a1;  // :170
          o = step < crit ? (float)(1.0 - step / crit) : 0.0f;
        }
        out[mo + (size_t)j * g.rows + i] = o;
      }
      vm[so] = qnan();
      cnt[so] = 0;
    }
  }
}

template <int Q>
void launch_height(const Geo& g, const float* elev, float* sh, const Region& r, hipStream_t s) {
  using T = Strip<Q>;
  dim3 grid((unsigned)((r.i1 - r.i0 + kLanes - 1) / kLanes), (unsigned)((r.j1 - r.j0 + T::out_rows - 1) / T::out_rows),
            (unsigned)(r.map >= 0 ? 1 : g.batch));
  hipLaunchKernelGGL(k_step_height_fast<Q>, grid, dim3(kLanes), 0, s, g, elev, sh, r);
}

template <int Q>
void launch_score(const Geo& g, double crit, float crit_lo, int ncrit, const float* sh, float* out, const Region& r,
                  hipStream_t s) {
  using T = Strip<Q>;
  dim3 grid((unsigned)((r.i1 - r.i0 + kLanes - 1) / kLanes), (unsigned)((r.j1 - r.j0 + T::out_rows - 1) / T::out_rows),
            (unsigned)(r.map >= 0 ? 1 : g.batch));
  hipLaunchKernelGGL(k_step_score_fast<Q>, grid, dim3(kLanes), 0, s, g, crit, crit_lo, ncrit, sh, out, r);
}

}  // namespace

#define TE_STEP_SHAPES(X) \
  X(0) X(1) X(2) X(4) X(5) X(8) X(9) X(10) X(13) X(16) X(17) X(18) X(20) X(25) X(26) X(29) X(32) X(34) X(36) X(37) \
  X(40) X(41) X(45) X(49) X(50) X(52) X(53) X(58) X(61) X(64) X(65) X(68) X(72) X(73) X(74) X(80) X(81) X(82) X(85) \
  X(89) X(90) X(97) X(98) X(100)

bool step_height_fast(int Q, const Geo& g, const float* elev, float* sh, const Region& r, hipStream_t s) {
  switch (Q) {
#define X(q) \
  case q:    \
    launch_height<q>(g, elev, sh, r, s); \
    return true;
    TE_STEP_SHAPES(X)
#undef X
    default:
      return false;
  }
}

bool step_score_fast(int Q, const Geo& g, double crit, int ncrit, const float* sh, float* out, const Region& r,
                     hipStream_t s) {
  // largest float <= crit
  float lo = (float)crit;
  if ((double)lo > crit) lo = nextafterf(lo, -INFINITY);
  switch (Q) {
#define X(q) \
  case q:    \
    launch_score<q>(g, crit, lo, ncrit, sh, out, r, s); \
    return true;
    TE_STEP_SHAPES(X)
#undef X
    default:
      return false;
  }
}

}  // namespace fast
}  // namespace te
