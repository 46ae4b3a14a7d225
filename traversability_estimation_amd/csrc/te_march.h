// te_march.h -- the "marching wavefront" framework shared by the fast (shape-specialised) kernels.
//
// One 64-lane wavefront owns 64 adjacent cells along the fast axis (grid_map row index i) and
// marches down the slow axis (column index j) over a strip of rows.  For a disc
// D = {(di,dj): di^2+dj^2 <= Q} every row r it visits is reduced ONCE along i into the nested run
// values  S_w(i, r) = reduce_{|di|<=w} f(i+di, r)  for the few distinct half-widths w of D (19 LDS
// reads for R=9, each value shared by up to 19 lanes), and S_{hw(|r-j|)} is then folded into the
// register accumulator of every output row j with |r-j| <= R.  The 2R+1 accumulators rotate; the
// rotation is resolved at compile time by unrolling one period of P = 2R+1 rows, so a 253-point
// stencil costs O(R) operations per cell instead of O(R^2) and nothing but the single row is staged
// on chip.  The disc shape (Q) is a template parameter: the run table is constexpr.
#pragma once
#include "te_internal.h"

namespace te {
namespace fast {

constexpr int isqrt_c(int v) {
  int r = 0;
  while ((r + 1) * (r + 1) <= v) ++r;
  return r;
}

template <int Q>
struct Shape {
  static constexpr int R = isqrt_c(Q);
  static constexpr int P = 2 * R + 1;  // accumulators in flight == rows per unrolled period
  // half-width of the run at column offset d (0 <= d <= R)
  static constexpr int hw(int d) { return isqrt_c(Q - d * d); }
  static constexpr int npoints() {
    int n = 0;
    for (int d = -R; d <= R; ++d) n += 2 * hw(d < 0 ? -d : d) + 1;
    return n;
  }
};

constexpr int kLanes = 64;
constexpr int kStripTarget = 128;  // rows of output per strip (rounded up to whole periods)

template <int Q, int TARGET = kStripTarget>
struct Strip {
  static constexpr int R = Shape<Q>::R, P = Shape<Q>::P;
  static constexpr int periods = (TARGET + 2 * R + P - 1) / P;
  static constexpr int steps = periods * P;       // rows visited per strip
  static constexpr int out_rows = steps - 2 * R;  // rows produced per strip
  static constexpr int W = kLanes + 2 * R;        // staged row width
  static constexpr int NLD = (P * W + kLanes - 1) / kLanes;  // global loads per lane per period
};

__device__ __forceinline__ float qnan() { return __builtin_nanf(""); }

// Issue the loads of one period (P rows x W columns starting at map row r0, map column c0) into
// registers; NaN outside the map.  All loads are issued back to back (they stay in flight while the
// previous period is processed); the caller writes them to LDS later.
template <int Q, int TARGET = kStripTarget, int NLD>
__device__ __forceinline__ void load_period(float (&v)[NLD], const float* __restrict__ layer, const Geo& g, int r0,
                                            int c0, int lane) {
  constexpr int P = Shape<Q>::P, W = Strip<Q, TARGET>::W;
  static_assert(NLD == Strip<Q, TARGET>::NLD, "staging register count");
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int idx = lane + k * kLanes;
    const int rr = idx / W, cc = idx - rr * W;
    const int r = r0 + rr, ci = c0 + cc;
    float t = qnan();
    if (idx < P * W && r >= 0 && r < g.cols && ci >= 0 && ci < g.rows) t = layer[(size_t)r * g.rows + ci];
    v[k] = t;
  }
}

// min/max that ignore (quiet) NaN operands, without the canonicalisation instruction the compiler
// adds around llvm.maxnum: staged values are either finite or the canonical quiet NaN.
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float vmin3(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float vmax2(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmin2(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

}  // namespace fast
}  // namespace te
