// te_march.h -- what the shape-specialised kernels share: the disc shapes, compile-time loops, NaN-ignoring min / max.
// (The marching framework itself is te_march5.h; the round-1 one -- PeriodLoader, a period of rows in LDS -- was retired
// with its last users in round 4.)  The scheme, for reference:
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
#include <type_traits>
#include <utility>

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

// The disc shapes (Q = largest di^2+dj^2 in the disc) with radius up to 10 cells: every sum of two squares <= 100.
#ifndef TE_DISC_SHAPES  // (tools: -D'TE_DISC_SHAPES(X)=X(81)' compiles one shape)
#define TE_DISC_SHAPES(X) \
  X(0) X(1) X(2) X(4) X(5) X(8) X(9) X(10) X(13) X(16) X(17) X(18) X(20) X(25) X(26) X(29) X(32) X(34) X(36) X(37) \
  X(40) X(41) X(45) X(49) X(50) X(52) X(53) X(58) X(61) X(64) X(65) X(68) X(72) X(73) X(74) X(80) X(81) X(82) X(85) \
  X(89) X(90) X(97) X(98) X(100)
#endif

__device__ __forceinline__ float qnan() { return __builtin_nanf(""); }

// Compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>).  Used where every
// register-array index has to be a constant (a "#pragma unroll" the optimiser declines leaves dynamic
// indices behind and the arrays go to scratch).
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
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
// max3 + integer add3 issued as one unit: {m, c} = {max3(m0, x, y), c0 + i + j}.  One asm statement so that
// the scheduler cannot drift the adds away from the max (which would keep every staged flag register
// alive across the emit blocks).
__device__ __forceinline__ void vmax3_add3(float& m, int& c, float m0, float x, float y, int c0, int i, int j) {
  asm("v_max3_f32 %0, %2, %3, %4\n\tv_add3_u32 %1, %5, %6, %7"
      : "=&v"(m), "=v"(c)
      : "v"(m0), "v"(x), "v"(y), "v"(c0), "v"(i), "v"(j));
}
// running max and min of the same operands as one unit (same reason as vmax3_add3)
__device__ __forceinline__ void vmax3_min3(float& mx, float& mn, float mx0, float mn0, float x, float y) {
  asm("v_max3_f32 %0, %2, %4, %5\n\tv_min3_f32 %1, %3, %4, %5" : "=&v"(mx), "=v"(mn) : "v"(mx0), "v"(mn0), "v"(x), "v"(y));
}
__device__ __forceinline__ void vmax3_min3(float& mx, float& mn, float mx0, float mn0, float x1, float x2, float n1,
                                           float n2) {
  asm("v_max3_f32 %0, %2, %4, %5\n\tv_min3_f32 %1, %3, %6, %7"
      : "=&v"(mx), "=v"(mn)
      : "v"(mx0), "v"(mn0), "v"(x1), "v"(x2), "v"(n1), "v"(n2));
}
__device__ __forceinline__ float vmax2_zero(float a) {  // max(a, 0) ignoring NaN (NaN -> 0)
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(a));
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
