// te_eig3.h -- lean general tail of the normals filter: smallest eigenpair of the 3x3 covariance of ANY neighbourhood
// (disc clipped by the map border, invalid cells) from its moments, without a division or a square root in double.
//
// NormalVectorsFilter (area method, un-vendored grid_map_filters; call site robot_filter_parameter.yaml:3-9):
//   n < 3 -> UnitZ; covariance = sum(p p^T)/n - mean mean^T; eigenvector of the smallest eigenvalue if the middle
//   eigenvalue is > 1e-8, else UnitZ; flipped to the positive axis (z here).
//
// Everything is scaled by n^2 (C' = n^2 C has integer x/y entries times res^2): no 1/n^2.
//   * smallest eigenvalue: Newton on the characteristic cubic p(l) = det - c1 l + tr l^2 - l^3 from l = 0.  p is convex
//     and decreasing on [0, l0], so the iteration rises monotonically to the smallest root and converges quadratically;
//     on terrain l0 << l1, l2 and p is almost linear there (2-3 steps).  The coefficient rounding (a few ulp of |C|^3
//     in det) moves l0 by ~1e-16 |C|, which is what the eigenvector needs (roughness does not use l0, see below).
//   * eigenvector: cross product of the x and y rows of C' - l0 I: (B E - D (c - l0), D B - (a - l0) E,
//     (a - l0)(c - l0) - B^2).  Its z component is the determinant of the shifted x/y block, positive and well
//     conditioned unless the normal is (nearly) horizontal; those cells are returned as unresolved.
//   * middle eigenvalue test without a square root: l1 l2 = P, l1 + l2 = S  =>  P/S <= l1 <= 2P/S.
//   * nz = vz/|v| is evaluated as sqrt(1 - m), m = (vx^2+vy^2)/|v|^2, with the series in double where float32 rounding
//     of nz decides the slope (te_normals3.hip).
// q = n^T C' n with the float32-rounded normal (RoughnessFilter.cpp:105-117), scaled by n^2 like C'.
#pragma once
#include "te_internal.h"

namespace te {
namespace fast {

// 1/x to ~2^-44 from the float32 reciprocal (x must be inside the float32 range)
__device__ __forceinline__ double rcp_fast(double x) {
  double r = (double)__builtin_amdgcn_rcpf((float)x);
  const double e = fma(-x, r, 1.0);
  return fma(r, e, r);
}
__device__ __forceinline__ double rsqrt_fast(double x) {  // 1/sqrt(x) to ~2^-44
  double y = (double)__builtin_amdgcn_rsqf((float)x);
  const double e = fma(-(x * y), y, 1.0);
  return fma(0.5 * y, e, y);
}

// nz = sqrt(1 - m) rounded to float32 (m = 1 - nz^2 in [0, 1], accurate to ~1e-13 relative)
__device__ __forceinline__ float nz_from_m(double m) {
  double pz = fma(m, 1.0 / 16.0, 0.125);
  pz = fma(m, pz, 0.5);
  const float nz_a = (float)fma(-m, pz, 1.0);  // 1 - m/2 - m^2/8 - m^3/16, |error| < 2^-36 for m < 2^-8
  const float om = (float)(1.0 - m);
  const float nz_b = __builtin_amdgcn_sqrtf(om);
  return om > 0.99609375f ? nz_a : nz_b;
}

// Returns 0: done; 1: unresolved (nearly horizontal normal, ambiguous middle eigenvalue, no convergence, range):
// the caller uses the cyclic Jacobi of te_cell.h.  q_scaled = n^T (n^2 C) n.
// EARLY: the Newton iteration stops as soon as every lane of the wavefront has converged (its last step below 1e-15 of the
// trace: the iteration is monotone and quadratic, on terrain three steps instead of six).  For kernels whose time IS this
// tail (k_normals_small); the marching kernels keep the fixed count -- their register budgets were tuned around it.
template <bool EARLY = false>
__device__ __forceinline__ int general_tail3(double res, int n, int si, int sj, int sii, int sij, int sjj, double Sz, double Siz,
                                             double Sjz, double Szz, float& nx, float& ny, float& nz, double& q_scaled) {
  const double dn = (double)n;
  const double F = fma(dn, Szz, -(Sz * Sz));  // n^2 var(z)
  if (n < 3) {  // UnitZ
    nx = 0.0f;
    ny = 0.0f;
    nz = 1.0f;
    q_scaled = F;
    return 0;
  }
  const double r2 = res * res;
  const double A = r2 * (double)((long long)n * sii - (long long)si * si);
  const double B = r2 * (double)((long long)n * sij - (long long)si * sj);
  const double Cc = r2 * (double)((long long)n * sjj - (long long)sj * sj);
  const double D = -res * fma(dn, Siz, -((double)si * Sz));
  const double E = -res * fma(dn, Sjz, -((double)sj * Sz));
  const double mxy = fma(A, Cc, -(B * B));
  const double mxz = fma(A, F, -(D * D));
  const double myz = fma(Cc, F, -(E * E));
  const double tr = A + Cc + F;
  const double c1 = mxy + mxz + myz;
  // det = F mxy - (Cc D^2 - 2 B D E + A E^2)
  const double det = fma(F, mxy, -fma(Cc * D, D, fma(A * E, E, -2.0 * B * D * E)));
  double lam = 0.0;
  const double tr2 = 2.0 * tr;
  bool ok = c1 > 0.0 && tr < 1e9 && tr > 1e-12;  // the float32 seeds of the reciprocals stay in range
#ifndef TE_EIG3_ITERS
#define TE_EIG3_ITERS 6
#endif
#pragma unroll
  for (int it = 0; it < TE_EIG3_ITERS; ++it) {
    const double p = fma(lam, fma(lam, tr - lam, -c1), det);
    const double dp = fma(lam, fma(-3.0, lam, tr2), -c1);  // < 0 left of the smallest root
    const double step = p * rcp_fast(dp);
    lam = lam - step;
    if constexpr (EARLY) {
      if (__all(!(fabs(step) > 1e-15 * tr))) break;  // (a lane outside the solver's range has step = NaN or tr <= 0: it does not hold the others up)
    }
  }
  lam = lam > 0.0 ? lam : 0.0;  // det rounded below zero on an exactly planar patch
  const double a0 = A - lam, c0 = Cc - lam;
  const double vx = fma(B, E, -(D * c0));
  const double vy = fma(D, B, -(a0 * E));
  const double vz = fma(a0, c0, -(B * B));
  ok = ok && (vz > 1e-4 * fabs(a0 * c0)) && (lam == lam);
  // residual of the third row (it must vanish for an eigenvector): catches a Newton iteration that has not converged
  {
    const double r3 = fma(D, vx, fma(E, vy, (F - lam) * vz));
    ok = ok && (fabs(r3) <= 1e-9 * (fabs(D * vx) + fabs(E * vy) + fabs(F * vz) + lam * vz));
  }
  // middle eigenvalue against NormalVectorsFilter's 1e-8 (scaled by n^2)
  const double S = tr - lam, P = fma(-lam, S, c1);
  const double thr = 1e-8 * dn * dn;
  const bool l1_big = P > thr * S, l1_small = 2.0 * P <= thr * S;
  ok = ok && (l1_big || l1_small);
  const double h2 = fma(vx, vx, vy * vy);
  const double nrm2 = fma(vz, vz, h2);
  const double inv2 = rcp_fast(nrm2);
  const float inv = __builtin_amdgcn_rsqf((float)nrm2);
  float fx = (float)vx * inv, fy = (float)vy * inv, fz = nz_from_m(h2 * inv2);
  if (!l1_big) {
    fx = 0.0f;
    fy = 0.0f;
    fz = 1.0f;
  }
  nx = fx;
  ny = fy;
  nz = fz;
  const double x = (double)fx, y = (double)fy, z = (double)fz;
  const double q = fma(A * x, x, fma(2.0 * B * x, y, fma(Cc * y, y, fma(2.0 * z, fma(D, x, E * y), F * z * z))));
  q_scaled = q > 0.0 ? q : 0.0;
  return ok ? 0 : 1;
}

}  // namespace fast
}  // namespace te
