// te_eig.h -- the fast general tail shared by the sliding-disc kernel (discs clipped by the map border) and the
// fix-up pass: Newton-refined reciprocal / square roots, polynomial acos, and the smallest eigenpair of the 3x3
// covariance from the neighbourhood moments via one Jacobi rotation + the secular equation.
#pragma once
#include "te_internal.h"

namespace te {
namespace fast {

// a raw (unclipped) score 1 - x / crit that lies within `band` of the clip at 0: see kExactNaNBits (te_internal.h)
__device__ __forceinline__ bool near_clip(float raw, float band) { return __builtin_fabsf(raw) < band; }
__device__ __forceinline__ float exact_nanf() { return __builtin_bit_cast(float, kExactNaNBits); }

__device__ __forceinline__ double rsqrt_nr(double x) {  // x > 0
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  y = y * fma(-hx * y, y, 1.5);
  y = y * fma(-hx * y, y, 1.5);
  return y;
}
__device__ __forceinline__ double rcp_nr(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = y * fma(-x, y, 2.0);
  y = y * fma(-x, y, 2.0);
  return y;
}
__device__ __forceinline__ double sqrt_nr(double x) { return x > 0.0 ? x * rsqrt_nr(x) : 0.0; }

// acos on [-1, 1], absolute error < 1e-14 (asin(y) = y + y^3 P(y^2) on |y| <= 1/2, degree-9 fit).
__device__ __forceinline__ double acos_poly(double x) {
  const double ax = fabs(x);
  const bool big = ax > 0.5;
  const double u = big ? 0.5 * (1.0 - ax) : ax * ax;
  const double y = big ? sqrt_nr(u) : ax;
  double p = 0.027906776267349036;
  p = fma(p, u, -0.0029394830760080953);
  p = fma(p, u, 0.015675506169091535);
  p = fma(p, u, 0.013187958701109649);
  p = fma(p, u, 0.017441488900185986);
  p = fma(p, u, 0.022366066582034674);
  p = fma(p, u, 0.03038218274113012);
  p = fma(p, u, 0.04464285243893878);
  p = fma(p, u, 0.07500000003583389);
  p = fma(p, u, 0.16666666666662183);
  const double as = fma(y * u, p, y);
  const double r = big ? 2.0 * as : (1.5707963267948966 - as) + 6.123233995736766e-17;
  return x < 0.0 ? 3.141592653589793 - r : r;
}

// float32 acos on [-1, 1]: same reduction, degree-5 fit (5e-10) + float rounding; 1 - |x| is exact for
// |x| >= 1/2, so the result keeps its RELATIVE accuracy for near-flat cells (acos -> 0).
__device__ __forceinline__ float acosf_poly(float x) {
  const float ax = fabsf(x);
  const bool big = ax > 0.5f;
  const float u = big ? 0.5f * (1.0f - ax) : ax * ax;
  const float y = big ? __builtin_amdgcn_sqrtf(u) : ax;
  float p = 3.369084721e-02f;
  p = fmaf(p, u, 1.714923835e-02f);
  p = fmaf(p, u, 3.110066274e-02f);
  p = fmaf(p, u, 4.459940153e-02f);
  p = fmaf(p, u, 7.500094543e-02f);
  p = fmaf(p, u, 1.666666634e-01f);
  const float as = fmaf(y * u, p, y);
  const float r = big ? 2.0f * as : 1.57079637f - as;
  return x < 0.0f ? 3.14159274f - r : r;
}

// the same for 0 <= x <= 1 (a normal turned towards +z)
__device__ __forceinline__ float acosf_poly01(float x) {
  const bool big = x > 0.5f;
  const float u = big ? 0.5f * (1.0f - x) : x * x;
  const float y = big ? __builtin_amdgcn_sqrtf(u) : x;
  float p = 3.369084721e-02f;
  p = fmaf(p, u, 1.714923835e-02f);
  p = fmaf(p, u, 3.110066274e-02f);
  p = fmaf(p, u, 4.459940153e-02f);
  p = fmaf(p, u, 7.500094543e-02f);
  p = fmaf(p, u, 1.666666634e-01f);
  const float as = fmaf(y * u, p, y);
  return big ? 2.0f * as : 1.57079637f - as;
}

// General tail for a disc clipped by the map border (or any validity pattern whose x/y moments are
// known): population covariance from the moments, smallest eigenpair of the 3x3 via a Jacobi
// rotation of the x/y block and a safeguarded Newton iteration on the secular equation of the
// resulting arrow matrix.  Returns false for the (measure-zero) configurations it does not
// resolve; the caller leaves those cells to the fix-up pass.  q_out = n^T C n with the float32 normal.
__device__ __forceinline__ bool border_tail(double res, int n, int si, int sj, int sii, int sij, int sjj, double Sz,
                                            double Siz, double Sjz, double Szz, float& nx, float& ny, float& nz,
                                            double& q_out) {
  if (n < 1) return false;
  const double dn = (double)n;
  const double inv_n2 = 1.0 / (dn * dn);
  const double r2 = res * res;
  const double cxx = r2 * (double)((long long)n * sii - (long long)si * si) * inv_n2;
  const double cxy = r2 * (double)((long long)n * sij - (long long)si * sj) * inv_n2;
  const double cyy = r2 * (double)((long long)n * sjj - (long long)sj * sj) * inv_n2;
  const double cxz = -res * fma(dn, Siz, -(double)si * Sz) * inv_n2;
  const double cyz = -res * fma(dn, Sjz, -(double)sj * Sz) * inv_n2;
  const double czz = fma(dn, Szz, -Sz * Sz) * inv_n2;
  double vx = 0.0, vy = 0.0, vz = 1.0;
  if (n >= 3) {
    double cs = 1.0, sn = 0.0, mu1 = cxx, mu2 = cyy;
    if (cxy != 0.0) {
      const double tau = (cyy - cxx) / (2.0 * cxy);
      double t = 1.0 / (fabs(tau) + sqrt(fma(tau, tau, 1.0)));
      t = tau < 0.0 ? -t : t;
      cs = 1.0 / sqrt(fma(t, t, 1.0));
      sn = t * cs;
      mu1 = cxx - t * cxy;
      mu2 = cyy + t * cxy;
    }
    const double ap = cs * cxz - sn * cyz, bp = sn * cxz + cs * cyz;
    const double ra = ap * ap, rb = bp * bp;
    double lam;
    if (ra == 0.0 && rb == 0.0) {
      if (!(czz <= mu1 && czz <= mu2)) return false;  // horizontal normal: leave to the general solver
      lam = czz;
    } else {
      double hi = czz;
      if (ra != 0.0 && mu1 < hi) hi = mu1;
      if (rb != 0.0 && mu2 < hi) hi = mu2;
      double lo = hi < 0.0 ? hi - fabs(hi) - 1e-300 : 0.0;
      lo = lo < -fabs(hi) ? lo : -fabs(hi);  // f(lo) >= 0 for a PSD matrix up to rounding
      lam = hi > 0.0 ? 0.0 : lo;
      double hb = hi;
      for (int it = 0; it < 40; ++it) {
        const double r1 = ra != 0.0 ? rcp_nr(mu1 - lam) : 0.0;
        const double r2_ = rb != 0.0 ? rcp_nr(mu2 - lam) : 0.0;
        const double f = (czz - lam) - ra * r1 - rb * r2_;
        const double fp = -1.0 - ra * r1 * r1 - rb * r2_ * r2_;
        if (f > 0.0)
          lo = lam;
        else
          hb = lam;
        double ln = lam - f * rcp_nr(fp);
        if (!(ln >= lo && ln <= hb)) ln = 0.5 * (lo + hb);
        const bool conv = fabs(ln - lam) <= 1e-15 * fabs(ln) || f == 0.0;
        lam = ln;
        if (__all(conv)) break;
      }
      if (ra == 0.0 && mu1 < lam) return false;
      if (rb == 0.0 && mu2 < lam) return false;
    }
    // middle eigenvalue from the invariants (NormalVectorsFilter keeps the eigenvector only if it is > 1e-8)
    const double tr = mu1 + mu2 + czz;
    const double c1 = mu1 * mu2 + (mu1 + mu2) * czz - ra - rb;
    const double s12 = tr - lam;
    const double p12 = c1 - lam * s12;
    double dsc = fma(s12, s12, -4.0 * p12);
    dsc = dsc > 0.0 ? dsc : 0.0;
    const double bigr = 0.5 * (s12 + sqrt(dsc));
    const double lam1 = bigr > 0.0 ? p12 / bigr : 0.0;
    if (lam1 > 1e-8) {
      const double v1 = ra != 0.0 ? ap / (lam - mu1) : 0.0;
      const double v2 = rb != 0.0 ? bp / (lam - mu2) : 0.0;
      const double wx = cs * v1 + sn * v2, wy = -sn * v1 + cs * v2;
      const double inv = 1.0 / sqrt(fma(wx, wx, fma(wy, wy, 1.0)));
      if (!(inv > 0.0)) return false;
      vx = wx * inv;
      vy = wy * inv;
      vz = inv;
    }
  }
  nx = (float)vx;
  ny = (float)vy;
  nz = (float)vz;
  const double x = (double)nx, y = (double)ny, z = (double)nz;
  double q = fma(cxx * x, x, fma(2.0 * cxy * x, y, fma(cyy * y, y, fma(2.0 * z, fma(cxz, x, cyz * y), czz * z * z))));
  q_out = q > 0.0 ? q : 0.0;
  return true;
}


}  // namespace fast
}  // namespace te
