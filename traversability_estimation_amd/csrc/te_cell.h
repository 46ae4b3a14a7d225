// te_cell.h -- per-cell arithmetic shared by the generic and the shape-specialised kernels:
// neighbourhood moments -> covariance -> surface normal (NormalVectorsFilter, un-vendored
// grid_map_filters) -> slope score (SlopeFilter.cpp:59-88) and roughness score (RoughnessFilter.cpp:73-132).
#pragma once
#include <math.h>

#include "te_internal.h"

namespace te {

// Neighbourhood moments in CENTRE-LOCAL coordinates: offsets (di, dj) are exact integers and
// dz = z - z_centre is exact in double, so the covariance has none of the cancellation of the
// reference's absolute-coordinate sums (it is translation invariant, so it is the same matrix).
struct Mom {
  int n, si, sj, sii, sij, sjj;
  double sz, siz, sjz, szz;
};

__device__ __forceinline__ void mom_zero(Mom& m) {
  m.n = m.si = m.sj = m.sii = m.sij = m.sjj = 0;
  m.sz = m.siz = m.sjz = m.szz = 0.0;
}

__device__ __forceinline__ void mom_add(Mom& m, int di, int dj, float z, double z0) {
  const bool v = (z == z);
  const double dz = v ? (double)z - z0 : 0.0;
  const int w = v ? 1 : 0;
  m.n += w;
  m.si += w * di;
  m.sj += w * dj;
  m.sii += w * di * di;
  m.sij += w * di * dj;
  m.sjj += w * dj * dj;
  m.sz += dz;
  m.siz = fma((double)di, dz, m.siz);
  m.sjz = fma((double)dj, dz, m.sjz);
  m.szz = fma(dz, dz, m.szz);
}

// Population covariance of the points (x, y, z) = (-res*di, -res*dj, dz) (x and y DEcrease with the
// indices, getPositionFromIndex).  c = {xx, xy, xz, yy, yz, zz}.
__device__ __forceinline__ void covariance(const Mom& m, double res, double c[6]) {
  const double n = (double)m.n;
  const double inv_n2 = 1.0 / (n * n);
  // integer central moments are exact
  const double cii = (double)((long long)m.n * m.sii - (long long)m.si * m.si);
  const double cij = (double)((long long)m.n * m.sij - (long long)m.si * m.sj);
  const double cjj = (double)((long long)m.n * m.sjj - (long long)m.sj * m.sj);
  const double ciz = fma(n, m.siz, -(double)m.si * m.sz);
  const double cjz = fma(n, m.sjz, -(double)m.sj * m.sz);
  const double czz = fma(n, m.szz, -m.sz * m.sz);
  const double r2 = res * res;
  c[0] = r2 * cii * inv_n2;
  c[1] = r2 * cij * inv_n2;
  c[2] = -res * ciz * inv_n2;
  c[3] = r2 * cjj * inv_n2;
  c[4] = -res * cjz * inv_n2;
  c[5] = czz * inv_n2;
}

// One Jacobi rotation annihilating a_pq of a symmetric 3x3 (r = the third index).
__device__ __forceinline__ void jacobi_rot(double& app, double& aqq, double& apq, double& arp, double& arq, double& v0p,
                                           double& v0q, double& v1p, double& v1q, double& v2p, double& v2q) {
  if (apq == 0.0) return;
  const double h = aqq - app;
  const double g100 = 100.0 * fabs(apq);
  double t;
  if (fabs(h) + g100 == fabs(h)) {
    t = apq / h;
  } else {
    const double theta = 0.5 * h / apq;
    t = 1.0 / (fabs(theta) + sqrt(fma(theta, theta, 1.0)));
    t = theta < 0.0 ? -t : t;
  }
  const double c = 1.0 / sqrt(fma(t, t, 1.0));
  const double s = t * c;
  const double tau = s / (1.0 + c);
  app -= t * apq;
  aqq += t * apq;
  apq = 0.0;
  const double rp = arp, rq = arq;
  arp = rp - s * fma(rp, tau, rq);
  arq = rq + s * fma(-rq, tau, rp);
  double a, b;
  a = v0p; b = v0q; v0p = a - s * fma(a, tau, b); v0q = b + s * fma(-b, tau, a);
  a = v1p; b = v1q; v1p = a - s * fma(a, tau, b); v1q = b + s * fma(-b, tau, a);
  a = v2p; b = v2q; v2p = a - s * fma(a, tau, b); v2q = b + s * fma(-b, tau, a);
}

// Eigenvector of the smallest eigenvalue of the symmetric matrix c (cyclic Jacobi, double) and the
// middle eigenvalue (NormalVectorsFilter keeps the eigenvector only if eigenvalues(1) > 1e-8).
static __device__ __noinline__ void smallest_eigvec(const double c[6], double nrm[3], double& lambda1) {
  double a00 = c[0], a01 = c[1], a02 = c[2], a11 = c[3], a12 = c[4], a22 = c[5];
  double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
#pragma unroll 1
  for (int sweep = 0; sweep < 32; ++sweep) {
    const double off = fabs(a01) + fabs(a02) + fabs(a12);
    const double dia = fabs(a00) + fabs(a11) + fabs(a22);
    if (off == 0.0 || (sweep > 3 && dia + 100.0 * off == dia)) break;
    jacobi_rot(a00, a11, a01, a02, a12, v00, v01, v10, v11, v20, v21);  // (p,q)=(0,1), r=2
    jacobi_rot(a00, a22, a02, a01, a12, v00, v02, v10, v12, v20, v22);  // (0,2), r=1
    jacobi_rot(a11, a22, a12, a01, a02, v01, v02, v11, v12, v21, v22);  // (1,2), r=0
  }
  // ascending order, first minimum wins (Eigen's selection sort)
  double w0 = a00, w1 = a11, w2 = a22;
  double x0 = v00, x1 = v10, x2 = v20;  // column 0
  double y0 = v01, y1 = v11, y2 = v21;  // column 1
  double z0 = v02, z1 = v12, z2 = v22;  // column 2
  // smallest -> slot 0
  if (w1 < w0 && w1 <= w2) {
    double t;
    t = w0; w0 = w1; w1 = t;
    t = x0; x0 = y0; y0 = t; t = x1; x1 = y1; y1 = t; t = x2; x2 = y2; y2 = t;
  } else if (w2 < w0 && w2 < w1) {
    double t;
    t = w0; w0 = w2; w2 = t;
    t = x0; x0 = z0; z0 = t; t = x1; x1 = z1; z1 = t; t = x2; x2 = z2; z2 = t;
  }
  lambda1 = w2 < w1 ? w2 : w1;
  nrm[0] = x0;
  nrm[1] = x1;
  nrm[2] = x2;
}

// nPoints < 3 or second eigenvalue <= 1e-8 -> UnitZ; flip towards the positive axis; round to
// float32 exactly where the reference stores the surface_normal_* layers.
__device__ __forceinline__ void normal_from_cov(const Mom& m, const double c[6], int axis, float nf[3]) {
  double nv[3] = {0.0, 0.0, 1.0};
  if (m.n >= 3) {
    double ev[3], l1;
    smallest_eigvec(c, ev, l1);
    if (l1 > 1e-8) {
      nv[0] = ev[0];
      nv[1] = ev[1];
      nv[2] = ev[2];
    }
  }
  const double dot = axis == 0 ? nv[0] : (axis == 1 ? nv[1] : nv[2]);
  const double sgn = dot < 0.0 ? -1.0 : 1.0;
  nf[0] = (float)(sgn * nv[0]);
  nf[1] = (float)(sgn * nv[1]);
  nf[2] = (float)(sgn * nv[2]);
}

// TE_OPT_NORMALS_RANK_RULE: is the scatter matrix of the disc rank-deficient (its points exactly on a plane or a line)?
// NormalVectorsFilter up to grid_map 1.6 -- the filter that wrote the reference's bag; from memory, the library is not
// vendored -- ran its eigen-solver only if covarianceMatrix.fullPivHouseholderQr().rank() >= 3 and returned UnitZ otherwise
// (oracle/te_oracle.c: teo_set_normals_rank_rule restates it on the centred points).  Here: full pivoting on the covariance
// from the centre-local moments (rank is scale-invariant); a pivot counts if it exceeds 1e-12 of the first.  On exactly
// planar data the third pivot is rounding noise of the moment form, ~1e-16 of the first (the oracle's centred form: 0);
// on the bag every other disc has >= 1.4e-3, and a disc of float32 terrain whose residual is below 1e-6 of its extent
// does not occur -- the threshold decides nothing.
__device__ __forceinline__ bool rank_deficient(const double c[6]) {
  double S[3][3] = {{c[0], c[1], c[2]}, {c[1], c[3], c[4]}, {c[2], c[4], c[5]}};
  double first = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    int pr = k, pc = k;
    double best = -1.0;
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int v = 0; v < 3; ++v)
        if (u >= k && v >= k && fabs(S[u][v]) > best) {
          best = fabs(S[u][v]);
          pr = u;
          pc = v;
        }
    if (k == 0) first = best;
    if (!(best > 1e-12 * first) || best == 0.0) return true;
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      const double t = S[k][v];
      S[k][v] = S[pr][v];
      S[pr][v] = t;
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const double t = S[u][k];
      S[u][k] = S[u][pc];
      S[u][pc] = t;
    }
#pragma unroll
    for (int u = 0; u < 3; ++u)
      if (u > k) {
        const double f = S[u][k] / S[k][k];
#pragma unroll
        for (int v = 0; v < 3; ++v)
          if (v >= k) S[u][v] -= f * S[k][v];
      }
  }
  return false;
}

__device__ __forceinline__ float slope_score(float nz, double crit) {
  const double slope = acos((double)nz);  // SlopeFilter.cpp:74
  return slope < crit ? (float)(1.0 - slope / crit) : 0.0f;
}

// RoughnessFilter.cpp:105-124 from the neighbourhood moments:
//   sum_i (n.(p_i - mean))^2 = N * n^T C n   with C the population covariance, n the float32 normal.
__device__ __forceinline__ float roughness_score(const Mom& m, const double c[6], const float nf[3], double crit) {
  if (m.n < 2) return 0.0f;  // n == 1: 0/0 = NaN -> "roughness < crit" false -> 0.0
  const double a = (double)nf[0], b = (double)nf[1], cc = (double)nf[2];
  const double q0 = fma(c[0], a, fma(c[1], b, c[2] * cc));
  const double q1 = fma(c[1], a, fma(c[3], b, c[4] * cc));
  const double q2 = fma(c[2], a, fma(c[4], b, c[5] * cc));
  double q = fma(a, q0, fma(b, q1, cc * q2));
  q = q > 0.0 ? q : 0.0;
  const double rough = sqrt(q * (double)m.n / (double)(m.n - 1));
  return rough < crit ? (float)(1.0 - rough / crit) : 0.0f;
}

}  // namespace te
