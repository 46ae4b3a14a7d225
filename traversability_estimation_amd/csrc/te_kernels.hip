// te_kernels.hip -- gfx950 (CDNA4) kernels of the traversability filter chain.
//
// Built with -ffp-contract=off: every fused multiply-add below is written explicitly (fma()), so
// the handful of places that must reproduce the reference's un-fused double arithmetic (cell
// positions and the CircleIterator membership test for tie radii) stay un-fused.
//
// Data layout: every layer is [batch][cols][rows] float32 == grid_map's column-major matrices;
// the fast (coalesced) axis is the grid_map row index i.  Invalid cell == non-finite.
//
// What each kernel restates (paths relative to the reference root):
//   k_step_height   StepFilter::update first pass      traversability_estimation_filters/src/StepFilter.cpp:112-144
//   k_step_score    StepFilter::update second pass     StepFilter.cpp:147-178
//   k_normals       NormalVectorsFilter (area method, un-vendored grid_map_filters)
//                   + SlopeFilter::update              SlopeFilter.cpp:59-88
//                   + RoughnessFilter::update          RoughnessFilter.cpp:73-132
//   k_combine       MathExpressionFilter fixed form    traversability_estimation/config/robot_filter_parameter.yaml:29-33
#include "te_internal.h"

#include <math.h>

namespace te {

namespace {

constexpr int TX = 64;  // tile extent along the fast axis (grid_map row index i): one wavefront
constexpr int TY = 16;  // tile extent along the slow axis (grid_map column index j)
constexpr int BY = 4;   // threads along j; each thread owns TY/BY cells
constexpr int CPT = TY / BY;

__device__ __forceinline__ float sanitize(float v) { return __builtin_isfinite(v) ? v : __builtin_nanf(""); }

// Stage the (TX+2K) x (TY+2K) neighbourhood of the tile at (i0, j0) into LDS, NaN outside the map
// (CircleIterator clamps at the borders: cells outside simply do not exist).
__device__ __forceinline__ void load_tile(float* __restrict__ tile, const float* __restrict__ layer, const Geo& g,
                                          int i0, int j0, int K) {
  const int tw = TX + 2 * K, th = TY + 2 * K;
  for (int tj = threadIdx.y; tj < th; tj += BY) {
    const int j = j0 - K + tj;
    const bool jin = (j >= 0) && (j < g.cols);
    const float* row = layer + (size_t)(jin ? j : 0) * g.rows;
    for (int ti = threadIdx.x; ti < tw; ti += TX) {
      const int i = i0 - K + ti;
      float v = __builtin_nanf("");
      if (jin && i >= 0 && i < g.rows) v = sanitize(row[i]);
      tile[tj * tw + ti] = v;
    }
  }
}

// CircleIterator::isInside for one offset, with the reference's own double arithmetic
// (getPositionFromIndex: position = (mapPosition + (0.5*length - 0.5*res)) + res*(-index)).
__device__ __forceinline__ bool tie_inside(const Geo& g, double r2, int i, int j, int di, int dj) {
  const double cx = g.ax + g.res * (double)(-i);
  const double cy = g.ay + g.res * (double)(-j);
  const double x = g.ax + g.res * (double)(-(i + di));
  const double y = g.ay + g.res * (double)(-(j + dj));
  const double dx = x - cx, dy = y - cy;
  return dx * dx + dy * dy <= r2;
}

// ------------------------------------------------------------------------------------------------
// StepFilter pass 1: step_height = max - min of the valid elevations in the first window
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TX* BY) void k_step_height(Geo g, Disc d, const float* __restrict__ elev,
                                                        float* __restrict__ sh, Region rg) {
  extern __shared__ float tile[];
  const int map = rg.map >= 0 ? rg.map : blockIdx.z;
  const size_t mo = (size_t)map * g.rows * g.cols;
  const int i0 = rg.i0 + blockIdx.x * TX, j0 = rg.j0 + blockIdx.y * TY;
  const int K = d.reach;
  const int tw = TX + 2 * K;
  load_tile(tile, elev + mo, g, i0, j0, K);
  __syncthreads();
  const int i = i0 + threadIdx.x;
  if (i >= rg.i1) return;
#pragma unroll 1
  for (int c = 0; c < CPT; ++c) {
    const int lj = threadIdx.y + c * BY;
    const int j = j0 + lj;
    if (j >= rg.j1) break;
    const float* ctr = tile + (lj + K) * tw + (threadIdx.x + K);
    const float z0 = *ctr;
    float out = __builtin_nanf("");
    if (z0 == z0) {  // StepFilter.cpp:113 only cells with a valid elevation
      float mn = z0, mx = z0;
      for (int dj = -d.R; dj <= d.R; ++dj) {
        const int hw = d.hw[dj < 0 ? -dj : dj];
        const float* row = ctr + dj * tw;
        for (int di = -hw; di <= hw; ++di) {
          const float z = row[di];
          mn = fminf(mn, z);  // fminf/fmaxf ignore NaN == the isValid() skip, :126
          mx = fmaxf(mx, z);
        }
      }
      for (int t = 0; t < d.n_ties; ++t) {
        const int di = d.tie_di[t], dj = d.tie_dj[t];
        if (tie_inside(g, d.r2, i, j, di, dj)) {
          const float z = ctr[dj * tw + di];
          mn = fminf(mn, z);
          mx = fmaxf(mx, z);
        }
      }
      out = (float)((double)mx - (double)mn);  // :143 double difference stored as float
    }
    sh[mo + (size_t)j * g.rows + i] = out;
  }
}

// ------------------------------------------------------------------------------------------------
// StepFilter pass 2
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float step_score(float smax, int ncells, bool valid, double crit, int ncrit) {
  if (!valid) return __builtin_nanf("");
  const double sm = (double)smax;
  const double a1 = (double)ncells / (double)ncrit * sm;
  const double step = sm < a1 ? sm : a1;  // std::min(stepMax, nCells/nCrit*stepMax), StepFilter.cpp:170
  return step < crit ? (float)(1.0 - step / crit) : 0.0f;
}

__global__ __launch_bounds__(TX* BY) void k_step_score(Geo g, Disc d, double crit, int ncrit,
                                                       const float* __restrict__ sh, float* __restrict__ out,
                                                       Region rg) {
  extern __shared__ float tile[];
  const int map = rg.map >= 0 ? rg.map : blockIdx.z;
  const size_t mo = (size_t)map * g.rows * g.cols;
  const int i0 = rg.i0 + blockIdx.x * TX, j0 = rg.j0 + blockIdx.y * TY;
  const int K = d.reach;
  const int tw = TX + 2 * K;
  load_tile(tile, sh + mo, g, i0, j0, K);
  __syncthreads();
  const int i = i0 + threadIdx.x;
  if (i >= rg.i1) return;
  const float critf_lo = (float)crit;  // only used as a quick filter; the exact test is the double compare
  (void)critf_lo;
#pragma unroll 1
  for (int c = 0; c < CPT; ++c) {
    const int lj = threadIdx.y + c * BY;
    const int j = j0 + lj;
    if (j >= rg.j1) break;
    const float* ctr = tile + (lj + K) * tw + (threadIdx.x + K);
    float smax = 0.0f;  // stepMax starts at 0.0, :149
    int ncells = 0;
    bool valid = false;
    for (int dj = -d.R; dj <= d.R; ++dj) {
      const int hw = d.hw[dj < 0 ? -dj : dj];
      const float* row = ctr + dj * tw;
      for (int di = -hw; di <= hw; ++di) {
        const float s = row[di];
        valid |= (s == s);
        smax = fmaxf(smax, s);
        ncells += ((double)s > crit) ? 1 : 0;  // NaN compares false
      }
    }
    for (int t = 0; t < d.n_ties; ++t) {
      const int di = d.tie_di[t], dj = d.tie_dj[t];
      if (tie_inside(g, d.r2, i, j, di, dj)) {
        const float s = ctr[dj * tw + di];
        valid |= (s == s);
        smax = fmaxf(smax, s);
        ncells += ((double)s > crit) ? 1 : 0;
      }
    }
    out[mo + (size_t)j * g.rows + i] = step_score(smax, ncells, valid, crit, ncrit);
  }
}

// ------------------------------------------------------------------------------------------------
// Normals + slope + roughness
// ------------------------------------------------------------------------------------------------
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

__device__ __forceinline__ void accumulate_disc(Mom& m, const Geo& g, const Disc& d, const float* ctr, int tw, int i,
                                                int j, double z0) {
  for (int dj = -d.R; dj <= d.R; ++dj) {
    const int hw = d.hw[dj < 0 ? -dj : dj];
    const float* row = ctr + dj * tw;
    for (int di = -hw; di <= hw; ++di) mom_add(m, di, dj, row[di], z0);
  }
  for (int t = 0; t < d.n_ties; ++t) {
    const int di = d.tie_di[t], dj = d.tie_dj[t];
    if (tie_inside(g, d.r2, i, j, di, dj)) mom_add(m, di, dj, ctr[dj * tw + di], z0);
  }
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
__device__ __noinline__ void smallest_eigvec(const double c[6], double nrm[3], double& lambda1) {
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

__global__ __launch_bounds__(TX* BY) void k_normals(Geo g, Disc dn, Disc dr, int same_disc, int axis,
                                                    double slope_crit, double rough_crit,
                                                    const float* __restrict__ elev, float* __restrict__ slope,
                                                    float* __restrict__ rough, float* __restrict__ onx,
                                                    float* __restrict__ ony, float* __restrict__ onz, Region rg) {
  extern __shared__ float tile[];
  const int map = rg.map >= 0 ? rg.map : blockIdx.z;
  const size_t mo = (size_t)map * g.rows * g.cols;
  const int i0 = rg.i0 + blockIdx.x * TX, j0 = rg.j0 + blockIdx.y * TY;
  const int K = dn.reach > dr.reach ? dn.reach : dr.reach;
  const int tw = TX + 2 * K;
  load_tile(tile, elev + mo, g, i0, j0, K);
  __syncthreads();
  const int i = i0 + threadIdx.x;
  if (i >= rg.i1) return;
#pragma unroll 1
  for (int c = 0; c < CPT; ++c) {
    const int lj = threadIdx.y + c * BY;
    const int j = j0 + lj;
    if (j >= rg.j1) break;
    const float* ctr = tile + (lj + K) * tw + (threadIdx.x + K);
    const float z0f = *ctr;
    const float qnan = __builtin_nanf("");
    float o_slope = qnan, o_rough = qnan, nf[3] = {qnan, qnan, qnan};
    if (z0f == z0f) {  // normals only where the input layer is valid; slope/roughness follow (SlopeFilter.cpp:71, RoughnessFilter.cpp:84)
      const double z0 = (double)z0f;
      Mom m;
      double cov[6];
      mom_zero(m);
      accumulate_disc(m, g, dn, ctr, tw, i, j, z0);
      covariance(m, g.res, cov);
      normal_from_cov(m, cov, axis, nf);
      o_slope = slope_score(nf[2], slope_crit);
      if (!same_disc) {
        mom_zero(m);
        accumulate_disc(m, g, dr, ctr, tw, i, j, z0);
        covariance(m, g.res, cov);
      }
      o_rough = roughness_score(m, cov, nf, rough_crit);
    }
    const size_t o = mo + (size_t)j * g.rows + i;
    slope[o] = o_slope;
    rough[o] = o_rough;
    if (onx) {
      onx[o] = nf[0];
      ony[o] = nf[1];
      onz[o] = nf[2];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// MathExpressionFilter, fixed form, float32, left to right (bit-exact with the shipped expression)
// ------------------------------------------------------------------------------------------------
__global__ void k_combine(Geo g, float w_scale, float w_slope, float w_step, float w_rough,
                          const float* __restrict__ slope, const float* __restrict__ step,
                          const float* __restrict__ rough, float* __restrict__ trav, Region rg) {
  const int map = rg.map >= 0 ? rg.map : blockIdx.z;
  const size_t mo = (size_t)map * g.rows * g.cols;
  const int i = rg.i0 + blockIdx.x * blockDim.x + threadIdx.x;
  const int j = rg.j0 + blockIdx.y;
  if (i >= rg.i1 || j >= rg.j1) return;
  const size_t o = mo + (size_t)j * g.rows + i;
  const float a = w_slope * slope[o];
  const float b = w_step * step[o];
  const float c = w_rough * rough[o];
  const float ab = a + b;
  const float abc = ab + c;
  trav[o] = w_scale * abc;
}

inline Region clamp_region(const Geo& g, const Region& r, int grow) {
  Region o = r;
  o.i0 = r.i0 - grow < 0 ? 0 : r.i0 - grow;
  o.j0 = r.j0 - grow < 0 ? 0 : r.j0 - grow;
  o.i1 = r.i1 + grow > g.rows ? g.rows : r.i1 + grow;
  o.j1 = r.j1 + grow > g.cols ? g.cols : r.j1 + grow;
  return o;
}

inline dim3 tile_grid(const Geo& g, const Region& r) {
  return dim3((unsigned)((r.i1 - r.i0 + TX - 1) / TX), (unsigned)((r.j1 - r.j0 + TY - 1) / TY),
              (unsigned)(r.map >= 0 ? 1 : g.batch));
}

inline size_t tile_bytes(int K) { return (size_t)(TX + 2 * K) * (TY + 2 * K) * sizeof(float); }

}  // namespace

int chain_max_reach(const ChainParams& p) {
  int a = p.normals.reach > p.rough.reach ? p.normals.reach : p.rough.reach;
  int b = p.step1.reach + p.step2.reach;
  return a > b ? a : b;
}

hipError_t launch_chain(const Geo& g, const ChainParams& p, const Layers& L, const Region& r, unsigned flags,
                        hipStream_t stream) {
  if (r.i1 <= r.i0 || r.j1 <= r.j0) return hipSuccess;
  const bool whole = (r.map < 0);
  const dim3 blk(TX, BY);
  // StepFilter pass 1 on the region dilated by the first window, pass 2 on first+second.
  const Region r1 = whole ? r : clamp_region(g, r, p.step1.reach);
  const Region r2 = whole ? r : clamp_region(g, r, p.step1.reach + p.step2.reach);
  const Region rn = whole ? r : clamp_region(g, r, p.normals.reach > p.rough.reach ? p.normals.reach : p.rough.reach);
  const Region rc = whole ? r : clamp_region(g, r, chain_max_reach(p));
  hipLaunchKernelGGL(k_step_height, tile_grid(g, r1), blk, tile_bytes(p.step1.reach), stream, g, p.step1, L.elev,
                     L.step_height, r1);
  hipLaunchKernelGGL(k_step_score, tile_grid(g, r2), blk, tile_bytes(p.step2.reach), stream, g, p.step2, p.step_crit,
                     p.step_ncrit, L.step_height, L.step, r2);
  const bool keep = (flags & TE_RUN_KEEP_NORMALS) != 0;
  const int Kn = p.normals.reach > p.rough.reach ? p.normals.reach : p.rough.reach;
  hipLaunchKernelGGL(k_normals, tile_grid(g, rn), blk, tile_bytes(Kn), stream, g, p.normals, p.rough,
                     p.same_rough_disc, p.axis, p.slope_crit, p.rough_crit, L.elev, L.slope, L.rough,
                     keep ? L.nx : nullptr, keep ? L.ny : nullptr, keep ? L.nz : nullptr, rn);
  const dim3 cgrid((unsigned)((rc.i1 - rc.i0 + 255) / 256), (unsigned)(rc.j1 - rc.j0),
                   (unsigned)(rc.map >= 0 ? 1 : g.batch));
  hipLaunchKernelGGL(k_combine, cgrid, dim3(256), 0, stream, g, p.w_scale, p.w_slope, p.w_step, p.w_rough, L.slope,
                     L.step, L.rough, L.trav, rc);
  return hipGetLastError();
}

}  // namespace te
