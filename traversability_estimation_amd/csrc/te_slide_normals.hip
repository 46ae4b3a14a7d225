// te_slide_normals.hip -- NormalVectorsFilter + SlopeFilter + RoughnessFilter (+ the
// MathExpressionFilter combine in the epilogue) as a SLIDING-DISC kernel for gfx950.
//
//   NormalVectorsFilter (area method; un-vendored grid_map_filters, call site
//                        traversability_estimation/config/robot_filter_parameter.yaml:3-9)
//   SlopeFilter::update      traversability_estimation_filters/src/SlopeFilter.cpp:59-88
//   RoughnessFilter::update  traversability_estimation_filters/src/RoughnessFilter.cpp:73-132
//   MathExpressionFilter     robot_filter_parameter.yaml:29-33 (fixed weighted-sum form, float32)
//
// One 64-lane wavefront owns 64 adjacent cells along the fast axis (grid_map row index i) and marches
// down the slow axis j.  Each lane keeps the four z-moments of ITS disc
//     Sz = sum dz,  Siz = sum di*dz,  Sjz = sum dj*dz,  Szz = sum dz^2       (double)
// and moves the disc one row down by adding the 2R+1 cells of its leading edge and removing the
// 2R+1 cells of its trailing edge (column di spans rows -h(di)..h(di)):
//     u = z_lead - z_trail, v = z_lead + z_trail
//     Sz += u;  Siz += di*u;  Szz += u*v;  Sjz += h*v + z_lead;  (then Sjz -= Sz_new)
// i.e. 2(2R+1) LDS reads and ~7(2R+1) flops per cell for a pi*R^2-point stencil, with O(1) state per
// lane.  dz = z - z_ref (one reference per strip) is exact in double and so are the sums of
// Sz/Siz/Sjz (all terms share a quantum far above 2^-52 of their magnitude); Szz rounds at 1e-16.
// Rows are staged once into an LDS ring of doubles; a row is read by 2R+1 lanes x 2 edges.
//
// With every cell of the disc valid the x/y moments are constants of the shape, the covariance is
//   [[c,0,a],[0,c,b],[a,b,d]],  c = res^2*sum(di^2)/N,
// and its smallest eigenpair has a closed form:  delta=(c-d)/2, h2=a^2+b^2, s=sqrt(delta^2+h2),
// t=delta+s:  normal ~ (-a, -b, t).  Rows whose discs can contain an invalid or out-of-map cell (and
// lanes next to the left/right border) are left NaN and flagged; the general kernel recomputes just
// those cells afterwards (te_kernels.hip: k_normals_fixup).  Invalid cells are staged as 0, so the
// exact sums recover as soon as the hole has left the window.
#include "te_internal.h"

namespace te {
namespace fast {

namespace {

constexpr int kLanes = 64;

struct SlideArgs {
  int h[kMaxRadiusCells + 1];  // half-height of disc column |di| (== half-width of row |dj|)
  int np;                      // cells in the disc
  int sii;                     // sum of di^2 over the disc
  int out_rows;                // rows per strip
  double slope_crit, inv_slope_crit, rough_crit, inv_rough_crit;
  float w_scale, w_slope, w_step, w_rough;
  int combine;
};

__device__ __forceinline__ float qnanf() { return __builtin_nanf(""); }

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

constexpr int ring_rows(int R) {
  int n = 4;
  while (n < 2 * R + 2) n *= 2;
  return n;
}

template <int R>
__global__ __launch_bounds__(kLanes) void k_normals_slide(Geo g, SlideArgs a, const float* __restrict__ elev,
                                                          const float* __restrict__ step, float* __restrict__ slope,
                                                          float* __restrict__ rough, float* __restrict__ trav,
                                                          float* __restrict__ onx, float* __restrict__ ony,
                                                          float* __restrict__ onz, int* __restrict__ block_flags,
                                                          Region rg) {
  constexpr int W = kLanes + 2 * R;
  constexpr int NR = ring_rows(R);
  constexpr int NX = (W + kLanes - 1) / kLanes;
  __shared__ double ring[NR * W];
  const int lane = threadIdx.x;
  const int map = rg.map >= 0 ? rg.map : blockIdx.z;
  const size_t mo = (size_t)map * g.rows * g.cols;
  const int i0 = rg.i0 + blockIdx.x * kLanes;
  const int js = rg.j0 + blockIdx.y * a.out_rows;
  const int jend = (js + a.out_rows < rg.j1) ? js + a.out_rows : rg.j1;
  const int i = i0 + lane;
  const float* __restrict__ em = elev + mo;
  const bool border_lane = (i - R < 0) || (i + R >= g.rows);
  const int c = lane + R;  // my column inside a staged row

  // ---- stage one row into the ring; returns whether it contains an invalid in-map cell ----------
  int dirty_until = js - R - 1;  // outputs j <= dirty_until may see an invalid / out-of-map cell
  double zref = 0.0;
  auto load_row = [&](int r, float (&pf)[NX]) {
#pragma unroll
    for (int x = 0; x < NX; ++x) {
      const int cc = lane + x * kLanes;
      const int ci = i0 - R + cc;
      float t = 0.0f;  // out-of-map columns: never read by a non-border lane
      if (cc < W && ci >= 0 && ci < g.rows) t = (r >= 0 && r < g.cols) ? em[(size_t)r * g.rows + ci] : qnanf();
      pf[x] = t;
    }
  };
  auto store_row = [&](int r, const float (&pf)[NX]) {
    bool bad = false;
    double* dst = ring + (r & (NR - 1)) * W;
#pragma unroll
    for (int x = 0; x < NX; ++x) {
      const int cc = lane + x * kLanes;
      const float t = pf[x];
      const bool ok = __builtin_isfinite(t);
      bad |= !ok;
      if (cc < W) dst[cc] = ok ? (double)t - zref : 0.0;
    }
    if (__any(bad)) dirty_until = r + R > dirty_until ? r + R : dirty_until;
  };

  // ---- prologue: rows js-R .. js+R, reference height, direct sum for the first output row --------
  {
    float pf[NX];
    load_row(js - R, pf);
    // reference = first finite value of the first rows (uniform); 0 if there is none yet
    bool found = false;
    for (int r = js - R; r <= js + R; ++r) {
      if (r > js - R) load_row(r, pf);
      if (!found) {
#pragma unroll
        for (int x = 0; x < NX; ++x) {
          const unsigned long long msk = __ballot(__builtin_isfinite(pf[x]) && pf[x] != 0.0f);
          if (!found && msk) {
            zref = (double)__shfl(pf[x], __ffsll((long long)msk) - 1);
            found = true;
          }
        }
      }
      store_row(r, pf);
    }
  }
  // rows staged before the reference was found used zref = 0: restage them if the reference changed
  if (zref != 0.0) {
    float pf[NX];
    for (int r = js - R; r <= js + R; ++r) {
      load_row(r, pf);
      store_row(r, pf);
    }
  }
  __syncthreads();

  double Sz = 0.0, Siz = 0.0, Sjz = 0.0, Szz = 0.0;
  for (int dj = -R; dj <= R; ++dj) {
    const int hw = a.h[dj < 0 ? -dj : dj];
    const double* row = ring + ((js + dj) & (NR - 1)) * W + c;
    double rs = 0.0, ri = 0.0, rq = 0.0;
    for (int di = -hw; di <= hw; ++di) {
      const double z = row[di];
      rs += z;
      ri = fma((double)di, z, ri);
      rq = fma(z, z, rq);
    }
    Sz += rs;
    Siz += ri;
    Sjz = fma((double)dj, rs, Sjz);
    Szz += rq;
  }

  const double inv_np = 1.0 / (double)a.np;
  const double cxx = g.res * g.res * ((double)a.sii * inv_np);
  const double nm1 = (double)a.np / (double)(a.np > 1 ? a.np - 1 : 1);
  bool need_fixup = false;

  float pf[NX];
  load_row(js + 1 + R, pf);

#pragma unroll 1
  for (int j = js; j < jend; ++j) {
    // ---- emit row j ---------------------------------------------------------------------------
    if (i < rg.i1) {
      const size_t o = mo + (size_t)j * g.rows + i;
      float stepv = 0.0f;
      if (a.combine) stepv = step[o];
      float nx = qnanf(), ny = qnanf(), nz = qnanf(), o_slope = qnanf(), o_rough = qnanf();
      bool done = false;
      if (j > dirty_until && !border_lane) {
        const double mz = Sz * inv_np;
        const double ca = -g.res * Siz * inv_np;  // cov(x,z), x = -res*di
        const double cb = -g.res * Sjz * inv_np;  // cov(y,z)
        const double cd = fma(Szz, inv_np, -mz * mz);
        const double delta = 0.5 * (cxx - cd);
        const double h2 = fma(ca, ca, cb * cb);
        const double s = sqrt_nr(fma(delta, delta, h2));
        const double t = delta >= 0.0 ? delta + s : h2 * rcp_nr(s - delta);
        if (t > 0.0 && t < 1e300) {
          if (cxx > 1e-8) {  // eigenvalues(1) == cxx here (NormalVectorsFilter's "> 1e-8" test)
            const double inv = rsqrt_nr(fma(t, t, h2));
            nx = (float)(-ca * inv);
            ny = (float)(-cb * inv);
            nz = (float)(t * inv);
          } else {
            nx = 0.0f;
            ny = 0.0f;
            nz = 1.0f;
          }
          const double sl = acos_poly((double)nz);  // SlopeFilter.cpp:74
          o_slope = sl < a.slope_crit ? (float)(1.0 - sl * a.inv_slope_crit) : 0.0f;
          // n^T C n with the float32 normal, C = [[c,0,a],[0,c,b],[a,b,d]]  (RoughnessFilter.cpp:105-117)
          const double x = (double)nx, y = (double)ny, z = (double)nz;
          double q = fma(cxx, fma(x, x, y * y), fma(2.0 * z, fma(ca, x, cb * y), cd * z * z));
          q = q > 0.0 ? q : 0.0;
          const double rgh = sqrt_nr(q * nm1);
          o_rough = rgh < a.rough_crit ? (float)(1.0 - rgh * a.inv_rough_crit) : 0.0f;
          done = true;
        }
      }
      need_fixup |= !done;
      slope[o] = o_slope;  // NaN == "to be recomputed by the fix-up pass if the centre is valid"
      rough[o] = o_rough;
      if (a.combine) {
        const float ta = a.w_slope * o_slope, tb = a.w_step * stepv, tc = a.w_rough * o_rough;
        const float tab = ta + tb;
        const float tabc = tab + tc;
        trav[o] = a.w_scale * tabc;
      }
      if (onx) {
        onx[o] = nx;
        ony[o] = ny;
        onz[o] = nz;
      }
    }
    if (j + 1 >= jend) break;
    // ---- bring in row j+1+R, start the load of the one after ------------------------------------
    store_row(j + 1 + R, pf);
    load_row(j + 2 + R, pf);
    // ---- slide the disc from row j to row j+1 ---------------------------------------------------
    double sj = 0.0;
#pragma unroll
    for (int di = -R; di <= R; ++di) {
      const int h = a.h[di < 0 ? -di : di];
      const double zl = ring[((j + 1 + h) & (NR - 1)) * W + c + di];
      const double zt = ring[((j - h) & (NR - 1)) * W + c + di];
      const double u = zl - zt, v = zl + zt;
      Sz += u;
      if (di != 0) Siz = fma((double)di, u, Siz);
      Szz = fma(u, v, Szz);
      sj += fma((double)h, v, zl);
    }
    Sjz = (Sjz + sj) - Sz;
  }
  if (__any(need_fixup) && lane == 0)
    block_flags[((size_t)(rg.map >= 0 ? 0 : blockIdx.z) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = 1;
}

template <int R>
void launch_r(const Geo& g, const SlideArgs& a, const Layers& L, bool keep, const Region& r, int* flags, FastGrid* fg,
              hipStream_t s) {
  fg->nbx = (r.i1 - r.i0 + kLanes - 1) / kLanes;
  fg->nby = (r.j1 - r.j0 + a.out_rows - 1) / a.out_rows;
  fg->nbz = r.map >= 0 ? 1 : g.batch;
  fg->out_rows = a.out_rows;
  dim3 grid((unsigned)fg->nbx, (unsigned)fg->nby, (unsigned)fg->nbz);
  (void)hipMemsetAsync(flags, 0, sizeof(int) * (size_t)fg->nbx * fg->nby * fg->nbz, s);
  hipLaunchKernelGGL(k_normals_slide<R>, grid, dim3(kLanes), 0, s, g, a, L.elev, L.step, L.slope, L.rough, L.trav,
                     keep ? L.nx : nullptr, keep ? L.ny : nullptr, keep ? L.nz : nullptr, flags, r);
}

}  // namespace

constexpr int kStripRows = 128;

// Normals + slope + roughness for a tie-free disc (same disc for normals and roughness, positive axis
// z, at least 3 cells).  Returns false if the shape is not supported by this kernel.
bool normals_fast(const Geo& g, const ChainParams& p, const Layers& L, bool keep_normals, bool combine,
                  const Region& r, int* flags, FastGrid* fg, hipStream_t s) {
  const Disc& d = p.normals;
  if (d.n_ties != 0 || d.R < 1 || d.R > 16 || d.npoints < 3) return false;
  SlideArgs a;
  int sii = 0;
  for (int k = 0; k <= kMaxRadiusCells; ++k) a.h[k] = k <= d.R ? d.hw[k] : -1;
  for (int dj = -d.R; dj <= d.R; ++dj) {
    const int hw = d.hw[dj < 0 ? -dj : dj];
    for (int di = -hw; di <= hw; ++di) sii += di * di;
  }
  a.np = d.npoints;
  a.sii = sii;
  a.out_rows = kStripRows;
  a.slope_crit = p.slope_crit;
  a.inv_slope_crit = 1.0 / p.slope_crit;
  a.rough_crit = p.rough_crit;
  a.inv_rough_crit = 1.0 / p.rough_crit;
  a.w_scale = p.w_scale;
  a.w_slope = p.w_slope;
  a.w_step = p.w_step;
  a.w_rough = p.w_rough;
  a.combine = combine ? 1 : 0;
  switch (d.R) {
#define X(q) \
  case q:    \
    launch_r<q>(g, a, L, keep_normals, r, flags, fg, s); \
    return true;
    X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)
#undef X
    default:
      return false;
  }
}

int normals_fast_max_blocks(const Geo& g) {
  return ((g.rows + kLanes - 1) / kLanes) * ((g.cols + kStripRows - 1) / kStripRows + 1) * g.batch;
}

}  // namespace fast
}  // namespace te
