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
#include "te_march.h"
#include "te_eig.h"

#include <cstdlib>

namespace te {
namespace fast {

namespace {

// waves per SIMD k_normals_slide is compiled for; launch_r sizes the interior strips to fill exactly these slots in one round
constexpr int kNormWaves = 2;

struct SlideArgs {
  int h[kMaxRadiusCells + 1];  // half-height of disc column |di| (== half-width of row |dj|)
  double hd[kMaxRadiusCells + 1];  // the same as doubles (kernel arguments stay in scalar registers)
  int np;                      // cells in the disc
  int sii;                     // sum of di^2 over the disc
  double slope_crit, inv_slope_crit, rough_crit, inv_rough_crit;
  float band_slope, band_rough;  // a raw score within this of the clip at 0 is left to the fix-up pass (kExactNaNBits, te_internal.h)
  float w_scale, w_slope, w_step, w_rough;
  int combine;
  // RoughnessFilter as a stand-alone plugin (RoughnessFilter.cpp:84-119): surface_normal_{x,y,z} are INPUT layers and the
  // only output is the roughness score of the plane through the disc's mean with THAT normal.  Discs that lie inside the
  // map and hold no invalid cell take the closed form  n^T C n  from the sliding moments; everything else (the frame,
  // holes, an invalid centre under a valid normal) is left NaN, flagged, and recomputed by the fix-up pass.
  int given;
  const int* gtab;  // [(2R+1)^2][6] x/y moments {n, si, sj, sii, sij, sjj} of the disc clipped by the map border
  int fi0, fj0, ntx, nty;  // fix-up flag grid: 64x16 tiles from (fi0, fj0), ntx x nty per map
  int fix_groups;          // workgroups of the fix-up pass: the flag of tile t lives at (t % groups) * kFixTiles + t / groups
  // One launch covers up to five rectangles (top, bottom, left, right frame with short strips and the
  // clipped-disc tail, then the interior): blocks [first, first+nbx*nby) belong to rectangle k.
  struct Sub {
    int i0, i1, j0, j1, out_rows, nbx, first, border;
  } sub[5];
  int nsub;
};

__device__ __forceinline__ float qnanf() { return __builtin_nanf(""); }

constexpr int ring_rows(int R) {
  int n = 4;
  while (n < 2 * R + 2) n *= 2;
  return n;
}

// One strip of one rectangle.  BORDER selects the variant that also handles discs clipped by the map
// border (slower, branchy).  In the interior variant the per-row work -- closed-form tail of row j,
// staging of row j+1+R, slide to row j+1 -- is ONE straight-line block (loads use clamped addresses,
// LDS writes are unconditional), so the latency-bound tail overlaps with the issue-bound slide; the
// predicated global stores of row j come last.
// Q >= 0: the disc is the compile-time shape Shape<Q> (R == Shape<Q>::R): run half-heights are literals and
// the ring rows come from a rotating table of row offsets (no per-column wrap bookkeeping).  Q < 0: run
// table from the arguments (any tie-free radius up to 16 cells).
template <int R, int Q, bool BORDER>
__device__ __forceinline__ void march(double* __restrict__ ring, const Geo& g, const SlideArgs& a, int k,
                                      const float* __restrict__ elev, const float* __restrict__ step,
                                      float* __restrict__ slope, float* __restrict__ rough, float* __restrict__ trav,
                                      float* __restrict__ onx, float* __restrict__ ony, float* __restrict__ onz,
                                      int* __restrict__ tile_flags, const Region& rg) {
  constexpr int W = kLanes + 2 * R;
  constexpr int NR = 2 * R + 3;  // rows j-R .. j+2+R: the ring reads of step j+1 are issued during step j
  constexpr int NX = (W + kLanes - 1) / kLanes;
  constexpr int kAhead = 4;      // rows (and step values) are fetched kAhead steps before they are needed
  const int lane = threadIdx.x;
  const int map = rg.map >= 0 ? rg.map : blockIdx.z;
  const size_t mo = (size_t)map * g.rows * g.cols;
  const int lb = (int)blockIdx.x - a.sub[k].first;
  const int sub_i1 = a.sub[k].i1, sub_j1 = a.sub[k].j1, out_rows = a.sub[k].out_rows;
  const int i0 = a.sub[k].i0 + (lb % a.sub[k].nbx) * kLanes;
  const int js = a.sub[k].j0 + (lb / a.sub[k].nbx) * out_rows;
  const int jend = (js + out_rows < sub_j1) ? js + out_rows : sub_j1;
  const int i = i0 + lane;
  const int ic = i < g.rows ? i : g.rows - 1;
  const float* __restrict__ em = elev + mo;
  // clip of my disc by the left/right map border (0: none; k>0: columns di < -R+k missing; k<0: di > R+k missing)
  // (lanes beyond the map edge compute garbage that is never stored: keep their table index in range)
  const int kx = (i >= g.rows) ? 0 : (i < R) ? (R - i) : ((g.rows - 1 - i < R) ? -(R - (g.rows - 1 - i)) : 0);
  const int c = lane + R;  // my column inside a staged row

  int dirty_until = js - R - 1;  // outputs j <= dirty_until may see an invalid cell
  double zref = 0.0;
  // global loads use clamped (always valid) addresses; whether the cell exists is decided when it is staged
  auto load_row = [&](int r, float (&pf)[NX]) {
    const int rc = r < 0 ? 0 : (r >= g.cols ? g.cols - 1 : r);
    const float* __restrict__ src = em + (size_t)rc * g.rows;
#pragma unroll
    for (int x = 0; x < NX; ++x) {
      int ci = i0 - R + lane + x * kLanes;
      ci = ci < 0 ? 0 : (ci >= g.rows ? g.rows - 1 : ci);
      pf[x] = src[ci];
    }
  };
  auto store_row = [&](int r, int slot, const float (&pf)[NX]) {
    const bool rin = r >= 0 && r < g.cols && r >= js - R;  // rows above the strip's first disc stay zero
    bool bad = false;
    double* dst = ring + slot * W;
    double v0 = 0.0;
#pragma unroll
    for (int x = 0; x < NX; ++x) {
      const int cc = lane + x * kLanes;
      const int ci = i0 - R + cc;
      const bool inmap = rin && cc < W && ci >= 0 && ci < g.rows;
      const float t = pf[x];
      const bool ok = inmap && __builtin_isfinite(t);
      bad |= inmap && !ok;
      const double v = ok ? (double)t - zref : 0.0;  // invalid / outside the map: contributes nothing to the z-sums
      if (x == 0) v0 = v;
      // lanes beyond the end of the row repeat their first write (same address, same value): no predication
      dst[cc < W ? cc : lane] = cc < W ? v : v0;
    }
    const int cand = __any(bad) ? r + R : dirty_until;
    dirty_until = cand > dirty_until ? cand : dirty_until;
  };

  // ---- prologue: empty ring, reference height -------------------------------------------------------
  // The march starts 2R+1 rows above the strip with an EMPTY disc (rows above js-R count as zeros and
  // are never staged), so the first real output needs no separate 253-point sum: 2R+1 ordinary steps.
  for (int idx = lane; idx < NR * W; idx += kLanes) ring[idx] = 0.0;
  {
    float pf0[NX];
    bool found = false;
    for (int r = (js - R < 0 ? 0 : js - R); r <= js + R && r < g.cols && !found; ++r) {  // uniform; normally one pass
      load_row(r, pf0);
#pragma unroll
      for (int x = 0; x < NX; ++x) {
        const unsigned long long msk = __ballot(__builtin_isfinite(pf0[x]) && pf0[x] != 0.0f);
        if (!found && msk) {
          zref = (double)__shfl(pf0[x], __ffsll((long long)msk) - 1);
          found = true;
        }
      }
    }
  }
  __syncthreads();
  double Sz = 0.0, Siz = 0.0, Sjz = 0.0, Szz = 0.0;
  constexpr int WU = ((2 * R + 1 + kAhead - 1) / kAhead) * kAhead;  // warm-up steps: whole groups of kAhead
  const int jstart = js - WU;
  int slot_j = 0;  // ring slot of row j (slot(r) = (r - jstart) mod NR)

  // Loop constants are pinned in VGPRs: as scalars they (with the 2(R+1) ring row offsets) overflow the
  // 102 SGPRs and every use in the row loop becomes a v_readlane from the spill register.
  double inv_np = 1.0 / (double)a.np;
  double cxx = g.res * g.res * ((double)a.sii * inv_np);
  double nm1 = (double)a.np / (double)(a.np > 1 ? a.np - 1 : 1);
  double nres = -g.res;
  asm volatile("" : "+v"(inv_np), "+v"(cxx), "+v"(nm1), "+v"(nres));
  constexpr bool kStatic = Q >= 0;
  double hdv[R + 1];
#pragma unroll
  for (int d = 0; d <= R; ++d) {
    hdv[d] = kStatic ? 0.0 : a.hd[d];
    if (!kStatic) asm volatile("" : "+v"(hdv[d]));
  }
  // the same for the output pointers (per-lane element 0 of this strip's map)
  // (typed as global-address-space pointers: through the asm the compiler would otherwise lose the address
  // space and emit flat stores, which also count as LDS operations for s_waitcnt)
  typedef float __attribute__((address_space(1))) gfloat;
  gfloat* v_slope = (gfloat*)(slope + mo);
  gfloat* v_rough = (gfloat*)(rough + mo);
  gfloat* v_trav = (gfloat*)(trav + mo);
  gfloat* v_nx = (gfloat*)(onx ? onx + mo : nullptr);
  gfloat* v_ny = (gfloat*)(onx ? ony + mo : nullptr);
  gfloat* v_nz = (gfloat*)(onx ? onz + mo : nullptr);
  asm volatile("" : "+v"(v_slope), "+v"(v_rough), "+v"(v_trav), "+v"(v_nx), "+v"(v_ny), "+v"(v_nz));
  const float slope_critf = (float)a.slope_crit, inv_slope_critf = (float)a.inv_slope_crit;
  const float rough_critf = (float)a.rough_crit, inv_rough_critf = (float)a.inv_rough_crit;
  // fix-up flags: one per 64x16 tile of the whole region the chain runs on (origin a.fi0, a.fj0)
  const int tile_col = (rg.map >= 0 ? 0 : (int)blockIdx.z) * a.ntx * a.nty + ((i0 - a.fi0) >> 6);  // + tile row * ntx

  // ring offsets (in doubles) of the leading / trailing row of disc column |di| = d, advanced every step
  int lead[R + 1], trail[R + 1];
  int rowoff[NR];  // kStatic: ring offset (in doubles) of row j-R+k of the current step
  if (kStatic) {
#pragma unroll
    for (int k = 0; k < NR; ++k) rowoff[k] = ((k - R + NR) % NR) * W;
  } else {
#pragma unroll
    for (int d = 0; d <= R; ++d) {
      const int h = a.h[d];
      lead[d] = ((1 + h) % NR) * W;
      trail[d] = ((NR - h) % NR) * W;
    }
  }
  float pfq[kAhead][NX];
  float stq[kAhead];
  auto load_step = [&](int jj) -> float {
    const int jc = jj < 0 ? 0 : (jj >= g.cols ? g.cols - 1 : jj);
    return step[mo + (size_t)jc * g.rows + ic];
  };
#pragma unroll
  for (int d = 0; d < kAhead; ++d) {
    load_row(jstart + 2 + R + d, pfq[d]);
    stq[d] = load_step(jstart + d);
  }
  // Software pipeline in two groups of disc columns, A = |di| <= RA and B = the rest: the ring reads of
  // group B (step j) are in flight while group A is summed, and those of group A (step j+1) while group B
  // is summed and the tail of row j+1 runs -- no more values are live than when a whole step was read at
  // once, but the LDS latency is off the critical path.
  constexpr int RA = (R + 1) / 3;
  struct Vals {
    double lp[R + 1], lm[R + 1], tp[R + 1], tm[R + 1];  // leading / trailing row, column +d / -d
  };
  // `ahead` = 1: the reads belong to the step after the one the row table describes
  auto fetch = [&](Vals& v, auto lo, auto hi, auto ahead) {
    if constexpr (kStatic) {
      constexpr int LO = decltype(lo)::value, HI = decltype(hi)::value;
      static_for<HI - LO + 1>([&](auto dc) __attribute__((always_inline)) {
        constexpr int d = LO + decltype(dc)::value;
        constexpr int h = Shape<Q>::hw(d);
        // columns +d and -d of the disc share their leading row (j+1+h) and trailing row (j-h)
        const double* rl = ring + rowoff[R + 1 + h + decltype(ahead)::value] + c;
        const double* rt = ring + rowoff[R - h + decltype(ahead)::value] + c;
        v.lp[d] = rl[d];
        v.tp[d] = rt[d];
        v.lm[d] = d ? rl[-d] : 0.0;
        v.tm[d] = d ? rt[-d] : 0.0;
      });
    } else {
#pragma unroll
      for (int d = decltype(lo)::value; d <= decltype(hi)::value; ++d) {
        const double* rl = ring + lead[d] + c;
        const double* rt = ring + trail[d] + c;
        v.lp[d] = rl[d];
        v.tp[d] = rt[d];
        v.lm[d] = d ? rl[-d] : 0.0;
        v.tm[d] = d ? rt[-d] : 0.0;
        lead[d] = lead[d] + W >= NR * W ? 0 : lead[d] + W;
        trail[d] = trail[d] + W >= NR * W ? 0 : trail[d] + W;
      }
    }
  };
  double sj = 0.0;
  auto consume = [&](const Vals& q, auto lo, auto hi) {
#pragma unroll
    for (int d = decltype(lo)::value; d <= decltype(hi)::value; ++d) {
      const double hh = kStatic ? (double)Shape<Q < 0 ? 0 : Q>::hw(d <= Shape<Q < 0 ? 0 : Q>::R ? d : 0) : hdv[d];
      {
        const double zl = q.lp[d], zt = q.tp[d];
        const double u = zl - zt, v = zl + zt;
        Sz += u;
        if (d != 0) Siz = fma((double)d, u, Siz);
        Szz = fma(u, v, Szz);
        sj += fma(hh, v, zl);
      }
      if (d != 0) {
        const double zl = q.lm[d], zt = q.tm[d];
        const double u = zl - zt, v = zl + zt;
        Sz += u;
        Siz = fma(-(double)d, u, Siz);
        Szz = fma(u, v, Szz);
        sj += fma(hh, v, zl);
      }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using IA = std::integral_constant<int, RA>;
  using IB = std::integral_constant<int, RA + 1>;
  using IR = std::integral_constant<int, R>;
  Vals vals;
  // bring in row j+2+R, fetch the row kAhead further down, slide the disc from row j to row j+1
  // (qs = j mod kAhead is a compile-time queue slot: the loops below are unrolled by kAhead)
  auto advance = [&](int j, float (&pf)[NX], float& st) {
    int sr = slot_j + 2 + R;
    sr = sr >= NR ? sr - NR : sr;
    store_row(j + 2 + R, sr, pf);
    load_row(j + 2 + R + kAhead, pf);
    st = load_step(j + kAhead);
    sj = 0.0;
    if (RA < R) fetch(vals, IB{}, IR{}, I0{});  // group B of step j
    consume(vals, I0{}, IA{});                  // group A of step j (read during step j-1)
    fetch(vals, I0{}, IA{}, I1{});              // group A of step j+1 (needs row j+2+R, staged above)
    if (RA < R) consume(vals, IB{}, IR{});
    Sjz = (Sjz + sj) - Sz;
    slot_j = slot_j + 1 >= NR ? 0 : slot_j + 1;
    if constexpr (kStatic) {  // the table moves on to step j+1: the oldest row's slot takes the next new row
      const int oldest = rowoff[0];
      static_for<NR - 1>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        rowoff[k] = rowoff[k + 1];
      });
      rowoff[NR - 1] = oldest;
    }
  };
  {  // row jstart+1+R (the first that can be real); everything above is a virtual zero row
    float pf0[NX];
    load_row(jstart + 1 + R, pf0);
    store_row(jstart + 1 + R, (1 + R) % NR, pf0);
    fetch(vals, I0{}, IA{}, I0{});
  }

#pragma unroll 1
  for (int j0 = jstart; j0 < js; j0 += kAhead) {  // warm-up: fill the disc
#pragma unroll
    for (int qs = 0; qs < kAhead; ++qs) advance(j0 + qs, pfq[qs], stq[qs]);
  }

#pragma unroll 1
  for (int j0 = js; j0 < jend; j0 += kAhead) {
#pragma unroll
  for (int qs = 0; qs < kAhead; ++qs) {
    const int j = j0 + qs;
    if (j >= jend) break;
    __builtin_amdgcn_sched_barrier(0);  // keep the kAhead unrolled rows apart (register pressure)
    // ---- tail of row j (values stay in registers until the stores below) --------------------------
    float nx = qnanf(), ny = qnanf(), nz = qnanf(), o_slope = qnanf(), o_rough = qnanf();
    bool done = false;
    // step score of this row, consumed HERE: if its use sank below advance() (which refills the queue slot) the
    // old value would still be live when the new load is issued, the load would get another register and the
    // copy back would force a full vmcnt(0) drain at the end of every row
    float tb = a.w_step * stq[qs];
    asm volatile("" : "+v"(tb));
    const int ky = (j < R) ? (R - j) : ((g.cols - 1 - j < R) ? -(R - (g.cols - 1 - j)) : 0);  // uniform
    if (a.given) {  // (uniform)
      const size_t og = (size_t)j * g.rows + ic;
      const float gxf = v_nx[og], gyf = v_ny[og], gzf = v_nz[og];
      const double gx = (double)gxf, gy = (double)gyf, gz = (double)gzf;
      const double mz = Sz * inv_np;
      const double ca = nres * Siz * inv_np;  // cov(x,z), x = -res*di
      const double cb = nres * Sjz * inv_np;  // cov(y,z)
      const double cd = fma(Szz, inv_np, -mz * mz);
      // sum of squared distances to the plane / N = n^T C n with C = [[cxx,0,ca],[0,cxx,cb],[ca,cb,cd]] (RoughnessFilter.cpp:105-117)
      double q = fma(cxx, fma(gx, gx, gy * gy), fma(2.0 * gz, fma(gx, ca, gy * cb), gz * gz * cd));
      q = q > 0.0 ? q : 0.0;
      const float rgh = __builtin_amdgcn_sqrtf((float)(q * nm1));
      const float rr = 1.0f - rgh * inv_rough_critf;
      const float rs = rgh < rough_critf ? rr : 0.0f;
      const bool have = __builtin_isfinite(gxf);  // :84 (the reference tests surface_normal_x only)
      const bool near = near_clip(rr, a.band_rough);  // a score at its clip: the fix-up pass decides zero / not zero
      const bool clean = (j > dirty_until) && (kx == 0) && (ky == 0) && !near;
      done = clean || !have;  // no normal: the layer stays NaN, nothing to recompute
      o_rough = (clean && have) ? rs : (near ? exact_nanf() : qnanf());
    } else if (BORDER && j > dirty_until && (kx != 0 || ky != 0)) {
      // disc clipped by the map border: the z-sums are already right (cells outside contribute 0),
      // the x/y moments of the clipped disc come from the host-built table
      double qrough = 0.0;
      const int* gt = a.gtab + ((ky + R) * (2 * R + 1) + (kx + R)) * 6;
      done = border_tail(g.res, gt[0], gt[1], gt[2], gt[3], gt[4], gt[5], Sz, Siz, Sjz, Szz, nx, ny, nz, qrough);
      if (done) {
        const double sl = acos_poly((double)nz);
        const float rs = (float)(1.0 - sl * a.inv_slope_crit);
        o_slope = sl < a.slope_crit ? rs : 0.0f;
        const int n = gt[0];
        const double rgh = n > 1 ? sqrt_nr(qrough * ((double)n / (double)(n - 1))) : 1e300;
        const float rr = (float)(1.0 - rgh * a.inv_rough_crit);
        o_rough = rgh < a.rough_crit ? rr : 0.0f;
        if (near_clip(rs, a.band_slope) || (n > 1 && near_clip(rr, a.band_rough))) {  // a score at its clip: the fix-up pass decides
          done = false;
          nx = ny = nz = o_rough = qnanf();
          o_slope = exact_nanf();
        }
      }
    } else {
      // straight-line closed-form tail; cells it cannot finish (dirty rows, degenerate t) are masked to NaN
      const double mz = Sz * inv_np;
      const double ca = nres * Siz * inv_np;  // cov(x,z), x = -res*di
      const double cb = nres * Sjz * inv_np;  // cov(y,z)
      const double cd = fma(Szz, inv_np, -mz * mz);
      const double delta = 0.5 * (cxx - cd);
      const double h2 = fma(ca, ca, cb * cb);
      const double s = sqrt_nr(fma(delta, delta, h2));
      // t = delta + s loses relative accuracy only when delta < 0 and h << |delta| (z-variance above the
      // x/y variance and almost no tilt: normal nearly horizontal); those cells go to the general path
      const double t = delta + s;
      done = (j > dirty_until) && (t > 0.0) && (t < 1e300) && (delta >= 0.0 || t > 1e-6 * s) && (kx == 0) &&
             (ky == 0);  // clipped discs belong to the BORDER launch
      const bool eig_ok = cxx > 1e-8;  // eigenvalues(1) == cxx here (NormalVectorsFilter's "> 1e-8" test)
      const double inv = rsqrt_nr(fma(t, t, h2));
      const float fz = eig_ok ? (float)(t * inv) : 1.0f;
      const float fx = eig_ok ? (float)(-ca * inv) : 0.0f;
      const float fy = eig_ok ? (float)(-cb * inv) : 0.0f;
      // slope = acos(float32 nz) (SlopeFilter.cpp:74); float32 evaluation, |error| < 3e-7 rad
      const float sl = acosf_poly(fz);
      const float rsl = 1.0f - sl * inv_slope_critf;
      const float ss = sl < slope_critf ? rsl : 0.0f;
      // roughness^2 * (N-1)/N = n^T C n = smallest eigenvalue (RoughnessFilter.cpp:105-117); with the
      // float32-rounded normal the quadratic form differs from lambda0 by O(c * 1e-15)
      double q = eig_ok ? 0.5 * (cxx + cd) - s : cd;
      q = q > 0.0 ? q : 0.0;
      const float rgh = __builtin_amdgcn_sqrtf((float)(q * nm1));
      const float rrg = 1.0f - rgh * inv_rough_critf;
      const float rs = rgh < rough_critf ? rrg : 0.0f;
      const bool near = near_clip(rsl, a.band_slope) || near_clip(rrg, a.band_rough);  // a score at its clip: the fix-up pass decides
      done = done && !near;
      nx = done ? fx : qnanf();
      ny = done ? fy : qnanf();
      nz = done ? fz : qnanf();
      o_slope = done ? ss : (near ? exact_nanf() : qnanf());
      o_rough = done ? rs : qnanf();
    }
    const float ta = a.w_slope * o_slope, tc = a.w_rough * o_rough;
    const float tab = ta + tb;
    const float tabc = tab + tc;
    const float o_trav = a.w_scale * tabc;

    // (the last one of a strip is not needed, but keeps the body branch-free)
    advance(j, pfq[qs], stq[qs]);

    // ---- stores of row j ------------------------------------------------------------------------------
    const bool emit = i < sub_i1;
    if (emit) {
      const size_t o = (size_t)j * g.rows + i;
      if (!a.given) v_slope[o] = o_slope;  // NaN == "to be recomputed by the fix-up pass if the centre is valid"
      v_rough[o] = o_rough;
      if (a.combine) v_trav[o] = o_trav;
      if (onx && !a.given) {
        v_nx[o] = nx;
        v_ny[o] = ny;
        v_nz[o] = nz;
      }
    }
    if (__any(emit && !done) && lane == 0) {
      const int t = tile_col + ((j - a.fj0) >> 4) * a.ntx;
      tile_flags[(t % a.fix_groups) * kFixTiles + t / a.fix_groups] = 1;
    }
  }
  }
}

template <int R, int Q>
__global__ __launch_bounds__(kLanes, kNormWaves) void k_normals_slide(Geo g, SlideArgs a, const float* __restrict__ elev,
                                                          const float* __restrict__ step, float* __restrict__ slope,
                                                          float* __restrict__ rough, float* __restrict__ trav,
                                                          float* __restrict__ onx, float* __restrict__ ony,
                                                          float* __restrict__ onz, int* __restrict__ tile_flags,
                                                          Region rg) {
  __shared__ double ring[(2 * R + 3) * (kLanes + 2 * R)];
  int k = 0;  // which rectangle this block works on (uniform)
#pragma unroll
  for (int t = 1; t < 5; ++t)
    if (t < a.nsub && (int)blockIdx.x >= a.sub[t].first) k = t;
  if (a.sub[k].border)
    march<R, Q, true>(ring, g, a, k, elev, step, slope, rough, trav, onx, ony, onz, tile_flags, rg);
  else
    march<R, Q, false>(ring, g, a, k, elev, step, slope, rough, trav, onx, ony, onz, tile_flags, rg);
}

constexpr int kStripRows = 128;       // interior strips
constexpr int kBorderStripRows = 32;  // the clipped-disc tail is slower: shorter strips finish with the rest

// Split the region into the frame (clipped discs; launched first, short strips) and the cells whose
// disc lies inside the map; every rectangle keeps its 64-column blocks aligned to r.i0.
template <int R, int Q>
void launch_r(const Geo& g, SlideArgs a, const Layers& L, bool keep, const Region& r, int* flags, hipStream_t s) {
  const int ja = r.j0 > R ? r.j0 : (R < r.j1 ? R : r.j1);                      // first interior row
  const int jb = r.j1 < g.cols - R ? r.j1 : (g.cols - R > ja ? g.cols - R : ja);  // one past the last
  int bxa = 0;
  const int nbx = (r.i1 - r.i0 + kLanes - 1) / kLanes;
  while (bxa < nbx && r.i0 + kLanes * bxa < R) ++bxa;
  int bxb = nbx;
  while (bxb > bxa && r.i0 + kLanes * bxb - 1 > g.rows - 1 - R) --bxb;  // last lane of block bxb-1
  const int il = r.i0 + kLanes * bxa < r.i1 ? r.i0 + kLanes * bxa : r.i1;
  const int ir = r.i0 + kLanes * bxb < r.i1 ? r.i0 + kLanes * bxb : r.i1;
  const int rect[5][5] = {{r.i0, il, ja, jb, 1},        // left
                          {ir, r.i1, ja, jb, 1},        // right
                          {r.i0, r.i1, r.j0, ja, 1},    // top
                          {r.i0, r.i1, jb, r.j1, 1},    // bottom
                          {il, ir, ja, jb, 0}};         // interior
  // interior strip height: as many strips as fit in ONE round of resident waves (3 per SIMD at <=168
  // VGPRs and 13 KB of LDS), so that no second, mostly idle, round is needed
  int border_blocks = 0;
  for (int t = 0; t < 4; ++t)
    if (rect[t][1] > rect[t][0] && rect[t][3] > rect[t][2])
      border_blocks += ((rect[t][1] - rect[t][0] + kLanes - 1) / kLanes) *
                       ((rect[t][3] - rect[t][2] + kBorderStripRows - 1) / kBorderStripRows);
  int interior_rows = kStripRows;
  if (ir > il && jb > ja) {
    const int maps = r.map >= 0 ? 1 : g.batch;
    const int capacity = kNormWaves * 4 * 256 / (maps > 0 ? maps : 1) - border_blocks;
    const int nbx_in = (ir - il + kLanes - 1) / kLanes;
    int strips = capacity / nbx_in;
    strips = strips < 1 ? 1 : strips;
    interior_rows = (jb - ja + strips - 1) / strips;
    interior_rows = interior_rows < 48 ? 48 : (interior_rows > 512 ? 512 : interior_rows);
  }
  int n = 0, first = 0;
  for (int t = 0; t < 5; ++t) {
    if (rect[t][1] <= rect[t][0] || rect[t][3] <= rect[t][2]) continue;
    SlideArgs::Sub& u = a.sub[n++];
    u.i0 = rect[t][0]; u.i1 = rect[t][1]; u.j0 = rect[t][2]; u.j1 = rect[t][3];
    u.border = rect[t][4];
    u.out_rows = u.border ? kBorderStripRows : interior_rows;
    u.nbx = (u.i1 - u.i0 + kLanes - 1) / kLanes;
    u.first = first;
    first += u.nbx * ((u.j1 - u.j0 + u.out_rows - 1) / u.out_rows);
  }
  a.nsub = n;
  for (int t = n; t < 5; ++t) a.sub[t] = a.sub[n ? n - 1 : 0];
  if (first == 0) return;
  hipLaunchKernelGGL((k_normals_slide<R, Q>), dim3((unsigned)first, 1, (unsigned)(r.map >= 0 ? 1 : g.batch)), dim3(kLanes),
                     0, s, g, a, L.elev, L.step, L.slope, L.rough, L.trav, keep ? L.nx : nullptr,
                     keep ? L.ny : nullptr, keep ? L.nz : nullptr, flags, r);
}

}  // namespace

// Normals + slope + roughness for a tie-free disc (same disc for normals and roughness, positive axis
// z, at least 3 cells).  Returns false if the shape is not supported by this kernel.
bool normals_fast(const Geo& g, const ChainParams& p, const Layers& L, bool keep_normals, bool combine,
                  const Region& r, int* flags, const int* gtab, FastGrid* fg, hipStream_t s, bool* combined) {
  const Disc& d = p.normals;
  if (normals_small(g, p, L, keep_normals, r, flags, fg, s)) {  // discs of at most 13 cells: one cell per thread
    *combined = false;
    return true;
  }
  if (d.n_ties != 0) {  // a tie radius: k_normals3's TIES march or nothing
    static const bool no_n3_ties = lab_flag("TE_NO_N3");
    if (no_n3_ties || !gtab || g.rows < 2 * d.reach + 1 || g.cols < 2 * d.reach + 1) return false;
    fg->ntx = (r.i1 - r.i0 + kLanes - 1) / kLanes;
    fg->nty = (r.j1 - r.j0 + 15) / 16;
    fg->nbz = r.map >= 0 ? 1 : g.batch;
    fg->frame = 0;
    if (!normals_fast3(g, p, L, keep_normals, r, flags, fg, s)) return false;
    *combined = false;
    return true;
  }
  if (d.R < 1 || d.R > 16 || d.npoints < 3) return false;
  SlideArgs a;
  int sii = 0;
  for (int k = 0; k <= kMaxRadiusCells; ++k) {
    a.h[k] = k <= d.R ? d.hw[k] : -1;
    a.hd[k] = (double)a.h[k];
  }
  for (int dj = -d.R; dj <= d.R; ++dj) {
    const int hw = d.hw[dj < 0 ? -dj : dj];
    for (int di = -hw; di <= hw; ++di) sii += di * di;
  }
  a.np = d.npoints;
  a.sii = sii;
  fg->ntx = (r.i1 - r.i0 + kLanes - 1) / kLanes;
  fg->nty = (r.j1 - r.j0 + 15) / 16;
  fg->nbz = r.map >= 0 ? 1 : g.batch;
  a.fi0 = r.i0;
  a.fj0 = r.j0;
  a.ntx = fg->ntx;
  a.nty = fg->nty;
  a.fix_groups = fix_groups(fg->ntx * fg->nty * fg->nbz);
  a.slope_crit = p.slope_crit;
  a.inv_slope_crit = 1.0 / p.slope_crit;
  a.rough_crit = p.rough_crit;
  a.inv_rough_crit = 1.0 / p.rough_crit;
  a.band_slope = clip_band_slope(p.slope_crit);
  a.band_rough = clip_band_rough(p.rough_crit);
  a.w_scale = p.w_scale;
  a.w_slope = p.w_slope;
  a.w_step = p.w_step;
  a.w_rough = p.w_rough;
  a.combine = combine ? 1 : 0;
  a.given = 0;
  a.gtab = gtab;
  if (g.rows < 2 * d.R + 1 || g.cols < 2 * d.R + 1 || !gtab) return false;  // both borders inside one disc
  // cells whose disc lies inside the map: k_normals3 (3 waves per SIMD); this file keeps the frame
  static const bool no_n3 = lab_flag("TE_NO_N3");
  fg->frame = 0;
  if (!no_n3 && normals_fast3(g, p, L, keep_normals, r, flags, fg, s)) {
    *combined = false;  // k_normals3 does not combine: the caller runs k_combine (or the footprint mask kernel does)
    return true;        // the frame and whatever that kernel flagged belong to the fix-up pass
  }
  *combined = combine;
  if (d.Q >= 1) {  // instantiated shape: compile-time run table
    switch (d.Q) {
#define X(q)                                                              \
  case q:                                                                 \
    if constexpr (q >= 1) {                                               \
      launch_r<Shape<q>::R, q>(g, a, L, keep_normals, r, flags, s);       \
      return true;                                                        \
    }                                                                     \
    break;
      TE_DISC_SHAPES(X)
#undef X
      default:
        break;
    }
  }
  switch (d.R) {
#define X(q) \
  case q:    \
    launch_r<q, -1>(g, a, L, keep_normals, r, flags, s); \
    return true;
    X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)
#undef X
    default:
      return false;
  }
}

// RoughnessFilter alone on the roughness disc with the normals of the layers (SlideArgs::given).  False: not taken.
bool roughness_given_fast(const Geo& g, const ChainParams& p, const Layers& L, const Region& r, int* flags, FastGrid* fg, hipStream_t s) {
  const Disc& d = p.rough;
  if (d.n_ties != 0 || d.R < 1 || d.R > 16 || d.npoints < 3 || !flags) return false;
  if (g.rows < 2 * d.R + 1 || g.cols < 2 * d.R + 1) return false;
  SlideArgs a;
  int sii = 0;
  for (int k = 0; k <= kMaxRadiusCells; ++k) {
    a.h[k] = k <= d.R ? d.hw[k] : -1;
    a.hd[k] = (double)a.h[k];
  }
  for (int dj = -d.R; dj <= d.R; ++dj) {
    const int hw = d.hw[dj < 0 ? -dj : dj];
    for (int di = -hw; di <= hw; ++di) sii += di * di;
  }
  a.np = d.npoints;
  a.sii = sii;
  fg->ntx = (r.i1 - r.i0 + kLanes - 1) / kLanes;
  fg->nty = (r.j1 - r.j0 + 15) / 16;
  fg->nbz = r.map >= 0 ? 1 : g.batch;
  fg->frame = 0;
  a.fi0 = r.i0;
  a.fj0 = r.j0;
  a.ntx = fg->ntx;
  a.nty = fg->nty;
  a.fix_groups = fix_groups(fg->ntx * fg->nty * fg->nbz);
  a.slope_crit = p.slope_crit;
  a.inv_slope_crit = 1.0 / p.slope_crit;
  a.rough_crit = p.rough_crit;
  a.inv_rough_crit = 1.0 / p.rough_crit;
  a.band_slope = clip_band_slope(p.slope_crit);
  a.band_rough = clip_band_rough(p.rough_crit);
  a.w_scale = a.w_slope = a.w_step = a.w_rough = 0.0f;
  a.combine = 0;
  a.given = 1;
  a.gtab = nullptr;
  if (d.Q >= 1) {  // instantiated shape: compile-time run table
    switch (d.Q) {
#define X(q)                                                      \
  case q:                                                         \
    if constexpr (q >= 1) {                                       \
      launch_r<Shape<q>::R, q>(g, a, L, true, r, flags, s);       \
      return true;                                                \
    }                                                             \
    break;
      TE_DISC_SHAPES(X)
#undef X
      default:
        break;
    }
  }
  switch (d.R) {
#define X(q) \
  case q:    \
    launch_r<q, -1>(g, a, L, true, r, flags, s); \
    return true;
    X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)
#undef X
    default:
      return false;
  }
}

// Host side: x/y moments of the disc clipped to di in [lo_i, hi_i], dj in [lo_j, hi_j] for every clip
// code (kx, ky) in [-R, R]^2 (see k_normals_slide).  out must hold (2R+1)^2 * 6 ints.
void build_clip_table(const Disc& d, int R, int* out) {
  for (int ky = -R; ky <= R; ++ky)
    for (int kx = -R; kx <= R; ++kx) {
      const int lo_i = kx > 0 ? -R + kx : -R, hi_i = kx < 0 ? R + kx : R;
      const int lo_j = ky > 0 ? -R + ky : -R, hi_j = ky < 0 ? R + ky : R;
      int n = 0, si = 0, sj = 0, sii = 0, sij = 0, sjj = 0;
      for (int dj = lo_j; dj <= hi_j; ++dj) {
        const int adj = dj < 0 ? -dj : dj;
        const int hw = adj <= d.R ? d.hw[adj] : -1;
        for (int di = -hw; di <= hw; ++di) {
          if (di < lo_i || di > hi_i) continue;
          ++n;
          si += di;
          sj += dj;
          sii += di * di;
          sij += di * dj;
          sjj += dj * dj;
        }
      }
      int* e = out + ((ky + R) * (2 * R + 1) + (kx + R)) * 6;
      e[0] = n; e[1] = si; e[2] = sj; e[3] = sii; e[4] = sij; e[5] = sjj;
    }
}

int normals_fast_max_blocks(const Geo& g) {
  const int ntiles = ((g.rows + kLanes - 1) / kLanes) * ((g.cols + 15) / 16) * g.batch;  // one flag per 64x16 tile ...
  // ... at (t % groups) * kFixTiles + t / groups; a region run has the tile count of its region, so the largest need of
  // any count up to ntiles
  const int small = ntiles < 2048 ? ntiles : 2048;
  const int a = fix_groups(ntiles) * kFixTiles, b = fix_groups(small) * kFixTiles;
  return a > b ? a : b;
}

}  // namespace fast
}  // namespace te
