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
#include "te_cell.h"
#include "te_eig.h"
#include "te_internal.h"

#include <cstdlib>

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
__device__ __forceinline__ void accumulate_disc(Mom& m, const Geo& g, const Disc& d, const float* ctr, int tw, int i,
                                                int j, double z0) {
  // One run (row of the disc) at a time: dj is constant along it, so only the moments in di are gathered per
  // cell (3 integer + 3 double updates) and the dj factors are applied once per run.
  for (int dj = -d.R; dj <= d.R; ++dj) {
    const int hw = d.hw[dj < 0 ? -dj : dj];
    const float* row = ctr + dj * tw;
    int rn = 0, rsi = 0, rsii = 0;
    double rsz = 0.0, rsiz = 0.0, rszz = 0.0;
#pragma unroll 4
    for (int di = -hw; di <= hw; ++di) {
      const float z = row[di];
      const bool v = (z == z);
      const double dz = v ? (double)z - z0 : 0.0;
      const int w = v ? 1 : 0;
      const int wdi = v ? di : 0;
      rn += w;
      rsi += wdi;
      rsii += wdi * di;
      rsz += dz;
      rsiz = fma((double)di, dz, rsiz);
      rszz = fma(dz, dz, rszz);
    }
    m.n += rn;
    m.si += rsi;
    m.sj += dj * rn;
    m.sii += rsii;
    m.sij += dj * rsi;
    m.sjj += dj * dj * rn;
    m.sz += rsz;
    m.siz += rsiz;
    m.sjz = fma((double)dj, rsz, m.sjz);
    m.szz += rszz;
  }
  for (int t = 0; t < d.n_ties; ++t) {
    const int di = d.tie_di[t], dj = d.tie_dj[t];
    if (tie_inside(g, d.r2, i, j, di, dj)) mom_add(m, di, dj, ctr[dj * tw + di], z0);
  }
}

struct NormalsArgs {
  Disc dn, dr;
  int same_disc, axis;
  double slope_crit, rough_crit;
  float w_scale, w_slope, w_step, w_rough;
  int combine;  // also write the traversability layer (reads the step layer)
  int given_normals;  // RoughnessFilter alone: surface_normal_{x,y,z} are INPUT layers (RoughnessFilter.cpp:108-110)
  int rank_rule;      // TE_OPT_NORMALS_RANK_RULE (te_cell.h: rank_deficient)
  float band_slope, band_rough;  // k_normals_fixup: a fast-tail score within this of its clip takes the generic arithmetic (te_internal.h)
};

// One cell from the LDS tile: normals -> slope -> roughness (-> combine), all outputs written.
__device__ __forceinline__ void normals_cell(const Geo& g, const NormalsArgs& a, const float* ctr, int tw, int i, int j,
                                             size_t o, const float* __restrict__ step, float* __restrict__ slope,
                                             float* __restrict__ rough, float* __restrict__ trav,
                                             float* __restrict__ onx, float* __restrict__ ony,
                                             float* __restrict__ onz) {
  const float z0f = *ctr;
  const float qnan = __builtin_nanf("");
  float o_slope = qnan, o_rough = qnan, nf[3] = {qnan, qnan, qnan};
  if (a.given_normals) {  // RoughnessFilter::update as a stand-alone plugin: the normals come from the map
    nf[0] = onx[o];
    nf[1] = ony[o];
    nf[2] = onz[o];
    if (__builtin_isfinite(nf[0])) {  // RoughnessFilter.cpp:84 (hard-coded surface_normal_x validity)
      Mom m;
      double cov[6];
      mom_zero(m);
      // the reference gathers the valid elevations of the window even if the centre itself is invalid
      accumulate_disc(m, g, a.dr, ctr, tw, i, j, (z0f == z0f) ? (double)z0f : 0.0);
      if (m.n >= 1) {
        covariance(m, g.res, cov);
        o_rough = roughness_score(m, cov, nf, a.rough_crit);
      } else {
        // 0 points: nPoints is a size_t, so sum / (nPoints - 1) = 0 / SIZE_MAX = 0 (RoughnessFilter.cpp:117):
        // roughness 0 -> score 1 when the critical value is positive
        o_rough = a.rough_crit > 0.0 ? 1.0f : 0.0f;
      }
    }
    rough[o] = o_rough;
    return;
  }
  if (z0f == z0f) {  // normals only where the input layer is valid; slope/roughness follow (SlopeFilter.cpp:71, RoughnessFilter.cpp:84)
    const double z0 = (double)z0f;
    Mom m;
    double cov[6];
    mom_zero(m);
    accumulate_disc(m, g, a.dn, ctr, tw, i, j, z0);
    covariance(m, g.res, cov);
    normal_from_cov(m, cov, a.axis, nf);
    if (a.rank_rule && m.n >= 3 && rank_deficient(cov)) {  // (uniform flag) UnitZ, towards the positive axis
      nf[0] = nf[1] = 0.0f;
      nf[2] = 1.0f;  // (UnitZ . axis is 1 or 0: never negative, no flip)
    }
    o_slope = slope_score(nf[2], a.slope_crit);
    if (!a.same_disc) {
      mom_zero(m);
      accumulate_disc(m, g, a.dr, ctr, tw, i, j, z0);
      covariance(m, g.res, cov);
    }
    o_rough = roughness_score(m, cov, nf, a.rough_crit);
  }
  slope[o] = o_slope;
  rough[o] = o_rough;
  if (a.combine) {
    const float ta = a.w_slope * o_slope, tb = a.w_step * step[o], tc = a.w_rough * o_rough;
    const float tab = ta + tb;
    const float tabc = tab + tc;
    trav[o] = a.w_scale * tabc;
  }
  if (onx) {
    onx[o] = nf[0];
    ony[o] = nf[1];
    onz[o] = nf[2];
  }
}

__global__ __launch_bounds__(TX* BY) void k_normals(Geo g, NormalsArgs a, const float* __restrict__ elev,
                                                    const float* __restrict__ step, float* __restrict__ slope,
                                                    float* __restrict__ rough, float* __restrict__ trav,
                                                    float* __restrict__ onx, float* __restrict__ ony,
                                                    float* __restrict__ onz, Region rg) {
  extern __shared__ float tile[];
  const int map = rg.map >= 0 ? rg.map : blockIdx.z;
  const size_t mo = (size_t)map * g.rows * g.cols;
  const int i0 = rg.i0 + blockIdx.x * TX, j0 = rg.j0 + blockIdx.y * TY;
  const int K = a.dn.reach > a.dr.reach ? a.dn.reach : a.dr.reach;
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
    normals_cell(g, a, tile + (lj + K) * tw + (threadIdx.x + K), tw, i, j, mo + (size_t)j * g.rows + i, step, slope,
                 rough, trav, onx, ony, onz);
  }
}

// Fix-up pass behind the shape-specialised kernel (te_fast_normals.hip): that kernel leaves NaN in the
// slope layer of every valid cell it could not finish (a neighbour invalid or outside the map, or a
// degenerate covariance) and raises the flag of its block.  Here a 64x16 tile whose block(s) are
// flagged collects those cells into a dense list and recomputes them with the general path, so that
// a thin frame of border cells costs only its own cells.
__global__ __launch_bounds__(TX* BY) void k_normals_fixup(Geo g, NormalsArgs a, const float* __restrict__ elev,
                                                          const float* __restrict__ step, float* __restrict__ slope,
                                                          float* __restrict__ rough, float* __restrict__ trav,
                                                          float* __restrict__ onx, float* __restrict__ ony,
                                                          float* __restrict__ onz, int* __restrict__ flags,
                                                          FastGrid fg, Region rg) {
  extern __shared__ float tile[];
  __shared__ unsigned short todo[TX * TY];
  __shared__ int ntodo;
  __shared__ unsigned long long pending;
  // A workgroup looks at kFixTiles tiles of the region (one flag per lane) and works through the flagged
  // ones; it clears the flags it consumes, so they are all zero again when the kernel ends and the slide
  // kernel needs no memset.  A clean map costs one flag load per kFixTiles tiles.  The tiles of a workgroup
  // are gridDim.x apart (flagged tiles come in runs -- along a hole boundary, the map frame -- and a run must
  // spread over many workgroups instead of queueing up in one), their flags are adjacent (te_internal.h).
  const int ntiles = fg.ntx * fg.nty * fg.nbz;
  const int tid0 = threadIdx.y * TX + threadIdx.x;
  if (tid0 < 64) {
    const int t = blockIdx.x + tid0 * (int)gridDim.x;
    const int slot = blockIdx.x * kFixTiles + tid0;
    bool f = tid0 < kFixTiles && t < ntiles && flags[slot] != 0;
    if (f) flags[slot] = 0;
    if (fg.frame > 0 && tid0 < kFixTiles && t < ntiles) {  // tiles that reach into the frame are always processed
      const int ftx = t % fg.ntx, fty = (t / fg.ntx) % fg.nty;
      const int fi0 = rg.i0 + ftx * TX, fj0 = rg.j0 + fty * TY;
      const int fi1 = fi0 + TX < rg.i1 ? fi0 + TX : rg.i1, fj1 = fj0 + TY < rg.j1 ? fj0 + TY : rg.j1;
      f = f || fi0 < fg.frame || fj0 < fg.frame || fi1 > g.rows - fg.frame || fj1 > g.cols - fg.frame;
    }
    const unsigned long long m = __ballot(f);
    if (tid0 == 0) pending = m;
  }
  __syncthreads();
  unsigned long long todo_tiles = pending;
  while (todo_tiles) {  // uniform
  const int bit = __ffsll((long long)todo_tiles) - 1;
  todo_tiles &= todo_tiles - 1;
  const int t = blockIdx.x + bit * (int)gridDim.x;
  const int tx = t % fg.ntx, ty = (t / fg.ntx) % fg.nty, mapz = t / (fg.ntx * fg.nty);
  const int map = rg.map >= 0 ? rg.map : mapz;
  const size_t mo = (size_t)map * g.rows * g.cols;
  const int i0 = rg.i0 + tx * TX;
  const int jb0 = rg.j0 + ty * TY;
  const int jb1 = jb0 + TY < rg.j1 ? jb0 + TY : rg.j1;
  const int K = a.dn.reach > a.dr.reach ? a.dn.reach : a.dr.reach;
  const int tw = TX + 2 * K;
  const int tid = threadIdx.y * TX + threadIdx.x;
  for (int j0 = jb0; j0 < jb1; j0 += TY) {
    __syncthreads();
    if (tid == 0) ntodo = 0;
    load_tile(tile, elev + mo, g, i0, j0, K);
    __syncthreads();
    for (int c = 0; c < CPT; ++c) {
      const int lj = threadIdx.y + c * BY;
      const int i = i0 + threadIdx.x, j = j0 + lj;
      bool need = false, exact = false;
      if (i < rg.i1 && j < jb1) {
        const float z0 = tile[(lj + K) * tw + (threadIdx.x + K)];
        const size_t oc = mo + (size_t)j * g.rows + i;
        // (given normals: the sliding kernel left NaN in the ROUGHNESS layer of the cells it could not finish; a cell takes
        // part if its normal is valid, whatever its own elevation is -- RoughnessFilter.cpp:84)
        const float s = a.given_normals ? rough[oc] : slope[oc];
        const bool have = a.given_normals ? __builtin_isfinite(onx[oc]) : (z0 == z0);
        const bool in_frame = fg.frame > 0 && (i < fg.frame || j < fg.frame || i >= g.rows - fg.frame || j >= g.cols - fg.frame);
        need = have && (!(s == s) || in_frame);
        exact = __builtin_bit_cast(unsigned, s) == kExactNaNBits;  // a fast tail's score sat at its clip: the generic arithmetic decides
        if (!a.given_normals && in_frame && !(z0 == z0)) {  // nobody else writes the frame: an invalid centre has no normal, slope or roughness
          const size_t o = mo + (size_t)j * g.rows + i;
          const float qn = __builtin_nanf("");
          slope[o] = qn;
          rough[o] = qn;
          if (a.combine) trav[o] = qn;
          if (onx) {
            onx[o] = qn;
            ony[o] = qn;
            onz[o] = qn;
          }
        }
      }
      // order-preserving compaction (one LDS atomic per wave): neighbouring lanes keep neighbouring cells, so
      // the tile reads of the gather stay (nearly) bank-conflict free
      const unsigned long long mask = __ballot(need);
      int base = 0;
      if (threadIdx.x == 0 && mask) base = atomicAdd(&ntodo, __popcll(mask));
      base = __shfl(base, 0);
      if (need) todo[base + __popcll(mask & ((1ull << threadIdx.x) - 1ull))] = (unsigned short)((exact ? 0x8000 : 0) | (lj * TX + threadIdx.x));
    }
    __syncthreads();
    const int n = ntodo;
    for (int k = tid; k < n; k += TX * BY) {
      const int c = todo[k] & 0x7fff;
      const bool exact = (todo[k] & 0x8000) != 0;
      const int lj = c / TX, li = c - lj * TX;
      const int i = i0 + li, j = j0 + lj;
      const float* ctr = tile + (lj + K) * tw + (li + K);
      const size_t o = mo + (size_t)j * g.rows + i;
      // the cells listed here have a valid centre; same general tail as the clipped discs of the slide kernel
      // (moments -> one Jacobi rotation + secular equation), the cyclic Jacobi of normals_cell only for the
      // configurations it does not resolve
      bool fast_done = false;
      if (a.same_disc && a.axis == 2 && !a.given_normals && !exact) {
        Mom m;
        mom_zero(m);
        accumulate_disc(m, g, a.dn, ctr, tw, i, j, (double)*ctr);
        float nx, ny, nz;
        double q = 0.0;
        fast_done = fast::border_tail(g.res, m.n, m.si, m.sj, m.sii, m.sij, m.sjj, m.sz, m.siz, m.sjz, m.szz, nx, ny, nz, q);
        if (fast_done) {
          const double sl = fast::acos_poly((double)nz);
          const float rs = (float)(1.0 - sl / a.slope_crit);
          const float o_slope = sl < a.slope_crit ? rs : 0.0f;
          const double rgh = m.n > 1 ? fast::sqrt_nr(q * ((double)m.n / (double)(m.n - 1))) : 1e300;
          const float rr = (float)(1.0 - rgh / a.rough_crit);
          const float o_rough = rgh < a.rough_crit ? rr : 0.0f;
          // this tail's normal can differ from the generic one by a float32 ulp: a score at its clip goes the generic way too
          fast_done = !(fast::near_clip(rs, a.band_slope) || (m.n > 1 && fast::near_clip(rr, a.band_rough)));
          if (fast_done) {
            slope[o] = o_slope;
            rough[o] = o_rough;
            if (a.combine) {
              const float ta = a.w_slope * o_slope, tb = a.w_step * step[o], tc = a.w_rough * o_rough;
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
        }
      }
      if (!fast_done) normals_cell(g, a, ctr, tw, i, j, o, step, slope, rough, trav, onx, ony, onz);
    }
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

// SlopeFilter::update as a stand-alone plugin: surface_normal_z is an INPUT layer (SlopeFilter.cpp:67-84)
__global__ void k_slope_from_nz(Geo g, double crit, const float* __restrict__ nz, float* __restrict__ slope) {
  const size_t n = (size_t)g.rows * g.cols * g.batch;
  const size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n) return;
  const float v = nz[o];
  slope[o] = __builtin_isfinite(v) ? slope_score(v, crit) : __builtin_nanf("");
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

// One reference plugin at a time (the drop-in SlopeFilter / StepFilter / RoughnessFilter adapters).
hipError_t launch_chain(const Geo& g, const ChainParams& p, const Layers& L, const Region& r, unsigned flags, hipStream_t stream);

hipError_t launch_filter(const Geo& g, const ChainParams& p, const Layers& L, int filter, unsigned flags,
                         hipStream_t stream) {
  const Region r = {-1, 0, 0, g.rows, g.cols};
  const dim3 blk(TX, BY);
  const bool use_fast = (flags & TE_RUN_GENERIC_KERNELS) == 0;
  if (filter == TE_FILTER_SLOPE) {
    const size_t n = (size_t)g.rows * g.cols * g.batch;
    hipLaunchKernelGGL(k_slope_from_nz, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, g, p.slope_crit, L.nz,
                       L.slope);
  } else if (filter == TE_FILTER_STEP) {
    if (!(use_fast && (fast::step_height_fast(p.step1.Q, g, L.elev, L.step_height, r, stream) ||
                       fast::step_height_ties(p.step1, g, L.elev, L.step_height, L.tie_scratch, r, stream))))
      hipLaunchKernelGGL(k_step_height, tile_grid(g, r), blk, tile_bytes(p.step1.reach), stream, g, p.step1, L.elev,
                         L.step_height, r);
    if (!(use_fast && (fast::step_score_fast(p.step2.Q, g, p.step_crit, p.step_ncrit, L.step_height, L.step, r, stream) ||
                       fast::step_score_ties(p.step2, g, p.step_crit, p.step_ncrit, L.step_height, L.step, L.tie_scratch, r, stream))))
      hipLaunchKernelGGL(k_step_score, tile_grid(g, r), blk, tile_bytes(p.step2.reach), stream, g, p.step2, p.step_crit,
                         p.step_ncrit, L.step_height, L.step, r);
  } else if (filter == TE_FILTER_ROUGHNESS) {
    NormalsArgs na;
    na.dn = p.rough;
    na.dr = p.rough;
    na.same_disc = 1;
    na.axis = p.axis;
    na.rank_rule = p.rank_rule;
    na.slope_crit = p.slope_crit;
    na.rough_crit = p.rough_crit;
    na.w_scale = na.w_slope = na.w_step = na.w_rough = 0.0f;
    na.combine = 0;
    na.given_normals = 1;
    na.band_slope = clip_band_slope(p.slope_crit);
    na.band_rough = clip_band_rough(p.rough_crit);
    // tie-free discs: the sliding kernel with the layers' normals (interior, hole-free discs in closed form from the
    // moments), the fix-up pass for the frame and the holes; otherwise the generic kernel on every cell
    FastGrid fg;
    if (use_fast && fast::roughness_given_fast(g, p, L, r, L.block_flags, &fg, stream))
      hipLaunchKernelGGL(k_normals_fixup, dim3((unsigned)fix_groups(fg.ntx * fg.nty * fg.nbz)), blk, tile_bytes(p.rough.reach), stream, g, na,
                         L.elev, L.step, L.slope, L.rough, L.trav, L.nx, L.ny, L.nz, L.block_flags, fg, r);
    else
      hipLaunchKernelGGL(k_normals, tile_grid(g, r), blk, tile_bytes(p.rough.reach), stream, g, na, L.elev, L.step, L.slope,
                         L.rough, L.trav, L.nx, L.ny, L.nz, r);
  } else if (filter == TE_FILTER_NORMALS) {
    // the normals pass alone, normals kept: the sliding kernel computes slope and roughness on the way (same disc), the
    // plugins that want those layers compute them from their own inputs later
    ChainParams q = p;
    q.rough = q.normals;
    q.same_rough_disc = 1;
    return launch_chain(g, q, L, r, (flags & TE_RUN_GENERIC_KERNELS) | TE_RUN_KEEP_NORMALS | TE_RUN_NORMALS_ONLY, stream);
  } else if (filter == TE_FILTER_COMBINE) {
    const dim3 cgrid((unsigned)((g.rows + 255) / 256), (unsigned)g.cols, (unsigned)g.batch);
    hipLaunchKernelGGL(k_combine, cgrid, dim3(256), 0, stream, g, p.w_scale, p.w_slope, p.w_step, p.w_rough, L.slope,
                       L.step, L.rough, L.trav, r);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
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
  const bool use_fast = (flags & TE_RUN_GENERIC_KERNELS) == 0;
  // The step filter and the normals/slope/roughness kernel are independent until the combine: on whole
  // maps they run concurrently on two HIP streams (each alone leaves the VALUs half idle), joined by an
  // event before the combine.
  const bool normals_only = (flags & TE_RUN_NORMALS_ONLY) != 0;  // measurement aid: the dominant kernel alone
  // SINGLE-CELL STEP WINDOWS and a normals disc k_normals_small takes (the reference's defaults on a map coarser than
  // 0.04 m): the step score needs no neighbour -- 1 where the elevation is valid, NaN elsewhere (te_normals_small.hip) --, so
  // that kernel writes the step layer and the weighted sum as well: the whole chain is ONE kernel
  // (4096^2, default parameters at res 0.05: 0.34 -> 0.24 ms; 256^2: 17 -> 5 us).  Whole-map runs only: a region run
  // re-filters dilated regions stage by stage as before.
  if (whole && use_fast && !normals_only && p.same_rough_disc && p.axis == 2) {  // a small launch: the whole chain in one kernel
    TraceRange tr("chain: one kernel (k_chain_window)");
    if (fast::chain_window(g, p, L, (flags & TE_RUN_KEEP_NORMALS) != 0, r, !(flags & kDeferCombine), stream)) return hipGetLastError();
  }
  if (whole && use_fast && !normals_only && p.same_rough_disc && p.axis == 2 && p.step1.n_ties == 0 && p.step1.Q == 0 && p.step2.n_ties == 0 &&
      p.step2.Q == 0) {
    const bool comb = !(flags & kDeferCombine);
    FastGrid fg;
    TraceRange tr("chain: normals + slope + roughness + single-cell step windows + combine (one kernel)");
    if (fast::normals_small(g, p, L, (flags & TE_RUN_KEEP_NORMALS) != 0, r, L.block_flags, &fg, stream, /*write_step*/ true, comb)) {
      return hipGetLastError();
    }
  }
  const bool overlap = whole && L.aux_stream != nullptr && !normals_only;
  hipStream_t ss = overlap ? L.aux_stream : stream;
  if (overlap) {
    (void)hipEventRecord(L.ev_fork, stream);
    (void)hipStreamWaitEvent(ss, L.ev_fork, 0);
  }
  if (!normals_only) {
    TraceRange tr("chain: step filter (height, score)");
    if (!(use_fast && (fast::step_height_fast(p.step1.Q, g, L.elev, L.step_height, r1, ss) ||
                       fast::step_height_ties(p.step1, g, L.elev, L.step_height, L.tie_scratch, r1, ss))))
      hipLaunchKernelGGL(k_step_height, tile_grid(g, r1), blk, tile_bytes(p.step1.reach), ss, g, p.step1, L.elev,
                         L.step_height, r1);
    if (!(use_fast && (fast::step_score_fast(p.step2.Q, g, p.step_crit, p.step_ncrit, L.step_height, L.step, r2, ss) ||
                       fast::step_score_ties(p.step2, g, p.step_crit, p.step_ncrit, L.step_height, L.step, L.tie_scratch, r2, ss))))
      hipLaunchKernelGGL(k_step_score, tile_grid(g, r2), blk, tile_bytes(p.step2.reach), ss, g, p.step2, p.step_crit,
                         p.step_ncrit, L.step_height, L.step, r2);
  }
  if (overlap) (void)hipEventRecord(L.ev_join, ss);
  const bool keep = (flags & TE_RUN_KEEP_NORMALS) != 0;
  const int Kn = p.normals.reach > p.rough.reach ? p.normals.reach : p.rough.reach;
  NormalsArgs na;
  na.dn = p.normals;
  na.dr = p.rough;
  na.same_disc = p.same_rough_disc;
  na.axis = p.axis;
  na.rank_rule = p.rank_rule;
  na.slope_crit = p.slope_crit;
  na.rough_crit = p.rough_crit;
  na.w_scale = p.w_scale;
  na.w_slope = p.w_slope;
  na.w_step = p.w_step;
  na.w_rough = p.w_rough;
  na.given_normals = 0;
  na.band_slope = clip_band_slope(p.slope_crit);
  na.band_rough = clip_band_rough(p.rough_crit);
  // the combine is fused into the normals kernel only when the step layer is complete before it starts;
  // region runs: the step reach may exceed the normals reach, combine separately
  na.combine = (whole && !overlap && !normals_only && !(flags & kDeferCombine)) ? 1 : 0;
  float* const knx = keep ? L.nx : nullptr;
  float* const kny = keep ? L.ny : nullptr;
  float* const knz = keep ? L.nz : nullptr;
  const bool fused_combine = whole && !overlap && !normals_only && !(flags & kDeferCombine);
  FastGrid fg;
  bool combined = false;
  TraceRange tr_normals("chain: normals + slope + roughness (+ fix-up)");
  if (use_fast && p.same_rough_disc && p.axis == 2 &&
      fast::normals_fast(g, p, L, keep, fused_combine, rn, L.block_flags, L.clip_table, &fg, stream, &combined)) {
    na.combine = combined ? 1 : 0;
    if (fg.frame >= 0)  // (k_normals_small settles every cell itself)
      hipLaunchKernelGGL(k_normals_fixup, dim3((unsigned)fix_groups(fg.ntx * fg.nty * fg.nbz)), blk, tile_bytes(Kn), stream, g, na, L.elev, L.step, L.slope,
                       L.rough, L.trav, knx, kny, knz, L.block_flags, fg, rn);
  } else {
    combined = na.combine != 0;
    hipLaunchKernelGGL(k_normals, tile_grid(g, rn), blk, tile_bytes(Kn), stream, g, na, L.elev, L.step, L.slope,
                       L.rough, L.trav, knx, kny, knz, rn);
  }
  if (overlap) (void)hipStreamWaitEvent(stream, L.ev_join, 0);
  // with the footprint pass right behind, its mask kernel (which reads the three scores anyway) combines
  if (!combined && !normals_only && !(flags & kDeferCombine)) {
    const dim3 cgrid((unsigned)((rc.i1 - rc.i0 + 255) / 256), (unsigned)(rc.j1 - rc.j0),
                     (unsigned)(rc.map >= 0 ? 1 : g.batch));
    hipLaunchKernelGGL(k_combine, cgrid, dim3(256), 0, stream, g, p.w_scale, p.w_slope, p.w_step, p.w_rough, L.slope,
                       L.step, L.rough, L.trav, rc);
  }
  return hipGetLastError();
}

}  // namespace te
