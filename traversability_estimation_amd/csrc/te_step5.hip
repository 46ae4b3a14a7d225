// te_step5.hip -- the shape-specialised StepFilter kernels on the te_march5.h framework.
//
//   k_step_height5<Q>  StepFilter::update first pass   traversability_estimation_filters/src/StepFilter.cpp:112-144
//   k_step_score5<Q>   StepFilter::update second pass  StepFilter.cpp:147-178
//
// Pure compare / select arithmetic on float32, so the results are bit-identical to the reference.  Invalid cells are
// staged as quiet NaN and v_max / v_min ignore them, which is exactly the reference's isValid() skip; cells outside the
// map are staged as NaN too (CircleIterator clamps at the border).
// First pass: the output exists only where the CENTRE is valid (:113).  The accumulators know nothing of the centre, so
// the row that holds it folds a poison into the maximum: +inf if the centre is invalid (it sticks: staged values are
// finite or NaN), NaN -- ignored -- otherwise; the emit turns an infinite difference into NaN with the same v_fma that
// canonicalises the inputs.  (round 3 kept the centre values of the 2R+1 pending rows in registers: 19 VGPRs.)
// RAW = true (tie radii, te_fast_step.hip): the running maximum / minimum, or maximum / count, are stored instead of
// the finished value, and the fold kernels k_step_height_ties / k_step_score_ties finish.
#include "te_march5.h"

namespace te {
namespace fast {

namespace {

constexpr int kHeight5Waves = 4, kScore5Waves = 4;  // waves per SIMD the kernels are compiled for
constexpr int kStep5Queue = 2;                       // passes in the prefetch queue (3 and 4 measured the same: profiles/r04_experiments.json)

struct Step5Args {
  const float* in;   // elevation / step_height
  float* out;        // step_height / traversability_step (RAW: the running maximum)
  float* out2;       // RAW: the running minimum / the count
  int rows, cols;
  long long map_cells;
  Region rg;
  int strip_rows;
  // second pass
  double crit, rcrit;
  float crit_lo;
  int ncrit;
};

// what both passes share: block geometry, the prefetch queue of the main rows and the halo load, the running offsets
template <int Q, int C_>
struct Step5Base {
  static constexpr int R = Shape<Q>::R, W = kLanes + 2 * R, C = C_;
  M5Lane<R> L;
  brsrc rs_in, rs_out, rs_out2;   // window column 0 of map row js - R (input) / js - 2R (outputs: the row the first pass's first row completes)
  unsigned row_bytes;             // one map row
  int cols, js, nout, r0;         // r0 = js - R: the strip's first input row (the descriptors' row 0)
  float pm0[C], pm1[C], ph[C];    // the queue: main cells of both rows, my halo cell

  __device__ __forceinline__ void init(const Step5Args& a, int lane, int i0, int js_, int jend, size_t mo) {
    L.init(lane, i0, a.rows);
    cols = a.cols;
    js = js_;
    nout = jend - js_;
    r0 = js_ - R;
    row_bytes = (unsigned)a.rows * 4u;
    // (descriptor bases may lie before their layer -- the first strip, the first block column: never dereferenced there)
    rs_in = make_rsrc(a.in + mo + ((long long)(js_ - R) * a.rows + (i0 - R)));
    rs_out = make_rsrc(a.out + mo + ((long long)(js_ - 2 * R) * a.rows + (i0 - R)));
    rs_out2 = make_rsrc(a.out2 ? a.out2 + mo + ((long long)(js_ - 2 * R) * a.rows + (i0 - R)) : a.out);
  }
  // Unconditional: rows above / below the map are read like any other -- every layer lives in the context's slab, which
  // keeps kSlabGuardRows rows of slack before its first and behind its last layer (te_internal.h), so the addresses are
  // valid memory, and stage_pair stages such rows as NaN whatever was read.  (A conditional load, besides its branch,
  // has the compiler fold stage_pair's canonicalisation into the load's block -- and wait for the load right there.)
  template <int q>
  __device__ __forceinline__ void load_pair(int r, ic<q>) {
    const unsigned so = (unsigned)(r - r0) * row_bytes;  // (uniform: one s_mul per pass)
    pm0[q] = bload_f(rs_in, L.o_main0, so);
    pm1[q] = bload_f(rs_in, L.o_main1, so);
    if constexpr (R > 0) ph[q] = bload_f(rs_in, L.o_halo, so);
  }
  // byte offset of output row j from the output descriptors' row 0 (map row js - 2R; a pass's second output row is the
  // first one's o_main1)
  __device__ __forceinline__ unsigned out_off(int j) const { return (unsigned)(j - (r0 - R)) * row_bytes; }
  // which rows of the pass starting at map row r lie inside the map (uniform)
  __device__ __forceinline__ bool pair_inside(int r) const { return (unsigned)r < (unsigned)(cols - 1); }  // 0 <= r and r + 1 < cols
  __device__ __forceinline__ bool row_inside(int r) const { return (unsigned)r < (unsigned)cols; }
  template <int n>
  __device__ __forceinline__ void rotate_queue(ic<n>) {  // slot s <- slot (s + n) % C
    if constexpr (n % C != 0) {
      float t0[C], t1[C], th[C];
      static_for<C>([&](auto sc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        t0[s] = pm0[(s + n) % C];
        t1[s] = pm1[(s + n) % C];
        th[s] = ph[(s + n) % C];
      });
      static_for<C>([&](auto sc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        pm0[s] = t0[s];
        pm1[s] = t1[s];
        ph[s] = th[s];
      });
    }
  }
  __device__ __forceinline__ bool emits(int j) const { return (unsigned)(j - js) < (unsigned)nout; }  // (uniform)
};

// ---- first pass: maximum - minimum of the valid elevations -------------------------------------------------------------
template <int Q, bool RAW, int C_>
struct HeightK : Step5Base<Q, C_> {
  using B = Step5Base<Q, C_>;
  using B::L;
  static constexpr int R = B::R, W = B::W, C = B::C;
  struct Acc {
    float mx, mn;
  };
  typedef Acc Run;
  float* lds;    // [2][W]
  float c0, c1;  // my staged cells of the pass
  float p0, p1;  // +inf where they are invalid, NaN otherwise (the poison the centre row folds in)

  template <int q>
  __device__ __forceinline__ void stage_pair(int r, ic<q>) {
    c0 = canon_nan(this->pm0[q]);
    c1 = canon_nan(this->pm1[q]);
    float vh = qnan();
    if constexpr (R > 0) vh = L.halo_in ? canon_nan(this->ph[q]) : qnan();
    if (__builtin_expect(!this->pair_inside(r), 0)) {  // a row outside the map: nothing there (CircleIterator clamps at the border)
      const bool in0 = this->row_inside(r), in1 = this->row_inside(r + 1);
      c0 = in0 ? c0 : qnan();
      c1 = in1 ? c1 : qnan();
      vh = (L.hrow ? in1 : in0) ? vh : qnan();
    }
    lds[R + L.lane] = c0;
    lds[W + R + L.lane] = c1;
    if constexpr (R > 0) lds[L.hlds] = vh;
    if constexpr (!RAW) {
      p0 = c0 == c0 ? qnan() : __builtin_inff();
      p1 = c1 == c1 ? qnan() : __builtin_inff();
    }
  }
  template <int slot>
  __device__ __forceinline__ void build(ic<slot>, Run (&s)[R + 1]) {
    const float* row = lds + slot * W + R + L.lane;
    s[0].mx = s[0].mn = slot ? c1 : c0;
    static_for<R>([&](auto dc) __attribute__((always_inline)) {
      constexpr int d = decltype(dc)::value + 1;
      const float a = row[-d], b = row[d];
      vmax3_min3(s[d].mx, s[d].mn, s[d - 1].mx, s[d - 1].mn, a, b);
    });
  }
  __device__ __forceinline__ void reset(Acc& a) { a.mx = a.mn = qnan(); }
  template <int E, int ROW>
  __device__ __forceinline__ void start(Acc& a, const Run& s) {
    a = s;
    if constexpr (!RAW && E == 0) a.mx = vmax2(a.mx, ROW ? p1 : p0);
  }
  template <int E>
  __device__ __forceinline__ void fold1(Acc& a, const Run& s) {
    a.mx = vmax2(a.mx, s.mx);
    a.mn = vmin2(a.mn, s.mn);
    if constexpr (!RAW && E == 0) a.mx = vmax2(a.mx, p0);
  }
  template <int E>
  __device__ __forceinline__ void fold2(Acc& a, const Run& s1, const Run& s2) {
    vmax3_min3(a.mx, a.mn, a.mx, a.mn, s1.mx, s2.mx, s1.mn, s2.mn);
    if constexpr (!RAW && E == 0) a.mx = vmax2(a.mx, p0);
    if constexpr (!RAW && E - 1 == 0) a.mx = vmax2(a.mx, p1);
  }
  template <int second>
  __device__ __forceinline__ void emit(ic<second>, int j, const Acc& a) {
    if (this->emits(j)) {
      const unsigned o = L.o_main0, so = this->out_off(j);
      if constexpr (RAW) {  // the fold over the circle cells comes first (k_step_height_ties)
        bstore_f(this->rs_out, o, so, a.mx);
        bstore_f(this->rs_out2, o, so, a.mn);
      } else {
        // StepFilter.cpp:143 (float)((double)max - (double)min) == max - min in float32: the double difference of two
        // floats rounded to float is the correctly rounded float difference (53 >= 2 * 24 + 2 bits).  :113 an invalid
        // centre: the maximum is +inf, the difference +inf (or NaN), and d * 0 + d makes it NaN; a finite d stays d.
        bstore_f(this->rs_out, o, so, canon_nan(__fsub_rn(a.mx, a.mn)));
      }
    }
  }
};

template <int Q, bool RAW, int C_>
__global__ __launch_bounds__(kLanes) __attribute__((amdgpu_waves_per_eu(kHeight5Waves, kHeight5Waves))) void k_step_height5(Step5Args a) {
  constexpr int R = Shape<Q>::R, W = kLanes + 2 * R;
  __shared__ float lds[2 * W];
  const int lane = threadIdx.x;
  const Region& rg = a.rg;
  const size_t mo = (size_t)(rg.map >= 0 ? rg.map : (int)blockIdx.z) * (size_t)a.map_cells;
  // (the last block of a row of blocks is shifted left to end at the region's edge: no lane is ever masked; the columns
  // it shares with its neighbour are written twice with the same bits)
  const int i0 = rg.i0 + (int)blockIdx.x * kLanes + kLanes > rg.i1 ? rg.i1 - kLanes : rg.i0 + (int)blockIdx.x * kLanes;
  const int js = rg.j0 + (int)blockIdx.y * a.strip_rows;
  if (js >= rg.j1) return;
  const int jend = js + a.strip_rows < rg.j1 ? js + a.strip_rows : rg.j1;
  HeightK<Q, RAW, C_> k;
  k.init(a, lane, i0, js, jend, mo);
  k.lds = lds;
  march5<Q>(k, js, jend);
}

// ---- second pass: maximum of the valid step heights and how many exceed the critical value ----------------------------
template <int Q, bool RAW, int C_>
struct ScoreK : Step5Base<Q, C_> {
  using B = Step5Base<Q, C_>;
  using B::L;
  static constexpr int R = B::R, W = B::W, C = B::C;
  struct Acc {
    float mx;
    int cn;
  };
  typedef Acc Run;
  float2* lds;  // [2][W] {step_height, (step_height > crit) as integer bits}
  Acc c0, c1;   // my staged cells of the pass
  float crit_lo;
  // emit
  double crit, rcrit;
  int ncrit;
  float one_if_crit;
  const double* ratio;  // LDS: nCells / nCellCritical_ for every possible count

  __device__ __forceinline__ Acc cell(float v) const {
    // (step heights are finite or NaN: the first pass canonicalises)
    Acc c;
    c.mx = v;
    c.cn = v > crit_lo ? 1 : 0;  // crit_lo = largest float <= critical_value: (double)s > crit  <=>  s > crit_lo
    return c;
  }
  template <int q>
  __device__ __forceinline__ void stage_pair(int r, ic<q>) {
    float v0 = this->pm0[q], v1 = this->pm1[q], vh = qnan();
    if constexpr (R > 0) vh = L.halo_in ? this->ph[q] : qnan();
    if (__builtin_expect(!this->pair_inside(r), 0)) {  // a row outside the map: nothing there
      const bool in0 = this->row_inside(r), in1 = this->row_inside(r + 1);
      v0 = in0 ? v0 : qnan();
      v1 = in1 ? v1 : qnan();
      vh = (L.hrow ? in1 : in0) ? vh : qnan();
    }
    c0 = cell(v0);
    c1 = cell(v1);
    lds[R + L.lane] = make_float2(c0.mx, __int_as_float(c0.cn));
    lds[W + R + L.lane] = make_float2(c1.mx, __int_as_float(c1.cn));
    if constexpr (R > 0) {
      const Acc h = cell(vh);
      lds[L.hlds] = make_float2(h.mx, __int_as_float(h.cn));
    }
  }
  template <int slot>
  __device__ __forceinline__ void build(ic<slot>, Run (&s)[R + 1]) {
    const float2* row = lds + slot * W + R + L.lane;
    s[0] = slot ? c1 : c0;
    static_for<R>([&](auto dc) __attribute__((always_inline)) {
      constexpr int d = decltype(dc)::value + 1;
      const float2 a = row[-d], b = row[d];
      vmax3_add3(s[d].mx, s[d].cn, s[d - 1].mx, a.x, b.x, s[d - 1].cn, __float_as_int(a.y), __float_as_int(b.y));
    });
  }
  __device__ __forceinline__ void reset(Acc& a) {
    a.mx = qnan();
    a.cn = 0;
  }
  template <int E, int ROW>
  __device__ __forceinline__ void start(Acc& a, const Run& s) {
    a = s;
  }
  template <int E>
  __device__ __forceinline__ void fold1(Acc& a, const Run& s) {
    a.mx = vmax2(a.mx, s.mx);
    a.cn += s.cn;
  }
  template <int E>
  __device__ __forceinline__ void fold2(Acc& a, const Run& s1, const Run& s2) {
    vmax3_add3(a.mx, a.cn, a.mx, s1.mx, s2.mx, a.cn, s1.cn, s2.cn);
  }
  template <int second>
  __device__ __forceinline__ void emit(ic<second>, int j, const Acc& a) {
    if (this->emits(j)) {
      const unsigned o = L.o_main0, so = this->out_off(j);
      if constexpr (RAW) {  // the fold over the circle cells comes first (k_step_score_ties)
        bstore_f(this->rs_out, o, so, a.mx);
        bstore_f(this->rs_out2, o, so, __int_as_float(a.cn));
        return;
      }
      // isValid: at least one valid step_height in the window (StepFilter.cpp:161), else the cell stays NaN.
      // nCells == 0: step = min(stepMax, 0 * stepMax) = 0 (:169-170) -> 1 - 0 / crit = 1 (0 if crit == 0: "0 < 0" fails);
      // nCells >= nCellCritical: the ratio is >= 1, so step = stepMax, and a counted cell means stepMax > crit -> 0.
      // Only 0 < nCells < nCellCritical needs the arithmetic, and a wavefront rarely holds such a cell.
      float o_ = a.cn == 0 ? one_if_crit : 0.0f;
      if (__builtin_expect(__any(a.cn > 0 && a.cn < ncrit), 0)) {
        const double sm = (double)vmax2_zero(a.mx);  // stepMax starts at 0.0 (:149)
        const double a1 = ratio[a.cn] * sm;           // nCells / nCellCritical_ * stepMax (:169)
        const double step = sm < a1 ? sm : a1;        // :170
        // step / crit without the division sequence: q0 = step * RN(1/crit), then two residual corrections
        // (Markstein: the first makes q faithful, the second correctly rounded), all branch-free
        const double q0 = step * rcrit;
        const double q1 = fma(fma(-q0, crit, step), rcrit, q0);
        const double q = fma(fma(-q1, crit, step), rcrit, q1);
        o_ = step < crit ? (float)(1.0 - q) : 0.0f;
      }
      o_ = (a.mx == a.mx) ? o_ : qnan();
      bstore_f(this->rs_out, o, so, o_);
    }
  }
};

template <int Q, bool RAW, int C_>
__global__ __launch_bounds__(kLanes) __attribute__((amdgpu_waves_per_eu(kScore5Waves, kScore5Waves))) void k_step_score5(Step5Args a) {
  using S = Shape<Q>;
  constexpr int R = S::R, W = kLanes + 2 * R;
  __shared__ float2 lds[2 * W];
  // nCells / nCellCritical_ for every possible count, divided once per block (exactly the reference's double division)
  __shared__ double ratio[S::npoints() + 1];
  const int lane = threadIdx.x;
  for (int c = lane; c <= S::npoints(); c += kLanes) ratio[c] = (double)c / (double)a.ncrit;
  const Region& rg = a.rg;
  const size_t mo = (size_t)(rg.map >= 0 ? rg.map : (int)blockIdx.z) * (size_t)a.map_cells;
  const int i0 = rg.i0 + (int)blockIdx.x * kLanes + kLanes > rg.i1 ? rg.i1 - kLanes : rg.i0 + (int)blockIdx.x * kLanes;  // see k_step_height5
  const int js = rg.j0 + (int)blockIdx.y * a.strip_rows;
  if (js >= rg.j1) return;
  const int jend = js + a.strip_rows < rg.j1 ? js + a.strip_rows : rg.j1;
  ScoreK<Q, RAW, C_> k;
  k.init(a, lane, i0, js, jend, mo);
  k.lds = lds;
  k.crit_lo = a.crit_lo;
  k.crit = a.crit;
  k.rcrit = a.rcrit;
  k.ncrit = a.ncrit;
  k.one_if_crit = 0.0 < a.crit ? 1.0f : 0.0f;
  k.ratio = ratio;
  __syncthreads();  // (one wave: orders the table writes before the first emit's reads)
  march5<Q>(k, js, jend);
}

long wave_slots5(int waves) {
  static const int ov = lab_int("TE_STEP_WAVES", 0);  // measurement aid: strips sized for this many waves per SIMD
  return 4L * device_cus() * (ov > 0 ? ov : waves);
}

// the shapes a whole-cell radius of 3 .. 10 cells leaves without its circle (2 cells: k_step_small) (te_march.h has them all): only these exist as RAW kernels
constexpr bool tie_free_part5(int Q) { return Q == 8 || Q == 13 || Q == 20 || Q == 34 || Q == 45 || Q == 61 || Q == 80 || Q == 98; }

template <int Q>
bool launch_step5(bool score, const Geo& g, Step5Args a, const Region& r, hipStream_t s) {
  const unsigned nx = (unsigned)((r.i1 - r.i0 + kLanes - 1) / kLanes), nz = (unsigned)(r.map >= 0 ? 1 : g.batch);
  a.rows = g.rows;
  a.cols = g.cols;
  a.map_cells = (long long)g.rows * g.cols;
  a.rg = r;
  a.strip_rows = plan_strip_rows(r.j1 - r.j0, (long)nx * nz, wave_slots5(score ? kScore5Waves : kHeight5Waves));
  const dim3 grid(nx, (unsigned)((r.j1 - r.j0 + a.strip_rows - 1) / a.strip_rows), nz);
  constexpr int C = kStep5Queue;
  if (a.out2) {
    if constexpr (tie_free_part5(Q)) {
      if (score)
        hipLaunchKernelGGL((k_step_score5<Q, true, C>), grid, dim3(kLanes), 0, s, a);
      else
        hipLaunchKernelGGL((k_step_height5<Q, true, C>), grid, dim3(kLanes), 0, s, a);
      return true;
    }
    return false;
  }
  if (score)
    hipLaunchKernelGGL((k_step_score5<Q, false, C>), grid, dim3(kLanes), 0, s, a);
  else
    hipLaunchKernelGGL((k_step_height5<Q, false, C>), grid, dim3(kLanes), 0, s, a);
  return true;
}

bool dispatch_step5(int Q, bool score, const Geo& g, const Step5Args& a, const Region& r, hipStream_t s) {
  if (r.i1 - r.i0 < kLanes || r.j1 <= r.j0) return false;  // the blocks are 64 cells wide and never mask a lane (the last one is shifted)
  switch (Q) {
#define X(q) \
  case q:    \
    return launch_step5<q>(score, g, a, r, s);
    TE_DISC_SHAPES(X)
#undef X
    default:
      return false;
  }
}

}  // namespace

// sh_min != nullptr: RAW (the running maximum goes to sh, the minimum to sh_min); false: shape / region not taken
bool step_height5(int Q, const Geo& g, const float* elev, float* sh, float* sh_min, const Region& r, hipStream_t s) {
  if (!layer_has_guard_rows(elev, g, sizeof(float))) return false;  // (the march loads rows beyond the layer: te_internal.h)
  Step5Args a = {};
  a.in = elev;
  a.out = sh;
  a.out2 = sh_min;
  return dispatch_step5(Q, false, g, a, r, s);
}

// out_count != nullptr: RAW (maximum to out, count to out_count)
bool step_score5(int Q, const Geo& g, double crit, int ncrit, const float* sh, float* out, float* out_count, const Region& r,
                 hipStream_t s) {
  if (!layer_has_guard_rows(sh, g, sizeof(float))) return false;
  Step5Args a = {};
  a.in = sh;
  a.out = out;
  a.out2 = out_count;
  a.crit = crit;
  a.rcrit = 1.0 / crit;
  float lo = (float)crit;  // largest float <= crit
  if ((double)lo > crit) lo = nextafterf(lo, -INFINITY);
  a.crit_lo = lo;
  a.ncrit = ncrit;
  return dispatch_step5(Q, true, g, a, r, s);
}

}  // namespace fast
}  // namespace te
