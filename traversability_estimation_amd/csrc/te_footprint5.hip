// te_footprint5.hip -- the sum of the circular footprint pass in SCATTER form (te_march5.h), 32-bit fixed point.
//
//   TraversabilityMap::traversabilityFootprint(radius, offset)   traversability_estimation/src/TraversabilityMap.cpp:307-318
//     -> isTraversable(center, radiusMax, traversability, radiusMin)              :654-746
//
// k_fp_slide4 (te_footprint4.hip) SLIDES the disc sum: 2(2R+1) LDS reads per row from a ring of 2R+2 rows (6.5 KB per
// wave: 4 waves per SIMD) and is bound by those reads (ds_read2_b32: 4 LDS cycles each, SQ_WAIT_INST_LDS 29 %).  A sum
// can be folded like the step filter's maxima instead: every row is reduced ONCE along i into its nested run sums
// (2R reads, R v_add3_u32) and the run sum of the matching half-width is added to each of the 2R+1 pending output rows.
// Half the LDS instructions, two rows of LDS per wave instead of twenty, under 96 registers: 5 waves per SIMD.
// A cell is one word
//     cell = round(T' * 2^k) | U << 27        T' = traversability (NaN -> default), U = 1: fails isTraversableForFilters
// with k chosen by the host such that the T-sum of a whole disc stays below 2^27 (k = 19 at R = 9, as in k_fp_slide4:
// the two kernels give the same bits).  A run sum holds at most 2R+1 <= 31 cells and cannot overflow 32 bits; the
// accumulators add with SATURATION (v_add_u32 ... clamp), so a disc's word is
//     < 2^27:  no untraversable cell, and the word IS the exact T-sum  ->  mean = T * 2^-k / n   (:732-735)
//     >= 2^27: at least one untraversable cell (however many: the sum sticks at 2^32 - 1 instead of wrapping)
// and nothing else is needed from it: a disc with an untraversable cell is 0 if that cell is its own centre or the
// inner radius is 0 (:694-704); otherwise its cell goes onto the list k_fp_blocked walks afterwards (te_footprint4.hip),
// exactly as k_fp_slide4 does.  Tie radii (cells exactly on the circle, decided per centre) stay with k_fp_slide4<Q, true>.
#include "te_internal.h"
#include "te_march5.h"

#include <type_traits>

namespace te {
namespace fast {

namespace {

constexpr int kF5Waves = 5;
constexpr int kF5UBit = 27;

struct F5Args {
  const float* trav;
  const uint8_t* untrav;
  float* footprint;
  int rows, cols;
  long long map_cells;
  int strip_rows;
  // the part of the map this launch covers: block columns [bx0, bx0 + nbx_l), output rows [j_lo, j_hi), map (< 0: blockIdx.z)
  int bx0, nbx_l, j_lo, j_hi, map;
  const int* gtab;  // clip table of the disc: {n, ...} per (ky, kx)
  double rmin;      // inner radius: 0 makes every disc with an untraversable cell 0 (:694-704), no walk needed
  float def, scale;    // cell = (unsigned)(T' * scale + 0.5) | U << 27, scale = 2^k
  double inv_scale;    // 2^-k
  unsigned* blocked_list;   // cells (index into the layer, all maps) whose disc holds an untraversable cell ...
  unsigned* blocked_count;  // ... [0] entries of the list, [1] entries that hold a cell, [4] entries of the scratch reserved (k_fp_mask resets them)
  unsigned* scratch;        // where a block collects its cells until its strip is done (Layers::fp_scratch)
  int chunk;                // granule of the scratch reservations
  size_t list_cap;          // (host side: the launcher refuses a grid whose unfinished chunks might not fit)
  // one byte per 64 x 4 cells, written by k_fp_mask: "holds an untraversable cell" (Layers::untrav_flags).  A strip whose
  // flags are all clear -- on terrain without obstacles every strip -- does not fetch its mask bytes: the byte loads stay
  // in the instruction stream, but their descriptor is given zero records, and a raw buffer load beyond its records
  // returns 0 without a memory request.  (Built without the loads, for timing only, the footprint pass went from 0.160 to
  // 0.148-0.154 ms: profiles/r04_experiments.json.)
  const uint8_t* untrav_flags;
  int flag_ntx, flag_nfy;
  // THE INNER DISC.  A disc that holds an untraversable cell is 0 when the FIRST such cell of the spiral lies within the
  // inner radius (:694-704: getCurrentRadius() = ring * res <= radiusMin), and rings are walked in ascending order: so
  // it is 0 iff some untraversable cell of the disc has an integer norm floor(|offset|) <= r_in, r_in the last ring with
  // r_in * res <= radiusMin -- a question about a smaller disc, which the march answers on the way (strips that hold an
  // untraversable cell at all: inner_row below).  Only the discs whose untraversable cells ALL lie between the two radii go
  // onto the list for k_fp_blocked: a third of what rounds 3-4 listed next to a kerb (radii 6 and 9 cells), and exactly
  // the ones that need the weighted mean of :705-711.
  int r_in;                  // -1: no inner radius (radiusMin = 0 is handled by rmin_zero)
  int k_start;               // spiral entries 0 .. k_start-1 are the inner disc's: none of them is untraversable in a listed disc
  unsigned inner_tab[17];    // [g], g = distance along the row to the nearest untraversable cell (r_in + 1: none within r_in):
                             // bit b set <=> a cell at that distance in a row |r_in - b| rows away lies within the inner disc
};

template <int Q>
struct SlideK {
  static constexpr int R = Shape<Q>::R, W = kLanes + 2 * R, C = 2;
  typedef unsigned Acc;
  typedef unsigned Run;
  const F5Args& a;
  M5Lane<R> L;   // float layers
  unsigned ob_main0, ob_main1, ob_halo;  // the same offsets in the byte layer
  brsrc rs_t, rs_u, rs_out;       // window column 0 of map row js - R (inputs) / js - 2R (output)
  unsigned row_bytes;
  int js, nout, r0;               // r0 = js - R: the strip's first input row (the descriptors' row 0)
  float tm0[C], tm1[C], th[C];
  unsigned um0[C], um1[C], uh[C];
  unsigned* lds;   // [2][W]
  unsigned c0, c1;
  // inner disc (strips that hold an untraversable cell): hb bit b = "output row (current input row) - r_in + b has an
  // untraversable cell within the inner disc among the rows staged so far"; hist: the finished answers, newest row in bit 0
  unsigned hb, hist;
  bool dirty;  // (uniform) some flag of the strip's window is set: the mask bytes are loaded and the inner disc is tracked
  int ri;
  unsigned in_word, in_sh;  // which 32-bit word of a row's window bits my run starts in, and at which bit
  size_t mo;
  int icol, kx, nt_mid;
  bool own, rmin_zero;
  float rnt;
  // the list: a block collects its cells in a SCRATCH reservation made when it first lists one -- room for every cell the
  // rest of its strip could list -- and copies them into the list proper when the strip is done, where it then reserves
  // exactly what it needs (rounded up to a wavefront's 64 entries, the rest kF4NoCell): two atomics per listing block, and
  // a dense list for k_fp_blocked.  (Rounds 3-4 reserved list chunks of 256 entries as they went: a strip that runs along a
  // kerb paid an atomic round trip every four to six rows, 20 us of a 50 us kernel with three boxes on the map; larger
  // chunks left k_fp_blocked more unused entries to step over, and handing it the sparse reservations themselves --
  // round 5's first attempt, with page counts -- unbalanced its wavefronts: 300 boxes 170 -> 290 us.)
  unsigned res_base;
  int res_len, listed_total;

  __device__ __forceinline__ SlideK(const F5Args& a_) : a(a_) {}

  __device__ __forceinline__ void init(int lane, int i0, int own_lo, int js_, int jend, size_t mo_, unsigned* lds_) {
    L.init(lane, i0, a.rows);
    ob_main0 = L.o_main0 / 4u;
    ob_main1 = L.o_main1 / 4u;
    ob_halo = L.o_halo / 4u;
    js = js_;
    nout = jend - js_;
    mo = mo_;
    lds = lds_;
    r0 = js_ - R;
    row_bytes = (unsigned)a.rows * 4u;
    rs_t = make_rsrc(a.trav + mo + ((long long)(js_ - R) * a.rows + (i0 - R)));
    {
      // the flags of every 64 x 4 block the strip's window touches: columns i0 - R .. i0 + 63 + R, rows js - R .. jend - 1 + R
      const int c0 = (i0 - R < 0 ? 0 : i0 - R) >> 6, c1 = (i0 + kLanes - 1 + R > a.rows - 1 ? a.rows - 1 : i0 + kLanes - 1 + R) >> 6;
      const int f0 = (js_ - R < 0 ? 0 : js_ - R) >> 2, f1 = (jend - 1 + R > a.cols - 1 ? a.cols - 1 : jend - 1 + R) >> 2;
      const int nc = c1 - c0 + 1, n = nc * (f1 - f0 + 1);
      const uint8_t* fl = a.untrav_flags + (size_t)(a.map >= 0 ? a.map : (int)blockIdx.z) * (size_t)a.flag_nfy * (size_t)a.flag_ntx;
      unsigned any = 0;
      for (int k = lane; k < n; k += kLanes) any |= fl[(size_t)(f0 + k / nc) * a.flag_ntx + (size_t)(c0 + k % nc)];
      const bool clean = !__any(any != 0u);  // (uniform)
      dirty = !clean && a.r_in >= 0;
      const void* ub = a.untrav + mo + ((long long)(js_ - R) * a.rows + (i0 - R));
      rs_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ub), /*stride*/ 0, /*num_records*/ clean ? 0 : 0x7fffffff, /*flags*/ 0x00020000);
    }
    rs_out = make_rsrc(a.footprint + mo + ((long long)(js_ - 2 * R) * a.rows + (i0 - R)));
    icol = i0 + lane;
    own = icol >= own_lo;
    rmin_zero = a.rmin == 0.0;
    kx = icol < R ? R - icol : (a.rows - 1 - icol < R ? -(R - (a.rows - 1 - icol)) : 0);
    nt_mid = a.gtab[((0 + R) * (2 * R + 1) + (kx + R)) * 6];  // cells of my disc on a row away from the top / bottom
    rnt = (float)(a.inv_scale / (double)nt_mid);
    hb = hist = 0u;
    ri = a.r_in >= 0 ? a.r_in : 0;
    in_word = (unsigned)(lane + R - ri) >> 5;
    in_sh = (unsigned)(lane + R - ri) & 31u;
    if (lane < 17) lds[2 * W + lane] = a.inner_tab[lane];  // (LDS operations of a wave execute in order: visible to the reads below)
    res_base = 0;
    res_len = 0;
    listed_total = 0;
  }
  // (unconditional: see Step5Base::load_pair in te_step5.hip and kSlabGuardRows)
  template <int q>
  __device__ __forceinline__ void load_pair(int r, ic<q>) {
    const unsigned sb = (unsigned)(r - r0) * (unsigned)a.rows, so = sb * 4u;  // (uniform; the mask is one byte per cell)
    tm0[q] = bload_f(rs_t, L.o_main0, so);
    tm1[q] = bload_f(rs_t, L.o_main1, so);
    th[q] = bload_f(rs_t, L.o_halo, so);
#ifdef TE_F5_NO_U  // (measurement only: the mask bytes not loaded -- what the three byte loads of a pass cost on a map without obstacles)
    um0[q] = um1[q] = uh[q] = 0u;
    (void)sb;
#else
    um0[q] = bload_u8(rs_u, ob_main0, sb);
    um1[q] = bload_u8(rs_u, ob_main1, sb);
    uh[q] = bload_u8(rs_u, ob_halo, sb);
#endif
  }
  template <int n>
  __device__ __forceinline__ void rotate_queue(ic<n>) {
    if constexpr (n % C != 0) {
      static_assert(C == 2, "a queue of two passes");
      const float f0 = tm0[0], f1 = tm1[0], fh = th[0];
      const unsigned u0 = um0[0], u1 = um1[0], u2 = uh[0];
      tm0[0] = tm0[1];
      tm1[0] = tm1[1];
      th[0] = th[1];
      um0[0] = um0[1];
      um1[0] = um1[1];
      uh[0] = uh[1];
      tm0[1] = f0;
      tm1[1] = f1;
      th[1] = fh;
      um0[1] = u0;
      um1[1] = u1;
      uh[1] = u2;
    }
  }
  __device__ __forceinline__ unsigned pack(float t, unsigned u) const {
    const float tt = __builtin_isfinite(t) ? t : a.def;  // :719-724
    // round(T' * 2^k): the product is exact, + 0.5 is exact below 2^23, the conversion truncates (and clamps at 0)
    return (unsigned)__builtin_fmaf(tt, a.scale, 0.5f) | (u << kF5UBit);
  }
  // One staged row for the inner disc.  mm: my 64 main cells' untraversable bits; hb2: the row's 2R halo cells (bits 0..R-1
  // the columns left of the block, R..2R-1 the columns right of it), cells outside the map already 0.  The row's window is
  // a bit string of W <= 94 bits in three uniform words; a lane needs the 2 r_in + 1 bits around its own column, which start
  // at bit in_sh = lane + R - r_in: a funnel shift (v_alignbit) of the two words that hold them -- which two is a property
  // of the lane (in_word: 0, 1 or 2), not of the row.
  __device__ __forceinline__ void inner_row(unsigned long long mm, unsigned long long hb2) {
    constexpr unsigned long long RM = (1ull << R) - 1ull;
    const unsigned long long lo = (hb2 & RM) | (mm << R);                 // window columns 0 .. 63
    const unsigned long long hi = (mm >> (64 - R)) | ((hb2 >> R) << R);   // window columns 64 .. W-1
    const unsigned w0 = (unsigned)lo, w1 = (unsigned)(lo >> 32), w2 = (unsigned)hi, w3 = (unsigned)(hi >> 32);  // (uniform)
    const unsigned wl = in_word == 0 ? w0 : (in_word == 1 ? w1 : w2);
    const unsigned wh = in_word == 0 ? w1 : (in_word == 1 ? w2 : w3);
    const unsigned x = __builtin_amdgcn_alignbit(wh, wl, in_sh) & ((2u << (2 * ri)) - 1u);  // bit ri: my own cell
    const unsigned left = x & ((2u << ri) - 1u);                              // bits 0 .. ri: my cell and the ones before it
    const unsigned dleft = (unsigned)__clz((int)left) - (unsigned)(31 - ri);  // (no such cell: __clz(0) = 32 -> ri + 1)
    const unsigned dright = (unsigned)(__ffs((int)(x >> ri)) - 1);            // (no such cell: __ffs(0) = 0 -> 0xffffffff)
    const unsigned g = dleft < dright ? dleft : dright;                       // 0 .. ri + 1
    hb |= lds[2 * W + g];
    hist = (hist << 1) | (hb & 1u);
    hb >>= 1;
  }
  template <int q>
  __device__ __forceinline__ void stage_pair(int r, ic<q>) {
    c0 = pack(tm0[q], um0[q]);
    c1 = pack(tm1[q], um1[q]);
    unsigned vh = L.halo_in ? pack(th[q], uh[q]) : 0u;  // cells outside the map: nothing
    unsigned u0 = um0[q], u1 = um1[q];
    if (__builtin_expect((unsigned)r >= (unsigned)(a.cols - 1), 0)) {  // a row of the pass outside the map
      const bool in0 = (unsigned)r < (unsigned)a.cols, in1 = (unsigned)(r + 1) < (unsigned)a.cols;
      c0 = in0 ? c0 : 0u;
      c1 = in1 ? c1 : 0u;
      u0 = in0 ? u0 : 0u;
      u1 = in1 ? u1 : 0u;
      vh = (L.hrow ? in1 : in0) ? vh : 0u;
    }
    lds[R + L.lane] = c0;
    lds[W + R + L.lane] = c1;
    lds[L.hlds] = vh;
    if (__builtin_expect(dirty, 0)) {  // (uniform)
      constexpr unsigned long long HM = (1ull << (2 * R)) - 1ull;
      const unsigned long long m0 = __ballot(u0 != 0u), m1 = __ballot(u1 != 0u);
      const unsigned long long mh = __ballot((vh >> kF5UBit) != 0u);  // lanes 0 .. 2R-1: row r, 2R .. 4R-1: row r + 1 (the others repeat lane 4R-1)
      if ((m0 | m1 | mh) == 0ull) {  // (uniform) no untraversable cell in either row's window: nothing to add, two answers move on
        hist = (hist << 2) | ((hb & 1u) << 1) | ((hb >> 1) & 1u);
        hb >>= 2;
      } else {
        inner_row(m0, mh & HM);
        inner_row(m1, (mh >> (2 * R)) & HM);
      }
    }
  }
  template <int slot>
  __device__ __forceinline__ void build(ic<slot>, Run (&s)[R + 1]) {
    const unsigned* row = lds + slot * W + R + L.lane;
    s[0] = slot ? c1 : c0;
    static_for<R>([&](auto dc) __attribute__((always_inline)) {
      constexpr int d = decltype(dc)::value + 1;
      s[d] = s[d - 1] + row[-d] + row[d];  // (at most 2R+1 <= 31 cells: no overflow)
    });
  }
  __device__ __forceinline__ void reset(Acc& x) { x = 0u; }
  template <int E, int ROW>
  __device__ __forceinline__ void start(Acc& x, const Run& s) {
    x = s;
  }
  template <int E>
  __device__ __forceinline__ void fold1(Acc& x, const Run& s) {
    x = __builtin_elementwise_add_sat(x, s);
  }
  template <int E>
  __device__ __forceinline__ void fold2(Acc& x, const Run& s1, const Run& s2) {
    x = __builtin_elementwise_add_sat(__builtin_elementwise_add_sat(x, s1), s2);
  }
  template <int second>
  __device__ __forceinline__ void emit(ic<second>, int j, const Acc& x) {
    if (!((unsigned)(j - js) < (unsigned)nout)) return;  // (uniform)
    float rn = rnt;
    if (__builtin_expect((unsigned)(j - R) >= (unsigned)(a.cols - 2 * R), 0)) {  // a row of the top / bottom frame (uniform)
      const int ky = j < R ? R - j : -(R - (a.cols - 1 - j));
      const int nt = a.gtab[((ky + R) * (2 * R + 1) + (kx + R)) * 6];
      rn = (float)(a.inv_scale / (double)nt);
    }
    float out = (float)x * rn;  // :732-735 no untraversable cell in the footprint: the mean (T < 2^27: the conversion is good to 2^-25)
    const bool blocked = x >= (1u << kF5UBit);
    const unsigned so = (unsigned)(j - (r0 - R)) * row_bytes;
    if (__builtin_expect(__any(blocked), 0)) {
      if (rmin_zero) {
        out = blocked ? 0.0f : out;  // :694-704 radiusMin = 0: the first untraversable cell, wherever it lies, gives 0
      } else {
        // An untraversable cell within the inner radius -- the centre itself is ring 0 -- makes the disc 0 (:694-704); the
        // answer for row j was finished when row j + r_in was staged, R - r_in rows before the row that completed the sum
        // (this pass's rows are bits 1 and 0 of hist).  The other discs with an untraversable cell go onto the list:
        // k_fp_blocked walks their spirals and stores their values after this kernel; nothing is stored for them here --
        // not by the block that lists them and not by a shifted last block that shares the column (on a region run its
        // neighbour need not be part of the launch, and the cell then keeps the value it has).
        const bool inner = ((hist >> (unsigned)((second ? R : R + 1) - ri)) & 1u) != 0u;
        out = (blocked && inner) ? 0.0f : out;
        const bool walk = blocked && !inner;
        const bool listed = walk && own;
        const unsigned long long bm = __ballot(listed);
        if (bm != 0ull) {
          const int n = __popcll(bm);
          const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0u));
          if (res_len == 0) {  // (uniform) rows are emitted in ascending order: this row and the ones below it can list 64 cells each
            const int rows_left = nout - (j - js);
            const unsigned want = ((unsigned)(rows_left * kLanes) + (unsigned)a.chunk - 1u) / (unsigned)a.chunk * (unsigned)a.chunk;
            unsigned base = 0;
            if (L.lane == 0) base = atomicAdd(a.blocked_count + 4, want);
            res_base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
            res_len = (int)want;
          }
          if (listed) a.scratch[res_base + (unsigned)(listed_total + rank)] = (unsigned)(mo + (size_t)j * a.rows + icol);
          listed_total += n;
        }
        if (!walk) bstore_f(rs_out, L.o_main0, so, out);
        return;
      }
    }
    // (a shifted last block stores the columns it shares with its neighbour too: the same bits)
    bstore_f(rs_out, L.o_main0, so, out);
  }
  __device__ __forceinline__ void finish() {
    if (listed_total == 0) return;  // (uniform)
    // my cells from the scratch into the list, padded to whole wavefronts (the stores above were this wave's own: a
    // workgroup-scope fence orders them before the loads below; the CU's L1 is coherent for its own waves)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    const unsigned want = ((unsigned)listed_total + (unsigned)kLanes - 1u) & ~((unsigned)kLanes - 1u);
    unsigned base = 0;
    if (L.lane == 0) {
      base = atomicAdd(a.blocked_count, want);
      atomicAdd(a.blocked_count + 1, (unsigned)listed_total);
    }
    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    for (unsigned q = (unsigned)L.lane; q < want; q += (unsigned)kLanes)
      a.blocked_list[base + q] = q < (unsigned)listed_total ? a.scratch[res_base + q] : kF4NoCell;
  }
};

template <int Q>
__global__ __launch_bounds__(kLanes) __attribute__((amdgpu_waves_per_eu(kF5Waves, kF5Waves))) void k_fp_slide5(F5Args a) {
  constexpr int R = Shape<Q>::R, W = kLanes + 2 * R;
  __shared__ unsigned lds[2 * W + 17];  // the two staged rows, the inner disc's table
  const int lane = threadIdx.x;
  const int bx = a.bx0 + (int)blockIdx.x % a.nbx_l, strip = (int)blockIdx.x / a.nbx_l;
  int i0 = bx * kLanes;
  i0 = i0 + kLanes > a.rows ? a.rows - kLanes : i0;  // the last block ends at the map edge (rows >= 64)
  const int js = a.j_lo + strip * a.strip_rows;
  if (js >= a.j_hi) return;
  const int jend = js + a.strip_rows < a.j_hi ? js + a.strip_rows : a.j_hi;
  const size_t mo = (size_t)(a.map >= 0 ? a.map : (int)blockIdx.z) * (size_t)a.map_cells;
  SlideK<Q> k(a);
  // the last block of a row of blocks is shifted left: the columns it shares with its neighbour are the neighbour's
  // (one list entry per cell; both store the same value)
  if (blockIdx.x == 0 && blockIdx.z == 0 && lane == 0) a.blocked_count[3] = (unsigned)a.k_start;  // for k_fp_blocked: the first spiral entry that can be untraversable in a listed disc
  k.init(lane, i0, bx * kLanes, js, jend, mo, lds);
  march5<Q>(k, js, jend);
  k.finish();
}

template <int Q>
bool launch_f5(const F5Args& a0, int batch, hipStream_t s) {
  F5Args a = a0;
  constexpr int R = Shape<Q>::R;
  static_assert(2 * R + 1 <= 31, "a run sum of 2R+1 cells with the untraversable flag at bit 27 must fit 32 bits");
  static const int waves_env = lab_int("TE_F5_WAVES", 0);  // measurement aid: strips sized for this many waves per SIMD
  const int capacity = (waves_env > 0 ? waves_env : kF5Waves) * 4 * device_cus();
  const int nz = a.map >= 0 ? 1 : (batch > 0 ? batch : 1);
  const int H = a.j_hi - a.j_lo;
  static const int max_strip = lab_int("TE_F5_MAX_STRIP", 512);
  a.strip_rows = plan_strip_rows(H, (long)a.nbx_l * nz, capacity, max_strip > 0 ? max_strip : 512);
  const int sr = a.strip_rows;
  a.chunk = sr >= 4 ? kF4Chunk : (sr * kLanes >= kF4Chunk / 2 ? kF4Chunk / 2 : kLanes);  // (a strip of one row lists at most 64 cells)
  const int nstrips = (H + sr - 1) / sr;
  // a block reserves at most its own cells rounded up to whole pages
  // (64 entries per row of every block -- a shifted last block reserves for the columns it shares with its neighbour too)
  if ((double)a.nbx_l * (double)nz * ((double)H * (double)kLanes + (double)nstrips * (double)a.chunk) > (double)a.list_cap) return false;
  const dim3 grid((unsigned)(a.nbx_l * nstrips), 1, (unsigned)nz);
  hipLaunchKernelGGL((k_fp_slide5<Q>), grid, dim3(kLanes), 0, s, a);
  return true;
}

}  // namespace

// Shapes: every disc shape up to radius 10 (te_march.h) except the single cell, and for radii 11 .. 15 every sum of two
// squares below 256.  Compiled in TE_PARTS parts like te_footprint4.hip (build.py).
#define TE_F5_P0(X) X(4) X(16) X(26) X(37) X(50) X(65) X(73) X(85) X(100) X(121) X(136) X(148) X(162) X(178) X(193) X(202) X(212) X(229)
#define TE_F5_P1(X) X(10) X(13) X(25) X(36) X(49) X(64) X(82) X(98) X(109) X(117) X(130) X(146) X(160) X(173) X(185) X(200) X(226) X(241) X(250)
#define TE_F5_P2(X) X(9) X(20) X(34) X(45) X(58) X(61) X(81) X(97) X(106) X(116) X(128) X(145) X(157) X(170) X(181) X(197) X(225) X(234) X(245)
#define TE_F5_P3(X) X(2) X(8) X(18) X(32) X(41) X(53) X(72) X(80) X(90) X(104) X(113) X(125) X(144) X(153) X(169) X(196) X(208) X(221) X(233) X(244)
#define TE_F5_P4(X) X(1) X(5) X(17) X(29) X(40) X(52) X(68) X(74) X(89) X(101) X(122) X(137) X(149) X(164) X(180) X(194) X(205) X(218) X(232) X(242)
#if !defined(TE_PARTS) || defined(TE_F5_SHAPES)
#undef TE_PARTS
#undef TE_PART
#define TE_PARTS 1
#define TE_PART 0
#endif
#if TE_PARTS != 1 && TE_PARTS != 5
#error "te_footprint5.hip is cut into 1 or 5 parts"
#endif
#ifndef TE_F5_SHAPES
#if TE_PARTS == 1
#define TE_F5_SHAPES(X) TE_F5_P0(X) TE_F5_P1(X) TE_F5_P2(X) TE_F5_P3(X) TE_F5_P4(X)
#else
#define TE_F5_CAT2(a, b) a##b
#define TE_F5_CAT(a, b) TE_F5_CAT2(a, b)
#define TE_F5_SHAPES(X) TE_F5_CAT(TE_F5_P, TE_PART)(X)
#endif
#endif
#define TE_F5_NAME2(k) f5_launch_part##k
#define TE_F5_NAME(k) TE_F5_NAME2(k)

// launches shape Q if it belongs to this part
bool TE_F5_NAME(TE_PART)(int Q, const void* args, int batch, hipStream_t s) {
  const F5Args& a = *static_cast<const F5Args*>(args);
  switch (Q) {
#define X(q) \
  case q:    \
    return launch_f5<q>(a, batch, s);
    TE_F5_SHAPES(X)
#undef X
    default:
      return false;
  }
}

#if TE_PART == 0
#if TE_PARTS > 1
bool f5_launch_part1(int Q, const void* args, int batch, hipStream_t s);
bool f5_launch_part2(int Q, const void* args, int batch, hipStream_t s);
bool f5_launch_part3(int Q, const void* args, int batch, hipStream_t s);
bool f5_launch_part4(int Q, const void* args, int batch, hipStream_t s);
#endif

// The scatter-form sum kernel of the footprint pass for a tie-free disc of an instantiated shape; false: not taken
// (the caller tries k_fp_slide4, then the double kernel).  tcap: upper bound of the finite values of the traversability
// layer, as the host can prove it (the layer was written by the chain: w_scale * (w_slope + w_step + w_rough) with
// non-negative weights); < 0: unknown.  On success the caller still owes footprint_blocked4 (finish = false) for the
// listed cells.
bool footprint_slide5(const Geo& g, const FootprintParams& p, const Layers& L, const int* clip_table, double tcap, hipStream_t s,
                      const Region* region, bool* needs_blocked) {
  const Disc& d = p.fp_disc;
  static const bool off = lab_flag("TE_NO_F5");  // measurement aid: k_fp_slide4 as in round 3
  if (off || d.n_ties != 0) return false;
  const int shape = d.Q, R = d.R;
  if (shape < 1 || R < 1 || 2 * R + 1 > 31 || p.reach != R || g.rows < kLanes || g.rows < 2 * R + 1 || g.cols < 2 * R + 1) return false;
  if ((double)g.rows * (double)g.cols * 4.0 >= 4294967296.0) return false;  // 32-bit list entries and byte offsets within a pass
  if (!L.fp_blocked || !L.fp_scratch || !L.fp_blocked_count || (double)g.rows * (double)g.cols * (double)g.batch > (double)L.fp_blocked_cap) return false;
  // the fixed-point scale: the T-sum of a whole disc (npoints cells of at most cap * 2^k + 1/2 each) must stay below
  // 2^27 -- the untraversable flag's bit -- and the default value that replaces NaN has to fit as well
  if (!(tcap >= 0.0) || !(p.def >= 0.0)) return false;
  // (the march loads rows beyond the layers it is given, te_internal.h; the mask bytes are a quarter of a float layer's
  // rows in bytes, so the float test of the mask's own rows is the stricter one)
  if (!layer_has_guard_rows(L.trav, g, sizeof(float)) || !layer_has_guard_rows(L.untrav, g, sizeof(uint8_t))) return false;
  const double cap = (tcap > p.def ? tcap : p.def) * (1.0 + 1e-6) + 1e-12;
  int k = 23;
  while (k >= 0 && (double)d.npoints * (cap * ldexp(1.0, k) + 1.0) >= (double)(1u << kF5UBit)) --k;
  if (k < 17) return false;  // rounding each value to 2^-17 could show at the 1e-5 level: the double kernel serves
  F5Args a;
  a.trav = L.trav;
  a.untrav = L.untrav;
  a.footprint = L.footprint;
  a.rows = g.rows;
  a.cols = g.cols;
  a.map_cells = (long long)g.rows * g.cols;
  a.strip_rows = 0;
  const int nbx = (g.rows + kLanes - 1) / kLanes;
  a.bx0 = region ? region->i0 / kLanes : 0;
  a.nbx_l = region ? (region->i1 - 1) / kLanes - a.bx0 + 1 : nbx;
  a.j_lo = region ? region->j0 : 0;
  a.j_hi = region ? region->j1 : g.cols;
  a.map = region ? region->map : -1;
  *needs_blocked = false;
  if (a.j_hi <= a.j_lo || a.nbx_l <= 0) return true;
  a.gtab = clip_table;
  a.rmin = p.rmin;
  a.def = (float)p.def;
  a.scale = (float)ldexp(1.0, k);
  a.inv_scale = ldexp(1.0, -k);
  {
    // the last ring within the inner radius, with the comparison k_fp_blocked and the reference make (:700: getCurrentRadius() <= radiusMin)
    int r_in = -1;
    while (r_in + 1 <= R && (double)(r_in + 1) * g.res <= p.rmin) ++r_in;
    a.r_in = p.rmin > 0.0 ? r_in : -1;
    // the spiral visits the cells of ring <= r_in first (rings in ascending order; the disc's two outer rings only where they lie in the disc)
    a.k_start = 0;
    if (a.r_in >= 0)
      for (int dj = -d.R; dj <= d.R; ++dj)
        for (int di = -d.hw[dj < 0 ? -dj : dj]; di <= d.hw[dj < 0 ? -dj : dj]; ++di) a.k_start += di * di + dj * dj < (a.r_in + 1) * (a.r_in + 1) ? 1 : 0;
    for (int gg = 0; gg < 17; ++gg) a.inner_tab[gg] = 0u;
    if (a.r_in >= 0) {
      const int ri = a.r_in;
      for (int gg = 0; gg <= ri; ++gg)  // (gg = ri + 1: no untraversable cell of the row within ri columns: no bit)
        for (int b = 0; b <= 2 * ri; ++b) {
          const int dj = ri - b < 0 ? b - ri : ri - b;
          // cells (di, dj) of ring <= ri: di^2 + dj^2 < (ri + 1)^2; the two outer rings of the spiral keep only cells of the disc
          int w = -1;
          while ((w + 1) * (w + 1) + dj * dj < (ri + 1) * (ri + 1)) ++w;
          if (dj <= d.R && d.hw[dj] < w) w = d.hw[dj];
          if (dj > d.R) w = -1;
          if (gg <= w) a.inner_tab[gg] |= 1u << b;
        }
    }
  }
  a.blocked_list = L.fp_blocked;
  a.blocked_count = L.fp_blocked_count;
  a.scratch = L.fp_scratch;
  a.chunk = kF4Chunk;
  a.list_cap = L.fp_blocked_cap;
  a.untrav_flags = L.untrav_flags;
  a.flag_ntx = untrav_flag_ntx(g.rows);
  a.flag_nfy = untrav_flag_nfy(g.cols);
  bool launched = f5_launch_part0(shape, &a, g.batch, s);
#if TE_PARTS > 1
  launched = launched || f5_launch_part1(shape, &a, g.batch, s) || f5_launch_part2(shape, &a, g.batch, s) || f5_launch_part3(shape, &a, g.batch, s) ||
             f5_launch_part4(shape, &a, g.batch, s);
#endif
  *needs_blocked = launched;
  return launched;
}
#endif  // TE_PART == 0

}  // namespace fast
}  // namespace te
