// te_march5.h -- the marching wavefront of te_march.h, rebuilt around what bounds these kernels on gfx950:
// a wave issues one instruction per ~8 cycles whatever its type, so a kernel of cheap 32-bit operations runs at
// (instructions per row) x 8 cycles / (waves per SIMD) -- both factors count, and the second is decided by registers.
//
// Same mathematics: one 64-lane wavefront owns 64 adjacent cells along i and marches down j; every row is reduced once
// along i into the nested run values S_w = reduce_{|di| <= w} f(i + di, r) and S_{hw(|r - j|)} is folded into the
// accumulator of every pending output row j; the rotation of the 2R+1 accumulators is resolved at compile time.
// What changed against the round-1 framework (PeriodLoader + one period of rows in LDS, 168 / 216 VGPRs, 3 / 2 waves
// per SIMD, 136 / 166 executed instructions per row):
//   * rows are staged two at a time ("a pass") from a prefetch queue of C passes: 3 loads per pass (two main rows and
//     ONE load for the halo columns of both rows, flattened over the lanes), no period of 25 staged values in registers;
//   * LDS holds the two rows of the pass only (2 x (64 + 2R) cells, 0.6-1.3 KB per wave: the LDS never limits the
//     residency); one wave per block, LDS operations of a wave execute in order: no barrier anywhere;
//   * the body is unrolled over P passes = 2P rows (two periods), so every pass folds two rows into each accumulator
//     with one 3-operand instruction and no odd row is left over; which LDS slot and which queue slot a row uses are
//     compile-time constants;
//   * invalid values are canonicalised with one v_fma (x * 0 + x: +-inf and NaN -> NaN, finite x -> x) instead of a
//     compare / wait state / select per staged value;
//   * strips are any number of rows (the march leaves at the first pass beyond its last row), sized by the launcher so
//     that the grid is one round of resident waves.
// The kernel-specific part is a policy K (see te_step5.hip, te_footprint5.hip).
#pragma once
#include "te_march.h"

namespace te {
namespace fast {

template <int V>
using ic = std::integral_constant<int, V>;

// Per-lane constants of a block: its own cell and its share of the halo columns of a pass.
// Lane L < 4R stages halo cell (row L / 2R of the pass, halo column L % 2R); the others repeat lane 4R - 1.
// All global accesses of a pass are  descriptor base (window column 0 = map column i0 - R of the strip's first row)  +
// running row offset  +  a per-lane byte offset that is never negative.
template <int R>
struct M5Lane {
  static constexpr int W = kLanes + 2 * R;
  int lane;
  int hrow;          // 0 / 1: which row of the pass my halo cell belongs to
  int hlds;          // its cell index in the two staged rows (hrow * W + window column)
  bool halo_in;      // the column lies inside the map
  unsigned o_main0, o_main1;  // byte offsets of my own cell in the first / second row of a pass
  unsigned o_halo;            // ... of my halo cell (my own cell's if the column lies outside the map)
  __device__ __forceinline__ void init(int lane_, int i0, int rows, int elem_bytes = 4) {
    lane = lane_;
    o_main0 = (unsigned)(R + lane) * (unsigned)elem_bytes;
    o_main1 = (unsigned)(rows + R + lane) * (unsigned)elem_bytes;
    if constexpr (R > 0) {
      constexpr int H2 = 2 * R;
      const int hl = lane < 2 * H2 ? lane : 2 * H2 - 1;
      hrow = hl / H2;
      const int hc = hl - hrow * H2;
      const int hcol = hc < R ? hc : kLanes + hc;
      halo_in = i0 - R + hcol >= 0 && i0 - R + hcol < rows;
      o_halo = (unsigned)(hrow * rows + (halo_in ? hcol : R + lane)) * (unsigned)elem_bytes;
      hlds = hrow * W + hcol;
    } else {
      hrow = 0;
      hlds = 0;
      halo_in = false;
      o_halo = o_main0;
    }
  }
};

// Global memory goes through raw buffer instructions: a 128-bit descriptor per layer in SGPRs (base = window column 0 of
// the strip's first row, built from blockIdx / kernel arguments only: provably wave-uniform), the lane's constant byte
// offset in a VGPR and the running row offset in an SGPR (`soffset`): buffer_load_dword v, v_off, s[desc], s_off offen --
// one instruction per access, one s_add per pass, no 64-bit address arithmetic in the lanes.  (num_records is the
// largest value: every address a pass touches lies in the context's slab, see kSlabGuardRows.)
typedef __amdgpu_buffer_rsrc_t brsrc;
__device__ __forceinline__ brsrc make_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), /*stride*/ 0, /*num_records*/ 0x7fffffff, /*flags*/ 0x00020000);
}
__device__ __forceinline__ float bload_f(brsrc rs, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}
__device__ __forceinline__ unsigned bload_u8(brsrc rs, unsigned voff, unsigned soff) {
  return (unsigned)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(rs, voff, soff, 0);
}
__device__ __forceinline__ void bstore_f(brsrc rs, unsigned voff, unsigned soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff, soff, 0);
}

// The march.  K provides
//   typedef Acc, Run;  static constexpr int C (passes in the prefetch queue)
//   load_pair(r, ic<q>)     prefetch map rows r, r + 1 (and their halo columns) into queue slot q (rows outside the map:
//                           whatever lies there -- the slab's guard rows or a neighbouring map / layer, see kSlabGuardRows)
//   stage_pair(r, ic<q>)    convert queue slot q = map rows r, r + 1 and write it to the LDS rows 0 / 1; rows outside the map: identity
//   build(ic<slot>, Run(&)[R+1])   nested run values of LDS row `slot`
//   reset(acc) / start<E, ROW>(acc, run) / fold1<E>(acc, run) / fold2<E>(acc, run1, run2)
//        E: offset (output row - input row) of the FIRST row folded; the second row of fold2 is at E - 1.
//        fold1 folds row 0 of the pass, fold2 rows 0 and 1, start row ROW (E == 0: that row holds the output's centre)
//   emit(ic<second>, j, acc)   output row j is complete (the policy checks js <= j < jend); second: the later row of the pass
//   rotate_queue(ic<n>)     queue slot s <- slot (s + n) % C   (after a body of P passes, n = P % C)
// Everything that depends on the row -- the policies' buffer offsets included -- is derived from ONE running scalar (r,
// two rows per pass): with r + constant per unrolled position the compiler precomputes all 2P of them per body and
// spills scalars to do so, and running offsets kept by the policies themselves ended up in vector registers (every
// buffer access then sits in a "waterfall" loop that proves its soffset uniform).
// One pass (two rows) at unrolled body position PC, then the rest of the body; true: the strip is finished.  (A chain of
// tail calls rather than a loop with a "done" flag: leaving through a flag merges the skipped passes' scalars as
// undefined values, which the compiler materialises as v_readfirstlane of an arbitrary VGPR -- one with a load in
// flight, and the pass then waits for that load.)
template <int Q, class K, int PC>
__device__ __forceinline__ bool march5_body(K& k, typename K::Acc (&acc)[Shape<Q>::P], int& r, const int r_end) {
  using S = Shape<Q>;
  constexpr int R = S::R, P = S::P, C = K::C;
  if constexpr (PC == P) {
    return false;
  } else {
    constexpr int b = 2 * PC;  // body position of the pass's first row
    constexpr int q = PC % C;
    if (__builtin_expect(r >= r_end, 0)) return true;
    // keep the passes apart: otherwise the scheduler hoists the LDS reads of several passes and their run values
    // are all live at once
    __builtin_amdgcn_sched_barrier(0);
    k.stage_pair(r, ic<q>{});
    k.load_pair(r + 2 * C, ic<q>{});
    typename K::Run s1[R + 1], s2[R + 1];
    k.build(ic<0>{}, s1);
    k.build(ic<1>{}, s2);
    if constexpr (R == 0) {  // a single-cell "disc": every row completes its own output
      k.template start<0, 0>(acc[0], s1[0]);
      k.emit(ic<0>{}, r, acc[0]);
      k.template start<0, 1>(acc[0], s2[0]);
      k.emit(ic<1>{}, r + 1, acc[0]);
    } else {
      static_for<P>([&](auto sc) __attribute__((always_inline)) {
        constexpr int sl = decltype(sc)::value;
        constexpr int e0 = ((sl - b % P) % P + P) % P;
        constexpr int e1 = e0 > R ? e0 - P : e0;  // slot sl holds output row (row b) + e1
        if constexpr (e1 == -R) {                  // row b is its last row; row b + 1 opens the next output of the slot
          k.template fold1<e1>(acc[sl], s1[S::hw(R)]);
          k.emit(ic<0>{}, r - R, acc[sl]);
          k.template start<R, 1>(acc[sl], s2[S::hw(R)]);
        } else {
          constexpr int w1 = S::hw(e1 < 0 ? -e1 : e1), w2 = S::hw(e1 - 1 < 0 ? 1 - e1 : e1 - 1);
          k.template fold2<e1>(acc[sl], s1[w1], s2[w2]);
          if constexpr (e1 - 1 == -R) {
            k.emit(ic<1>{}, r + 1 - R, acc[sl]);
            k.reset(acc[sl]);
          }
        }
      });
    }
    r += 2;
    return march5_body<Q, K, PC + 1>(k, acc, r, r_end);
  }
}

template <int Q, class K>
__device__ __forceinline__ void march5(K& k, const int js, const int jend) {
  using S = Shape<Q>;
  constexpr int R = S::R, P = S::P, C = K::C;
  // rows the unconditional loads reach beyond a layer: R above its first row, R + the prefetch distance (2C rows, two per
  // pass) + the pass's second row below its last -- the slack every layer a policy loads from must have (layer_has_guard_rows)
  static_assert(R + 2 * C + 2 <= kSlabGuardRows, "the march's loads must stay inside the guard rows around a layer");
  typename K::Acc acc[P];
  static_for<P>([&](auto c) __attribute__((always_inline)) { k.reset(acc[decltype(c)::value]); });
  int r = js - R;               // map row of the current pass's first row
  const int r_end = jend + R;   // one past the last input row of the strip
  static_for<C>([&](auto qc) __attribute__((always_inline)) { k.load_pair(r + 2 * decltype(qc)::value, qc); });
#pragma unroll 1
  while (!march5_body<Q, K, 0>(k, acc, r, r_end)) k.rotate_queue(ic<P % C>{});
  // (r stays "live" behind the loop: the early exits of the body meet in one latch block, and a scalar that is dead on
  // those paths arrives there as v_readfirstlane of an UNDEFINED vector register -- the allocator picks one with a load
  // in flight, and every other pass waits for that load)
  asm volatile("" ::"s"(r));
}

// Rows per strip such that (column blocks x maps x strips) fills `slots` resident waves in one round (at least 1 row;
// more than max_rows per strip gains nothing and keeps short maps from waiting on one long strip)
inline int plan_strip_rows(int rows, long columns, long slots, int max_rows = 512) {
  long strips = slots / (columns > 0 ? columns : 1);
  if (strips < 1) strips = 1;
  long per = (rows + strips - 1) / strips;
  if (per > max_rows) per = max_rows;
  if (per < 1) per = 1;
  return (int)per;
}

__device__ __forceinline__ float canon_nan(float x) {  // finite x -> x (exactly), +-inf / NaN -> NaN
  return __builtin_fmaf(x, 0.0f, x);
}

}  // namespace fast
}  // namespace te
