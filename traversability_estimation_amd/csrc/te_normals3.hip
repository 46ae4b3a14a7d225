// te_normals3.hip -- NormalVectorsFilter + SlopeFilter + RoughnessFilter for cells whose disc lies inside the map,
// third formulation of the sliding-disc kernel for gfx950.
//
//   NormalVectorsFilter (area method; un-vendored grid_map_filters, call site
//                        traversability_estimation/config/robot_filter_parameter.yaml:3-9)
//   SlopeFilter::update      traversability_estimation_filters/src/SlopeFilter.cpp:59-88
//   RoughnessFilter::update  traversability_estimation_filters/src/RoughnessFilter.cpp:73-132
//
// Same mathematics as te_slide_normals.hip (one wavefront owns 64 adjacent cells along i and marches down j; each
// lane slides the four double z-moments of its disc by the 2R+1 cells of the leading and of the trailing edge), laid
// out for what bounds this code on gfx950: ONE WAVE ISSUES AT MOST ONE INSTRUCTION PER ~8.3 CYCLES, whatever its type
// (profiles/r01_valu_rates.txt), while a SIMD retires one fp64 operation per ~4.3 cycles.  Two waves per SIMD can
// therefore keep the fp64 pipe busy only if they issue nothing but fp64 operations; the previous kernel (2 waves per
// SIMD, ~440 instructions per row of which 190 fp64) ran at the per-wave issue limit with the fp64 pipe half idle.
// Here:
//   * 3 waves per SIMD: <= 168 VGPRs and a ring of exactly 2R+2 rows (12 x 13.1 KB per CU at R = 9);
//   * the ring is addressed through NR/C "chunk" base registers that rotate once every C rows, the row loop is unrolled
//     C times and every LDS address is  base register + immediate: no per-row address arithmetic, no scalar ring
//     bookkeeping (the previous kernel spent ~130 scalar instructions per row, mostly on that);
//   * the y-moment uses  Sjz' = Sjz - (Sz + Sz')/2 + sum_g (h_g + 1/2) * V_g,  V_g = sum of (lead + trail) over the
//     columns of half-height h_g: one add per column and one fma per distinct height instead of an fma + add per column;
//   * the tail keeps fp64 only where the result needs it: D = N*Szz - Sz^2, s = sqrt(delta^2 + h2) (float32 rsq seed +
//     one Newton step: 2^-45), the smallest eigenvalue (delta + D) - s for the roughness, and  m = 1 - nz^2 =
//     h2 / (2 s t)  (float32 rcp seed + one Newton step).  nz = sqrt(1 - m) is evaluated as the series
//     1 - m/2 - m^2/8 - m^3/16 in double where float32 rounding of nz decides the slope (m < 2^-8: |error| < 2^-36),
//     and as a float32 square root elsewhere (there one float32 ulp of nz moves the slope by < 1.2e-6 rad);
//   * global addresses are a scalar row pointer + a constant lane offset; blocks never need a lane mask (the last
//     block of a row of blocks is shifted left to end at the region's edge and recomputes a few columns).
//   * discs clipped by the map border need no other moments (cells outside the map count as zeros): rows of the top and
//     bottom frame and the first / last block column take the x/y moments of the clipped disc from the host-built table
//     and the general tail of te_eig3.h; those block columns are cut into shorter strips so that all blocks of the one
//     round finish together.
// Cells whose disc meets an invalid cell are left NaN and their 64x16 tile is flagged for k_normals_fixup, as before.
#include "te_internal.h"
#include "te_march.h"
#include "te_n3_plan.h"
#include "te_tie_triple.h"
#include "te_eig.h"
#include "te_eig3.h"

#include <cstdlib>

namespace te {
namespace fast {

namespace {

#ifndef TE_N3_WAVES
#define TE_N3_WAVES 3
#endif
#ifndef TE_N3_DYN_LDS
#define TE_N3_DYN_LDS (TE_N3_WAVES > 3)
#endif
#ifndef TE_N3_PRIO
#define TE_N3_PRIO 0
#endif
#ifndef TE_N3_WHATIF
#define TE_N3_WHATIF 0
#endif
#ifndef TE_N3_HWHATIF  // (the sparse-hole march's measurement builds, tools/lab/r05_exp11.sh)
#define TE_N3_HWHATIF 0
#endif
constexpr int kN3Waves = TE_N3_WAVES;  // waves per SIMD the kernel is compiled for
// Sparse-hole march: the cells whose disc holds an invalid cell wait in a per-block queue (global scratch, it stays in L2)
// until 64 of them fill a wavefront for the general tail.  An item is 48 bytes: Sz, Siz, Sjz, Szz, the six x/y moments
// of the valid cells packed into three words, and (row << 8 | lane).  Up to C rows are appended between two looks at
// the queue (C <= 6, 64 cells each) on top of a remainder below 64.
constexpr int kHoleQueueItems = 512, kHoleItemBytes = 48, kHoleQueueBytes = kHoleQueueItems * kHoleItemBytes;
#ifndef TE_N3_ORDER
#define TE_N3_ORDER 0  // 1: the columns are consumed in the reverse order of their reads (one LDS wait per step)
#endif

struct N3Args {
  const float* elev;
  float* slope;
  float* rough;
  float* nx;
  float* ny;
  float* nz;
  int* tile_flags;
  int rows, cols;
  long long map_cells;
  int map;             // >= 0: this map only; < 0: blockIdx.z
  int i_lo, i_hi;      // columns [i_lo, i_hi) and rows [j_lo, j_hi) of the region, i_hi - i_lo >= 64
  int j_lo, j_hi;
  // Block columns whose lanes include cells of the left / right map frame (the first edge0 and the last edge1 block
  // columns: the last block is shifted left to end at the region's edge, so its neighbour can reach the frame too)
  // and the rows of the top / bottom frame take the general tail; the rest -- n_int block columns, rows [jf_lo, jf_hi) --
  // the closed-form tail.  Edge columns are cut into shorter strips (rows_edge < rows_int): their rows take longer.
  int nbx, edge0, edge1, n_int, s_int, s_edge, rows_int, rows_edge, n_top, jf_lo, jf_hi;
  const int* gtab;     // [(2R+1)^2][6] x/y moments {n, si, sj, sii, sij, sjj} of the disc clipped by the map border
  double res;
  double Nd, K1h, Kr2, kinv;  // N; N*res^2*sum(di^2)/2; (N*res)^2; 1/(N(N-1))
  int Ni, SIIi;               // N and sum(di^2) = sum(dj^2) of the full disc as integers (discs with invalid cells)
  int sparse_holes;           // the map holds few invalid cells (host's count at upload): HOLES = 1 instead of 2
  int no_holes;               // ... none at all: the clean march alone, on its slim ring (k_normals3s)
  int skip_clean;             // sparse holes on nearly every strip: straight to the HOLES = 1 march (Layers::skip_clean)
  int short_strips;           // dense march, invalid cells counted: strips of 32 rows, more blocks than resident slots (launch3)
  char* hole_queue;           // HOLES = 1: kHoleQueueBytes of global scratch per block (cells waiting for the general tail)
  float inv_slope_crit, inv_rough_crit;
  float band_slope, band_rough;  // a raw score within this of the clip at 0 is left to the fix-up pass (kExactNaNBits, te_internal.h)
  float Krf;                  // N*res (normals only)
  int fi0, fj0, ntx, nty, fix_groups;  // fix-up flag grid (64x16 tiles from (fi0, fj0))
  // TIE RADII (radius a whole number R of cells): the cells exactly on the circle belong to a disc or not as
  // CircleIterator::isInside decides from rounded positions, centre by centre.  The TIES march slides the shape R^2,
  // circle included (gtab: its clip table), and every row takes the rejected circle cells of its centre out of the
  // moments again before the general tail: (+-R, 0) and (0, +-R) in the kernel, the n_gen others (3-4-5 radii) from
  // gen_tab (di & 0xff | (dj & 0xff) << 8).  r2, ax, ay: what isInside needs.  n_ties = 0: an ordinary radius.
  int n_ties, n_gen;
  const int* gen_tab;
  double r2, ax, ay;
};

constexpr int n3_chunk_rows(int NR, bool slim = false) {
  const int pref[] = {4, 5, 6, 3, 7, 8, 9, 10, 11, 2};
  // (the slim ring's row counts, 2R, rarely divide by 4: chunks of 3 or 2 rows -- with 6 rows in one unrolled body the
  // scheduler holds the ring reads of several steps at once and spills: 168 registers + 1.9 KB of scratch at R = 9)
  const int pref_slim[] = {4, 3, 2, 5, 6, 7, 8, 9, 10, 11};
  for (int k = 0; k < 10; ++k) {
    const int c = slim ? pref_slim[k] : pref[k];
    if (NR % c == 0) return c;
  }
  return 1;
}

typedef float __attribute__((address_space(1))) gfloat;

// One block: columns [i0, i0 + 64), output rows [js, jend).  GENERAL: every row takes the x/y moments of its (possibly
// clipped) disc from the table and the general tail -- the blocks of the first / last block column and of the top / bottom
// frame rows; otherwise every disc of the block lies inside the map and the closed-form tail is used.
// HOLES (0 / 1 / 2): how the march deals with invalid cells.
//   0: not at all -- it stops at the first invalid cell it stages and returns the row it had reached (every row before it
//      is finished); the kernel runs the rest of the strip with HOLES = 1 or 2.  (Rounds 3-4 started the strip again:
//      with 0.01-0.03 % speckle a third of the pass, 0.28 -> 0.22-0.25 ms.)
//   1: SPARSE holes.  Every ring row carries a bit mask of its invalid cells (hm); a disc that holds such a row subtracts
//      the x/y moments of the invalid cells it contains from those of the full (or clipped) disc and takes the general
//      tail.  The work is proportional to the dirty rows in the disc and the invalid cells in them: with 0.1 % speckle
//      1.4x cheaper than (2), with 1 % 1.4x dearer, in solid unobserved regions 5x.  Which of the two a launch uses is
//      decided by the host from the fraction of invalid cells counted when the elevation was uploaded (a kernel that
//      holds all three marches and hands a strip from one to the next measured slower in EVERY case: the compiler
//      allocates registers for the union, 168 + scratch instead of 151).
//   2: DENSE holes.  Invalid cells are held in the ring as a marker value and the six x/y moments of the VALID cells are
//      slid like the z-moments while a dirty row is in the ring (about 240 integer operations per row, whatever the
//      number of holes).  Inside an unobserved region -- not one cell in the window of the ring's rows -- a step has nothing
//      to compute and only stages the next row (void_step below).  With counted REGIONS the launch cuts the map into strips
//      of 32 rows (launch3, Layers::short_strips): a strip along a region's edge costs 2.6x a clean one, and the pass lasted
//      as long as its slowest strip.
// A clean strip -- the common case by far -- thus runs code that contains nothing of the hole handling: kept in one loop behind run-time tests it cost the
// clean map 8 % (the compiler merges what the two kinds of step have in common into a maze of conditional regions).
// SLIM (clean march of a shape whose centre column alone reaches rows j +- R, i.e. hw(1) < R -- the radii just above a
// whole number of cells, the bench's 9.000009 among them): the ring holds the 2R rows j-R+1 .. j+R only.  The two cells
// of the slide that lie outside it are the centre column's: the leading one (row j+1+R) is the lane's own cell of the
// row that is staged in this very step -- taken from its prefetch register --, the trailing one (row j-R) is the
// lane's own cell of the ring row that was overwritten one step earlier -- read back just before that.  18 rows of 82
// doubles at R = 9 are 11 808 bytes: 12 single-wave blocks per CU instead of 11 (tools/census.hip: 12 288 bytes admit 12,
// 13 120 admit 11), i.e. three waves on every SIMD and strips 8 % shorter.
// Returns the first row of the strip that is NOT done: jend, or -- HOLES = 0 only -- the row at which the clean march met an
// invalid cell (every row before it is finished and stored, its tile flags written; the caller runs [that row, jend) with
// the march that handles invalid cells).
template <int Q, bool KEEP, bool GENERAL, int HOLES, bool TIES = false, bool SLIM = false>
__device__ __forceinline__ int march3(const N3Args& a, double* ring, unsigned long long (*hm)[2], const int i0, const int own_lo, const int js,
                                       const int jend) {
  constexpr int R = Shape<Q>::R;
  constexpr int W = kLanes + 2 * R;
  static_assert(!SLIM || (HOLES == 0 && !TIES && R >= 1 && Shape<Q>::hw(1) < R), "the slim ring serves the clean march of a shape whose centre column alone is 2R+1 cells high");
  constexpr int NR = SLIM ? 2 * R : 2 * R + 2;  // rows j-R .. j+1+R: exactly what one slide reads (SLIM: j-R+1 .. j+R)
  constexpr int LEAD = SLIM ? 1 : 2;             // the step at row j stages map row j + LEAD + R
  constexpr int OLD = SLIM ? R - 1 : R;          // the oldest ring row is map row j - OLD
  constexpr int C = n3_chunk_rows(NR, SLIM);
  constexpr int NC = NR / C;
  constexpr int RB = W * 8;  // bytes per ring row
  char* const ringb = reinterpret_cast<char*>(ring);
  const int lane = threadIdx.x;
  const int map = a.map >= 0 ? a.map : (int)blockIdx.z;
  const size_t mo = (size_t)map * (size_t)a.map_cells;
  const float* __restrict__ em = a.elev + mo;  // uniform

  // ---- reference height: any finite elevation of the strip's first rows (uniform) ---------------------------------
  float zref32 = 0.0f;
  {
    bool found = false;
    for (int r = js; r <= js + R && r < a.cols && !found; ++r) {
      const float t = em[(size_t)r * a.rows + i0 + lane];
      const unsigned long long msk = __ballot(__builtin_isfinite(t));
      if (msk) {
        zref32 = __shfl(t, __ffsll((long long)msk) - 1);
        found = true;
      }
    }
  }
  const double zref = (double)zref32;

  // ---- ring ----------------------------------------------------------------------------------------------------------
  // An invalid cell -- and a cell that is not there: outside the map -- is held as +0.0: it adds nothing to the z-sums.
  // Which cells of a ring row are invalid is kept beside the ring (hm, HOLES march only).
  // chunk base registers: byte address of the chunk + the lane's own column
  unsigned vb[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) vb[c] = (unsigned)(c * C * RB + lane * 8);
  // halo cell of this lane: window columns 0..R-1 and 64+R..W-1, one load covers both sides (lanes >= 2R repeat the last)
  const int hl = lane < 2 * R ? lane : 2 * R - 1;
  const int hcol = hl < R ? hl : kLanes + hl;  // window column
  const int vhd = hcol * 8 - lane * 8;         // added to a chunk base
  const int lmain = lane;            // lane index of the main cell relative to column i0
  const bool halo_in = i0 - R + hcol >= 0 && i0 - R + hcol < a.rows;  // halo columns beyond the map edge hold zeros
  const int lhalo = halo_in ? hcol - R : lane;  // ... of the halo cell (any valid address if there is none)
  // clip of my disc by the left / right map border (0: none; k > 0: columns di < -R+k missing; k < 0: di > R+k missing)
  const int icol = i0 + lane;
  // The last block of a row of blocks is shifted left to end at the region's edge; the columns it shares with its
  // neighbour belong to the neighbour (the two blocks use different reference heights, so a borderline decision --
  // "unresolved, leave it to the fix-up pass" -- can differ between them, and mixed stores would race).
  const bool own = icol >= own_lo;
  const int kx = icol < R ? R - icol : (a.rows - 1 - icol < R ? -(R - (a.rows - 1 - icol)) : 0);

  // bit k: row (oldest row of the ring + k) holds an invalid cell of the map inside this block's window ("dirty")
  unsigned dmask = 0;
  // HOLES = 2, bit k likewise: no cell of that row's window is there at all (invalid, or outside the map) -- while that holds
  // for every row of the ring, inside an unobserved region, a step has nothing to compute (the march loop below)
  unsigned amask = 0;
  bool row_void = false;
  constexpr unsigned kAllRows = NR >= 32 ? 0xffffffffu : ((1u << NR) - 1u);
  constexpr unsigned kTopBit = 1u << (NR - 1);
  constexpr unsigned kDiscMask = (1u << (2 * R + 1)) - 1u;  // the rows j-R .. j+R of the disc of row j
  bool row_dirty = false;
  float pmq[C], phq[C];          // prefetched rows (main cell, halo cell), loaded C steps before they are staged
  typedef const float __attribute__((address_space(1))) cgfloat;
  // (signed 64-bit arithmetic: the first rows of a strip at the top of the map lie above the map and are never loaded)
  cgfloat* ldp = (cgfloat*)(em + ((long long)(js - R) * a.rows + i0));  // uniform: column i0 of the next row to load
  auto load_row = [&](int r, float& pm, float& ph) __attribute__((always_inline)) {
    if ((GENERAL ? r >= 0 : true) && r < a.cols) {  // (only a GENERAL block starts above the map)
      pm = ldp[lmain];
      ph = ldp[lhalo];
    }
    ldp += a.rows;
  };
  // converts the prefetched row r and writes it to ring slot (chunk base vbase, row offset ro); rows and halo columns
  // outside the map are absent like invalid cells, but do not make a row dirty: their discs are clipped, not broken
  auto stage_row = [&](int r, unsigned vbase, int ro, float pm, float ph) __attribute__((always_inline)) {
    const bool rin = GENERAL ? (r >= 0 && r < a.cols) : true;  // (below the map: stale finite values, never part of an output)
    const bool okm = __builtin_isfinite(pm) && rin, okh = __builtin_isfinite(ph) && halo_in && rin;
    const float tm = okm ? pm : zref32, th = okh ? ph : zref32;
    typedef unsigned __attribute__((ext_vector_type(2))) u32x2;
    u32x2 bm = __builtin_bit_cast(u32x2, (double)tm - zref);  // +0.0 where absent ...
    u32x2 bh = __builtin_bit_cast(u32x2, (double)th - zref);
    if (HOLES == 2) {
      // ... turned into the marker, the smallest denormal (one select on the low word): it adds nothing to the z-sums
      // (absorbed by rounding, and 0 when squared) and can be told from every valid dz, which is a difference of two
      // float32 values (a multiple of 2^-149, or exactly 0)
      bm.x = okm ? bm.x : 1u;
      bh.x = okh ? bh.x : 1u;
    }
    *reinterpret_cast<u32x2*>(ringb + vbase + (ro * RB + R * 8)) = bm;
    *reinterpret_cast<u32x2*>(ringb + (vbase + vhd) + ro * RB) = bh;
    if (HOLES == 1) {
      // invalid cells of the map in this row, as bits over the window columns: lane k holds column R + k (main) and
      // column hcol (halo; lanes >= 2R repeat lane 2R - 1 and are ignored)
      const unsigned long long mm = __ballot(!__builtin_isfinite(pm) && rin);
      const unsigned long long mh = __ballot(!__builtin_isfinite(ph) && halo_in && rin) & ((1ull << (2 * R)) - 1ull);
      constexpr unsigned long long RM = (1ull << R) - 1ull;
      const unsigned long long lo = (mh & RM) | (mm << R);                 // columns 0 .. 63
      const unsigned long long hi = (mm >> (64 - R)) | ((mh >> R) << R);   // columns 64 .. W-1
      const int slot = (int)(__builtin_amdgcn_readfirstlane(vbase) / (unsigned)RB) + ro;  // (lane 0: chunk base)
      if (lane == 0) {
        hm[slot][0] = lo;
        hm[slot][1] = hi;
      }
      row_dirty = (mm | mh) != 0ull;
    } else {
      row_dirty = rin && __any(!__builtin_isfinite(pm) || (!__builtin_isfinite(ph) && halo_in));
      if (HOLES == 2) row_void = !__any(okm || okh);
    }
  };
  // The march starts with the disc of its first row summed directly: rows js-R .. js+R+1 go into ring rows 0 .. 2R+1
  // (the layout step j = js, u = 0 expects), C rows in flight at a time, then every lane adds up the four moments of its
  // disc column by column (unrolled: as rolled loops the sums waited for every LDS read and cost what they saved).  (Sliding in from an empty disc took 2R+1 full steps per strip -- a sixth of the kernel on the
  // 4096^2 map; the direct sums cost about four steps' worth of instructions.)
  __syncthreads();
#pragma unroll
  for (int k = 0; k < C; ++k) pmq[k] = phq[k] = 0.0f;
  // SLIM: the lane's own cell of map row r as the ring would hold it (dz, +0.0 where absent) and whether it is an invalid
  // cell of the map -- the centre column's cells that are not in the ring
  auto own_cell = [&](int r, float pm, bool& invalid) __attribute__((always_inline)) {
    const bool rin = GENERAL ? (r >= 0 && r < a.cols) : true;
    const bool okm = __builtin_isfinite(pm) && rin;
    invalid = !__builtin_isfinite(pm) && rin;
    return (double)(okm ? pm : zref32) - zref;
  };
  double ctr_old = 0.0;  // SLIM: my own cell of map row j - R (the centre column's trailing cell of the next slide)
  if constexpr (SLIM) {
    float pm0 = 0.0f, ph0 = 0.0f;
    load_row(js - R, pm0, ph0);
    bool inv = false;
    ctr_old = own_cell(js - R, pm0, inv);
    dmask |= __any(inv) ? 1u : 0u;
  }
  static_for<NC>([&](auto cc) __attribute__((always_inline)) {
    constexpr int c = decltype(cc)::value;
#pragma unroll
    for (int k = 0; k < C; ++k) load_row(js - OLD + c * C + k, pmq[k], phq[k]);
#pragma unroll
    for (int k = 0; k < C; ++k) {
      stage_row(js - OLD + c * C + k, vb[c], k, pmq[k], phq[k]);
      dmask |= row_dirty ? 1u << (c * C + k) : 0u;
      if (HOLES == 2) amask |= row_void ? 1u << (c * C + k) : 0u;
    }
  });
#pragma unroll
  for (int k = 0; k < C; ++k) load_row(js + R + LEAD + k, pmq[k], phq[k]);  // rows j + LEAD + R of the first C steps
  if (!TIES && HOLES == 0 && __builtin_expect(dmask != 0, 0)) return js;  // an invalid cell: this strip needs the other march (SLIM: the fix-up pass, see k_normals3s)

  double Sz = 0.0, Siz = 0.0, Sjz = 0.0, Szz = 0.0;
  static_for<2 * R + 1>([&](auto ec) __attribute__((always_inline)) {
    constexpr int e = decltype(ec)::value - R;  // column offset
    constexpr int h = Shape<Q>::hw(e < 0 ? -e : e);
    double cs = 0.0, cj = 0.0;
    static_for<2 * h + 1>([&](auto rc) __attribute__((always_inline)) {
      constexpr int dj = decltype(rc)::value - h;
      constexpr int p = OLD + dj;  // ring row of map row js + dj (SLIM: row js - R is not in the ring, ctr_old has the one cell of it)
      double z;
      if constexpr (p < 0)
        z = ctr_old;
      else
        z = *reinterpret_cast<const double*>(ringb + vb[(p < 0 ? 0 : p) / C] + (((p < 0 ? 0 : p) % C) * RB + (R + e) * 8));
      cs += z;
      if (dj != 0) cj = fma((double)dj, z, cj);
      Szz = fma(z, z, Szz);
    });
    Sz += cs;
    if (e != 0) Siz = fma((double)e, cs, Siz);
    Sjz += cj;
  });
  // output pointers of this block's first row (uniform base + lane), advanced by one map row per output row
  gfloat* p_slope = (gfloat*)(a.slope + mo + (size_t)js * a.rows + i0);
  gfloat* p_rough = (gfloat*)(a.rough + mo + (size_t)js * a.rows + i0);
  gfloat* p_nx = KEEP ? (gfloat*)(a.nx + mo + (size_t)js * a.rows + i0) : nullptr;
  gfloat* p_ny = KEEP ? (gfloat*)(a.ny + mo + (size_t)js * a.rows + i0) : nullptr;
  gfloat* p_nz = KEEP ? (gfloat*)(a.nz + mo + (size_t)js * a.rows + i0) : nullptr;
  const int tile_base = (a.map >= 0 ? 0 : (int)blockIdx.z) * a.ntx * a.nty;

  // Tiles to hand to the fix-up pass: one bit per 16-row tile row this strip touches, written out after the march (the
  // index arithmetic of the flag array stays out of the row loop).
  unsigned long long flag_rows = 0;
  const int tile_row0 = (js - a.fj0) >> 4;
  auto flag_tiles = [&](int j) __attribute__((always_inline)) { flag_rows |= 1ull << ((((j - a.fj0) >> 4) - tile_row0) & 63); };
  auto write_flags = [&]() __attribute__((always_inline)) {
    if (lane == 0) {
      const int tc0 = (i0 - a.fi0) >> 6, tc1 = (i0 + kLanes - 1 - a.fi0) >> 6;  // a shifted block straddles two tiles
      while (flag_rows) {
        const int k = __ffsll((long long)flag_rows) - 1;
        flag_rows &= flag_rows - 1;
        const int tr = (tile_row0 + k) * a.ntx;
        const int t0 = tile_base + tr + tc0, t1 = tile_base + tr + tc1;
        a.tile_flags[(t0 % a.fix_groups) * kFixTiles + t0 / a.fix_groups] = 1;
        a.tile_flags[(t1 % a.fix_groups) * kFixTiles + t1 / a.fix_groups] = 1;
      }
    }
  };

  // ---- tail of row j: closed-form smallest eigenpair of [[c,0,ca],[0,c,cb],[ca,cb,cd]] scaled by N^2 ---------------
  // (values only: the stores are issued at the end of the step, behind the staging of the next row, so that the wait
  // for the prefetched row never includes this row's stores)
  float o_slope, o_rough, fx = 0.0f, fy = 0.0f, fz = 0.0f;
  bool deferred = false;  // HOLES = 1: this lane's cell of the current row waits in the queue (its closed form is meaningless)
  auto tail = [&](int j) __attribute__((always_inline)) {
    bool bad, near;
    {
      const double D = fma(a.Nd, Szz, -(Sz * Sz));        // N^2 var(z)
      const double dl = fma(-0.5, D, a.K1h);              // delta = (N^2 cxx - D) / 2
      const double hs = fma(Siz, Siz, Sjz * Sjz);
      const double h2 = a.Kr2 * hs;                       // N^4 (ca^2 + cb^2)
      const double q = fma(dl, dl, h2);
      const float qf = (float)q;
      double y = (double)__builtin_amdgcn_rsqf(qf);       // 1/sqrt(q), 2^-22
      {
        const double e = fma(-(q * y), y, 1.0);
        y = fma(0.5 * y, e, y);                           // 2^-44
      }
      const double s = q * y;                             // sqrt(delta^2 + h2)
      const double t = dl + s;                            // nz ~ t
      const double X = fma(dl, s, q);                     // s * t
      double r = (double)__builtin_amdgcn_rcpf((float)X);
      {
        const double e = fma(-X, r, 1.0);
        r = fma(r, e, r);
      }
      const double m2 = h2 * r;                           // 2 (1 - nz^2) = h2 / (s t)
      // nz = sqrt(1 - m2/2): series where float32 rounding of nz matters, float32 square root elsewhere
      double pz = fma(m2, 1.0 / 128.0, 1.0 / 32.0);
      pz = fma(m2, pz, 0.25);
      const float nz_a = (float)fma(-m2, pz, 1.0);
      const float om = (float)fma(-0.5, m2, 1.0);         // nz^2
      const float nz_b = __builtin_amdgcn_sqrtf(om);
      fz = om > 0.99609375f ? nz_a : nz_b;                // m < 2^-8
      // degenerate: t <= 0 / not finite, t cancelled (delta < 0 and almost no tilt), q outside float32
      bad = !(t > 1e-6 * s) || !__builtin_isnormal(qf);
      // roughness^2 (N-1)/N = smallest eigenvalue = (cxx + cd)/2 - s (RoughnessFilter.cpp:105-117)
      const double lam = fma(0.5, D, a.K1h) - s;
      float rq = (float)(lam * a.kinv);
      rq = rq > 0.0f ? rq : 0.0f;
      const float rgh = __builtin_amdgcn_sqrtf(rq);
      const float rr = fmaf(-rgh, a.inv_rough_crit, 1.0f);
      o_rough = fmaxf(rr, 0.0f);
      near = near_clip(rr, a.band_rough);
      if (KEEP) {
        // normal ~ (N res Siz, N res Sjz, t) / sqrt(2 s t)
        const float inv = __builtin_amdgcn_rsqf((float)(2.0 * X));
        fx = (float)Siz * a.Krf * inv;
        fy = (float)Sjz * a.Krf * inv;
      }
    }
    // slope = acos(float32 nz) (SlopeFilter.cpp:74); float32 evaluation, |error| < 3e-7 rad
    const float sl = acosf_poly01(fz);
    const float rs = fmaf(-sl, a.inv_slope_crit, 1.0f);
    o_slope = fmaxf(rs, 0.0f);
    near = near || near_clip(rs, a.band_slope);
    bad = bad || near;  // (a score at its clip: the fix-up pass decides zero / not zero, TraversabilityMap.cpp:869, :897)
    if (__builtin_expect(__any(bad && own && !deferred), 0)) {
      const float qn = near ? exact_nanf() : __builtin_nanf("");
      o_slope = bad ? qn : o_slope;
      o_rough = bad ? qn : o_rough;
      fx = bad ? qn : fx;
      fy = bad ? qn : fy;
      fz = bad ? qn : fz;
      flag_tiles(j);
    }
  };
  // slope and roughness scores behind a general tail (normal fz, q = n^2 n^T C n, n cells); true: a score at its clip
  auto scores_general = [&](float nzf, double qs, int n, bool poly01) __attribute__((always_inline)) {
    const float sl = poly01 ? acosf_poly01(nzf) : acosf_poly(nzf);
    const float rs = fmaf(-sl, a.inv_slope_crit, 1.0f);
    o_slope = fmaxf(rs, 0.0f);
    float rq = (float)(qs * rcp_fast((double)n * (double)(n - 1)));
    rq = rq > 0.0f ? rq : 0.0f;
    const float rgh = __builtin_amdgcn_sqrtf(rq);
    const float rr = fmaf(-rgh, a.inv_rough_crit, 1.0f);
    o_rough = n > 1 ? fmaxf(rr, 0.0f) : 0.0f;  // n == 1: 0/0 -> "roughness < crit" false -> 0
    return near_clip(rs, a.band_slope) || (n > 1 && near_clip(rr, a.band_rough));
  };
  // rows of the top / bottom frame and block columns with lanes in the left / right frame: x/y moments of the clipped
  // disc from the host-built table, general tail (te_eig3.h); the z-moments are already right (cells outside count 0)
  auto tail_clipped = [&](int j, int ky) __attribute__((always_inline)) {
    const int* gt = a.gtab + ((ky + R) * (2 * R + 1) + (kx + R)) * 6;
    const int n = gt[0];
    double qs = 0.0;
    const int unresolved = general_tail3(a.res, n, gt[1], gt[2], gt[3], gt[4], gt[5], Sz, Siz, Sjz, Szz, fx, fy, fz, qs);
    const bool near = scores_general(fz, qs, n, false);
    if (__builtin_expect(__any((unresolved != 0 || near) && own), 0)) {
      const float qn = near ? exact_nanf() : __builtin_nanf("");
      const bool bad = unresolved != 0 || near;
      o_slope = bad ? qn : o_slope;
      o_rough = bad ? qn : o_rough;
      fx = bad ? qn : fx;
      fy = bad ? qn : fy;
      fz = bad ? qn : fz;
      flag_tiles(j);
    }
  };
  // ---- tie radii (TIES march, clean strips): the moments of the shape with its circle minus the circle cells that
  // isInside() rejects for this centre, general tail.  A strip with an invalid cell is left to the fix-up pass whole.
  // (+-R, 0): dy = 0, the decision is the same for every row -- taken once per lane, with its share of the x/y moments
  bool xf_p = false, xf_m = false;
  int xn = 0, xsi = 0, xsii = 0;
  int n0 = 0, si0 = 0, sj0 = 0, sii0 = 0, sij0 = 0, sjj0 = 0;  // interior blocks: the moments of the unclipped shape (one table entry)
  if constexpr (TIES) {
    const double xi = a.ax + a.res * (double)(-icol);  // cell_x (te_geom.h)
    const double dxp = (a.ax + a.res * (double)(-(icol + R))) - xi, dxm = (a.ax + a.res * (double)(-(icol - R))) - xi;
    xf_p = !(dxp * dxp + 0.0 <= a.r2) && icol + R < a.rows;
    xf_m = !(dxm * dxm + 0.0 <= a.r2) && icol - R >= 0;
    xn = (xf_p ? 1 : 0) + (xf_m ? 1 : 0);
    xsi = (xf_p ? R : 0) - (xf_m ? R : 0);
    xsii = R * R * xn;
    if constexpr (!GENERAL) {
      const int* gt = a.gtab + (R * (2 * R + 1) + R) * 6;
      n0 = gt[0]; si0 = gt[1]; sj0 = gt[2]; sii0 = gt[3]; sij0 = gt[4]; sjj0 = gt[5];
    }
  }
  // the circle's other cells (te_tie_triple.h: the eight of a Pythagorean triple, radii of 5 and 10 cells here): dx * dx of
  // isInside()'s test per lane, -inf for a cell outside the map (it passes the test: never taken out)
  constexpr int TA = TIES ? tie_triple_a(R) : 0, TB = TIES ? tie_triple_b(R) : 0;
  static_assert(!TIES || tie_triples(R) <= 1, "one Pythagorean triple per radius");
  double dxsq[4] = {0.0, 0.0, 0.0, 0.0};  // di = -TB, -TA, +TA, +TB
  if constexpr (TA != 0) {
    const double xi = a.ax + a.res * (double)(-icol);
    constexpr int d4[4] = {-TB, -TA, TA, TB};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double dx = (a.ax + a.res * (double)(-(icol + d4[q]))) - xi;
      dxsq[q] = (unsigned)(icol + d4[q]) < (unsigned)a.rows ? dx * dx : -__builtin_inf();
    }
  }
  auto tail_ties = [&](int j, int ky, auto uc) __attribute__((always_inline)) {
    constexpr int u = decltype(uc)::value;
    constexpr int pc = u + R;  // ring position of map row j
    // my window's cell (di, row at ring position p): chunk base register + immediate, like the slide's reads
    auto cell = [&](auto pst, auto dst) __attribute__((always_inline)) {
      constexpr int p = decltype(pst)::value, di = decltype(dst)::value;
      return *reinterpret_cast<const double*>(ringb + vb[(p / C) % NC] + ((p % C) * RB + (R + di) * 8));
    };
    typedef std::integral_constant<int, R> RP;
    typedef std::integral_constant<int, -R> RM;
    typedef std::integral_constant<int, 0> Z0;
    int n, si, sj, sii, sij, sjj;
    if constexpr (GENERAL) {
      const int* gt = a.gtab + ((ky + R) * (2 * R + 1) + (kx + R)) * 6;
      n = gt[0]; si = gt[1]; sj = gt[2]; sii = gt[3]; sij = gt[4]; sjj = gt[5];
    } else {
      n = n0; si = si0; sj = sj0; sii = sii0; sij = sij0; sjj = sjj0;
    }
    n -= xn;
    si -= xsi;
    sii -= xsii;
    const double zp = xf_p ? cell(std::integral_constant<int, pc>{}, RP{}) : 0.0;
    const double zm = xf_m ? cell(std::integral_constant<int, pc>{}, RM{}) : 0.0;
    double lSz = (Sz - zp) - zm;
    double lSiz = fma((double)R, zm, fma(-(double)R, zp, Siz));
    double lSjz = Sjz;
    double lSzz = fma(-zm, zm, fma(-zp, zp, Szz));
    // (0, +-R): dx = 0, the same answer for every lane (uniform branches)
    const double yj = a.ay + a.res * (double)(-j);  // cell_y
    {
      const int jj = j - R;
      if (jj >= 0) {
        const double dy = (a.ay + a.res * (double)(-jj)) - yj;
        if (!(0.0 + dy * dy <= a.r2)) {
          const double z = cell(std::integral_constant<int, pc - R>{}, Z0{});
          n -= 1;
          sj += R;
          sjj -= R * R;
          lSz -= z;
          lSjz = fma((double)R, z, lSjz);
          lSzz = fma(-z, z, lSzz);
        }
      }
    }
    {
      const int jj = j + R;
      if (jj < a.cols) {
        const double dy = (a.ay + a.res * (double)(-jj)) - yj;
        if (!(0.0 + dy * dy <= a.r2)) {
          const double z = cell(std::integral_constant<int, pc + R>{}, Z0{});
          n -= 1;
          sj -= R;
          sjj -= R * R;
          lSz -= z;
          lSjz = fma(-(double)R, z, lSjz);
          lSzz = fma(-z, z, lSzz);
        }
      }
    }
    if constexpr (TA != 0) {  // the other circle cells: both coordinates decide -- dx * dx from the lane, dy * dy from the row
      constexpr int d4[4] = {-TB, -TA, TA, TB};
      double dysq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int jj = j + d4[q];
        const double dy = (a.ay + a.res * (double)(-jj)) - yj;
        dysq[q] = (unsigned)jj < (unsigned)a.cols ? dy * dy : -__builtin_inf();
      }
      static_for<4>([&](auto qic) __attribute__((always_inline)) {
        constexpr int qi = decltype(qic)::value;
        constexpr int di = d4[qi];
        static_for<2>([&](auto sc) __attribute__((always_inline)) {
          constexpr int qj = (qi == 0 || qi == 3) ? (decltype(sc)::value == 0 ? 1 : 2) : (decltype(sc)::value == 0 ? 0 : 3);  // |di| = TB: dj = -+TA; |di| = TA: dj = -+TB
          constexpr int dj = d4[qj];
          const bool fail = !(dxsq[qi] + dysq[qj] <= a.r2);
          const double zc = cell(std::integral_constant<int, pc + dj>{}, std::integral_constant<int, di>{});
          const double zz = fail ? zc : 0.0;
          n -= fail ? 1 : 0;
          si -= fail ? di : 0;
          sj -= fail ? dj : 0;
          sii -= fail ? di * di : 0;
          sij -= fail ? di * dj : 0;
          sjj -= fail ? dj * dj : 0;
          lSz -= zz;
          lSiz = fma(-(double)di, zz, lSiz);
          lSjz = fma(-(double)dj, zz, lSjz);
          lSzz = fma(-zz, zz, lSzz);
        });
      });
    }
    double qs = 0.0;
    const int unresolved = general_tail3(a.res, n, si, sj, sii, sij, sjj, lSz, lSiz, lSjz, lSzz, fx, fy, fz, qs);
    const bool near = scores_general(fz, qs, n, false);
    if (__builtin_expect(__any((unresolved != 0 || near) && own), 0)) {
      const float qn = near ? exact_nanf() : __builtin_nanf("");
      const bool bad = unresolved != 0 || near;
      o_slope = bad ? qn : o_slope;
      o_rough = bad ? qn : o_rough;
      fx = bad ? qn : fx;
      fy = bad ? qn : fy;
      fz = bad ? qn : fz;
      flag_tiles(j);
    }
  };
  auto leave_to_fixup = [&](int j_from) __attribute__((always_inline)) {  // rows [j_from, jend) of this block: every tile they touch
    // (the fix-up pass takes the valid cells of a flagged tile whose slope is NaN: a region run finds old values there)
    // (... and an invalid centre is nobody's business there: NaN in every layer, as the march itself leaves it)
    const size_t o0 = mo + (size_t)j_from * a.rows + i0 + lane;
    for (int jj = j_from; jj < jend; ++jj) {
      const size_t o = o0 + (size_t)(jj - j_from) * a.rows;
      if (own) {
        a.slope[o] = a.rough[o] = __builtin_nanf("");
        if (KEEP) a.nx[o] = a.ny[o] = a.nz[o] = __builtin_nanf("");
      }
    }
    for (int jj = j_from; jj < jend; jj += 16) flag_tiles(jj);
    if (j_from < jend) flag_tiles(jend - 1);
  };
  if constexpr (TIES) {
    if (__builtin_expect(dmask != 0, 0)) {  // an invalid cell among the first rows
      leave_to_fixup(js);
      write_flags();
      return jend;
    }
  }
  // ---- rows whose disc holds invalid cells ------------------------------------------------------------------------------
  // x/y moments of the VALID cells = those of the full (GENERAL: clipped) disc minus those of the invalid cells in it.
  // The invalid cells come from the bit masks of the dirty ring rows of the disc (uniform loop over those rows, one
  // broadcast LDS read each; a lane shifts its run of the row out of the mask and walks the set bits -- with sparse
  // holes there is one, rarely two).  The z-moments are already right: an invalid cell is +0.0 in the ring.
  unsigned qhead = 0, qtail = 0;  // uniform; items qhead .. qtail-1 (mod kHoleQueueItems) wait
  char* const qb = (HOLES == 1 && !GENERAL && !KEEP)
                       ? a.hole_queue + ((size_t)blockIdx.x + (size_t)gridDim.x * blockIdx.z) * (size_t)kHoleQueueBytes
                       : nullptr;
  // the general tail for up to 64 waiting cells, one per lane; their slope / roughness go straight to the layers
  auto flush_queue = [&](unsigned count) {
    // the items were written by other lanes of THIS wave: workgroup scope -- wait for the stores, the CU's L1 is coherent
    // for its own waves.  (Agent scope writes back and invalidates the XCD's L2 on this chip: the launch took 2.8 ms.)
#if TE_N3_HWHATIF == 3  // (measurement only: the queue is filled and never read)
    qhead += count;
    return;
#endif
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    const bool act = (unsigned)lane < count;
    const char* it = qb + (size_t)((qhead + (unsigned)lane) & (kHoleQueueItems - 1)) * kHoleItemBytes;
    typedef double __attribute__((ext_vector_type(2))) d2;
    typedef unsigned __attribute__((ext_vector_type(4))) u4;
    d2 m0 = d2{0.0, 0.0}, m1 = d2{0.0, 0.0};
    u4 w = u4{3u, 0u, 0u, 0u};
    if (act) {
      m0 = *reinterpret_cast<const d2*>(it);
      m1 = *reinterpret_cast<const d2*>(it + 16);
      w = *reinterpret_cast<const u4*>(it + 32);
    }
    const int qn_ = (int)(w.x & 0xffffu), qsi = (int)(short)(w.x >> 16), qsj = (int)(short)(w.y & 0xffffu), qsii = (int)(w.y >> 16),
              qsij = (int)(short)(w.z & 0xffffu), qsjj = (int)(w.z >> 16);
    const int jj = (int)(w.w >> 8), ll = (int)(w.w & 0xffu);
    float gx, gy, gz;
    double qs = 0.0;
    const int unresolved = general_tail3(a.res, qn_, qsi, qsj, qsii, qsij, qsjj, m0.x, m0.y, m1.x, m1.y, gx, gy, gz, qs);
    const float sl = acosf_poly01(gz);
    const float rs_ = fmaf(-sl, a.inv_slope_crit, 1.0f);
    float s_out = fmaxf(rs_, 0.0f);
    float rq = (float)(qs * rcp_fast((double)qn_ * (double)(qn_ - 1)));
    rq = rq > 0.0f ? rq : 0.0f;
    const float rgh = __builtin_amdgcn_sqrtf(rq);
    const float rr_ = fmaf(-rgh, a.inv_rough_crit, 1.0f);
    float r_out = qn_ > 1 ? fmaxf(rr_, 0.0f) : 0.0f;  // n == 1: 0/0 -> "roughness < crit" false -> 0
    const bool near = near_clip(rs_, a.band_slope) || (qn_ > 1 && near_clip(rr_, a.band_rough));  // a score at its clip: the fix-up pass decides
    const bool bad = act && (unresolved != 0 || near);
    if (bad) s_out = r_out = near ? exact_nanf() : __builtin_nanf("");
    unsigned long long bm = __ballot(bad);
    while (__builtin_expect(bm != 0ull, 0)) {  // unresolved cells go to the fix-up pass (rare)
      const int l = __builtin_ctzll(bm);
      bm &= bm - 1ull;
      flag_tiles(__builtin_amdgcn_readlane(jj, l));
    }
    if (act) {
      const size_t o = mo + (size_t)jj * a.rows + (size_t)(i0 + ll);
      a.slope[o] = s_out;
      a.rough[o] = r_out;
    }
    qhead += count;
  };
  auto tail_sparse = [&](int j, auto uc) __attribute__((always_inline)) {
    constexpr int u = decltype(uc)::value;
    const int slot0 = (int)(__builtin_amdgcn_readfirstlane(vb[0]) / (unsigned)RB) + u;  // ring slot of map row j - R
    int hn = 0, hi_ = 0, hj = 0, hii = 0, hij = 0, hjj = 0;
    bool nocentre = false;
    unsigned bits = dmask & kDiscMask;  // bit k: map row j - R + k is dirty (uniform)
#if TE_N3_HWHATIF == 1  // (measurement only: the invalid cells are not looked at)
    bits = 0;
#endif
    // (Taking the invalid cells one at a time for all lanes -- scalar loops over the uniform row masks, a lane only tests
    // di^2 <= Q - dj^2 -- measured slower: the pass 0.60 ms against 0.32 at 0.1 % speckle, tools/lab/r05_exp12.sh.  That
    // build smeared bit 31 of a mask over its upper word -- phantom cells in one dirty row of 80 --, which does not explain
    // a factor; it was not measured again.)
    while (bits) {
      const int k = __builtin_ctz(bits);
      bits &= bits - 1u;
      int sl = slot0 + k;
      sl = sl >= NR ? sl - NR : sl;
      sl = sl >= NR ? sl - NR : sl;
      const unsigned long long lo = hm[sl][0], hi = hm[sl][1];
      const int dj = k - R, adj = dj < 0 ? -dj : dj;
      const int w = (int)__builtin_sqrtf((float)(Q - adj * adj));  // half-width of the disc's run in this row (exact: small integers)
      const int sh = lane + R - w;                                  // window column of the run's first cell, 0 .. 63 + R
      const unsigned long long x = sh < 64 ? ((lo >> sh) | (sh ? hi << (64 - sh) : 0ull)) : (hi >> (sh - 64));
      unsigned b = (unsigned)x & ((2u << (2 * w)) - 1u);            // bit t: cell di = t - w of the run is invalid
      if (dj == 0) nocentre = ((b >> w) & 1u) != 0;
      while (b) {
        const int di = __builtin_ctz(b) - w;
        b &= b - 1u;
        hn += 1;
        hi_ += di;
        hj += dj;
        hii += di * di;
        hij += di * dj;
        hjj += dj * dj;
      }
    }
    if constexpr (!GENERAL && !KEEP && TE_N3_HWHATIF != 4) {  // (4: the general tail in place for every row with a dirty row in its disc)
      // Interior blocks: the lanes whose disc holds no invalid cell take the closed form like on a clean row; the
      // others (22 % of the cells at 0.1 % speckle, while 79 % of the ROWS hold at least one) are put into the queue
      // with their moments and get the general tail later, 64 at a time (flush_queue).
#if TE_N3_HWHATIF == 2  // (measurement only: the walk, but nobody waits for a general tail)
      hn = 0;
#endif
      deferred = hn > 0;  // (for tail(): neither these lanes nor the invalid centres below are its business)
      tail(j);
      deferred = hn > 0 && !nocentre && own;
      if (nocentre) o_slope = o_rough = __builtin_nanf("");  // no slope or roughness where the input layer is invalid
      const unsigned long long dm = __ballot(deferred);
      if (dm != 0ull) {
        const unsigned pos = (qtail + __builtin_amdgcn_mbcnt_hi((unsigned)(dm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)dm, 0u))) & (kHoleQueueItems - 1);
        if (deferred) {
          typedef double __attribute__((ext_vector_type(2))) d2;
          typedef unsigned __attribute__((ext_vector_type(4))) u4;
          char* it = qb + (size_t)pos * kHoleItemBytes;
          *reinterpret_cast<d2*>(it) = d2{Sz, Siz};
          *reinterpret_cast<d2*>(it + 16) = d2{Sjz, Szz};
          const unsigned w0 = (unsigned)(a.Ni - hn) | ((unsigned)(-hi_) << 16);
          const unsigned w1 = ((unsigned)(-hj) & 0xffffu) | ((unsigned)(a.SIIi - hii) << 16);
          const unsigned w2 = ((unsigned)(-hij) & 0xffffu) | ((unsigned)(a.SIIi - hjj) << 16);
          *reinterpret_cast<u4*>(it + 32) = u4{w0, w1, w2, ((unsigned)j << 8) | (unsigned)lane};
        }
        qtail += (unsigned)__popcll(dm);
      }
      return;
    }
    int n0 = a.Ni, si0 = 0, sj0 = 0, sii0 = a.SIIi, sij0 = 0, sjj0 = a.SIIi;
    if (GENERAL) {
      const int ky = j < R ? R - j : (a.cols - 1 - j < R ? -(R - (a.cols - 1 - j)) : 0);  // uniform
      const int* gt = a.gtab + ((ky + R) * (2 * R + 1) + (kx + R)) * 6;
      n0 = gt[0]; si0 = gt[1]; sj0 = gt[2]; sii0 = gt[3]; sij0 = gt[4]; sjj0 = gt[5];
    }
    const int Mn = n0 - hn;
    double qs = 0.0;
    const int unresolved = general_tail3(a.res, Mn, si0 - hi_, sj0 - hj, sii0 - hii, sij0 - hij, sjj0 - hjj, Sz, Siz, Sjz, Szz, fx, fy, fz, qs);
    const bool near = scores_general(fz, qs, Mn, true);
    const bool bad = (unresolved != 0 || near) && !nocentre;  // (no normal, slope or roughness where the input layer is invalid)
    const float qn = (near && !nocentre) ? exact_nanf() : __builtin_nanf("");
    if (nocentre || bad) {
      o_slope = qn;
      o_rough = qn;
      fx = qn;
      fy = qn;
      fz = qn;
    }
    if (__builtin_expect(__any(bad && own), 0)) flag_tiles(j);
  };
  // ---- HOLES == 2: x/y moments of the VALID cells, slid like the z-moments ----------------------------------------------
  // They are kept only while a dirty row is in the ring ("holes" mode).  On entry they are counted once from the ring
  // itself (absent cells carry the marker, whatever the reason: invalid, outside the map, above the strip), so no case
  // analysis of borders and warm-up is needed; a disc without a dirty row never looks at them.
  int Mn = 0, Mi = 0, Mj = 0, Mii = 0, Mij = 0, Mjj = 0;
  bool holes = false;
  auto absent = [&](double v) __attribute__((always_inline)) { return __builtin_bit_cast(unsigned long long, v) == 1ull; };
  auto count_moments = [&](int u) __attribute__((always_inline)) {
    // disc of row j: ring rows u .. u+2R counted from the row vb[0] points to
    const int slot0 = (int)(__builtin_amdgcn_readfirstlane(vb[0]) / RB) + u;
    int n = 0, si = 0, sj = 0, sii = 0, sij = 0, sjj = 0;
#pragma unroll 1
    for (int dj = -R; dj <= R; ++dj) {
      const int hw = isqrt_c(Q - dj * dj);
      int sl = slot0 + R + dj;
      sl = sl >= NR ? sl - NR : sl;
      sl = sl >= NR ? sl - NR : sl;
      const double* row = ring + sl * W + lane + R;
#pragma unroll 1
      for (int di = -hw; di <= hw; ++di) {
        const int w = absent(row[di]) ? 0 : 1;
        n += w;
        si += w * di;
        sii += w * di * di;
        sj += w * dj;
        sij += w * di * dj;
        sjj += w * dj * dj;
      }
    }
    Mn = n; Mi = si; Mj = sj; Mii = sii; Mij = sij; Mjj = sjj;
  };
  auto tail_dense = [&](int j, auto uc) __attribute__((always_inline)) {
    constexpr int u = decltype(uc)::value;
    constexpr int pc = u + R;  // ring position of row j
    const double ctr = *reinterpret_cast<const double*>(ringb + vb[(pc / C) % NC] + ((pc % C) * RB + R * 8));
    double qs = 0.0;
    const int unresolved = general_tail3(a.res, Mn, Mi, Mj, Mii, Mij, Mjj, Sz, Siz, Sjz, Szz, fx, fy, fz, qs);
    const bool near = scores_general(fz, qs, Mn, true);
    const bool nocentre = absent(ctr);  // no normal, slope or roughness where the input layer is invalid
    const bool bad = (unresolved != 0 || near) && !nocentre;
    const float qn = (near && !nocentre) ? exact_nanf() : __builtin_nanf("");
    if (nocentre || bad) {
      o_slope = qn;
      o_rough = qn;
      fx = qn;
      fy = qn;
      fz = qn;
    }
    if (__builtin_expect(__any(bad && own), 0)) flag_tiles(j);
  };
  auto store_row = [&]() __attribute__((always_inline)) {
    if (own && !deferred) {
      p_slope[lane] = o_slope;
      p_rough[lane] = o_rough;
    }
    p_slope += a.rows;
    p_rough += a.rows;
    if (KEEP) {
      if (own) {
        p_nx[lane] = fx;
        p_ny[lane] = fy;
        p_nz[lane] = fz;
      }
      p_nx += a.rows;
      p_ny += a.rows;
      p_nz += a.rows;
    }
  };

  // ---- slide from row j to row j+1 at unrolled position u (the oldest row j-R is row u of chunk role 0) -------------
  auto slide = [&](auto uc, double lead_c = 0.0) __attribute__((always_inline)) {  // lead_c (SLIM): my own cell of row j + 1 + R
    constexpr int u = decltype(uc)::value;
    double sv[R + 1];  // sum of (lead + trail) over the columns of half-height h
    const double Sz0 = Sz;
    // all ring reads of the step first ...
    double zl[2 * R + 1], zt[2 * R + 1];  // indexed by column offset e + R
    static_for<R + 1>([&](auto dc) __attribute__((always_inline)) {
      constexpr int d = decltype(dc)::value;
      constexpr int h = Shape<Q>::hw(d);
      if constexpr (SLIM && d == 0) {  // the centre column: neither cell is in the ring
        zl[R] = lead_c;
        zt[R] = ctr_old;
      } else {
        constexpr int pl = u + OLD + 1 + h, pt = u + OLD - h;  // ring positions of the leading row j+1+h and the trailing row j-h
        constexpr int al = (pl / C) % NC, ol = pl % C, at = (pt / C) % NC, ot = pt % C;
        const char* rl = ringb + vb[al];
        const char* rt = ringb + vb[at];
        zl[R + d] = *reinterpret_cast<const double*>(rl + (ol * RB + (R + d) * 8));
        zt[R + d] = *reinterpret_cast<const double*>(rt + (ot * RB + (R + d) * 8));
        if (d != 0) {
          zl[R - d] = *reinterpret_cast<const double*>(rl + (ol * RB + (R - d) * 8));
          zt[R - d] = *reinterpret_cast<const double*>(rt + (ot * RB + (R - d) * 8));
        }
      }
    });
    // ... then the moment updates, column by column
#if TE_N3_ORDER == 1
    // (last read first: one wait for the whole step instead of one per column pair)
    static_for<R + 1>([&](auto dcr) __attribute__((always_inline)) {
      constexpr int d = R - decltype(dcr)::value;
#else
    static_for<R + 1>([&](auto dcr) __attribute__((always_inline)) {
      constexpr int d = decltype(dcr)::value;
#endif
      constexpr int h = Shape<Q>::hw(d);
      // first column of its height group in the order the columns are visited
#if TE_N3_ORDER == 1
      constexpr bool first = d == R || Shape<Q>::hw(d < R ? d + 1 : R) != h;
#else
      constexpr bool first = d == 0 || Shape<Q>::hw(d > 0 ? d - 1 : 0) != h;
#endif
      auto column = [&](auto ec, bool init) __attribute__((always_inline)) {
        constexpr int e = decltype(ec)::value - R;  // column offset
        const double uu = zl[R + e] - zt[R + e], vv = zl[R + e] + zt[R + e];
        Sz += uu;
        if (e != 0) Siz = fma((double)e, uu, Siz);
        Szz = fma(uu, vv, Szz);
        if (init)
          sv[h] = vv;
        else
          sv[h] += vv;
      };
#if TE_N3_ORDER == 1
      if (d != 0) column(std::integral_constant<int, R - d>{}, first);
      column(std::integral_constant<int, R + d>{}, first && d == 0);
#else
      column(std::integral_constant<int, R + d>{}, first);
      if (d != 0) column(std::integral_constant<int, R - d>{}, false);
#endif
    });
    double acc = Sjz;
    static_for<R + 1>([&](auto dc) __attribute__((always_inline)) {
      constexpr int d = decltype(dc)::value;
      constexpr int h = Shape<Q>::hw(d);
      constexpr bool first = d == 0 || Shape<Q>::hw(d > 0 ? d - 1 : 0) != h;
      if (first) acc = fma((double)h + 0.5, sv[h], acc);
    });
    Sjz = fma(-0.5, Sz0 + Sz, acc);
  };

  // the slide of a step in holes mode: the z-moments as above (absent cells add nothing) and the six x/y moments of the
  // valid cells.  With w = 1 for a valid cell, column e of half-height h, leading cell wl (row j+1+h), trailing wt (j-h):
  //   n'  = n  + sum (wl - wt)             i'  = i  + sum e (wl - wt)        ii' = ii + sum e^2 (wl - wt)
  //   j'  = j  - n  + sum [h wl + (h+1) wt]
  //   ij' = ij - i  + sum e [h wl + (h+1) wt]
  //   jj' = jj - 2j + n + sum [h^2 wl - (h+1)^2 wt]          (n, i, j on the right: before the step)
  auto slide_holes = [&](auto uc) __attribute__((always_inline)) {
    constexpr int u = decltype(uc)::value;
    double sv[R + 1];
    int wl_g[R + 1], wt_g[R + 1], el_g[R + 1], et_g[R + 1];  // per half-height: sum wl, sum wt, sum e wl, sum e wt
    int q_l = 0, q_t = 0;                                     // sum e^2 wl, sum e^2 wt
    const double Sz0 = Sz;
    static_for<R + 1>([&](auto dc) __attribute__((always_inline)) {
      constexpr int d = decltype(dc)::value;
      constexpr int h = Shape<Q>::hw(d);
      constexpr int pl = u + R + 1 + h, pt = u + R - h;
      constexpr int al = (pl / C) % NC, ol = pl % C, at = (pt / C) % NC, ot = pt % C;
      constexpr bool first = d == 0 || Shape<Q>::hw(d > 0 ? d - 1 : 0) != h;
      const char* rl = ringb + vb[al];
      const char* rt = ringb + vb[at];
      auto column = [&](auto ec, bool init) __attribute__((always_inline)) {
        constexpr int e = decltype(ec)::value - R;
        const double zl = *reinterpret_cast<const double*>(rl + (ol * RB + (R + e) * 8));
        const double zt = *reinterpret_cast<const double*>(rt + (ot * RB + (R + e) * 8));
        const double uu = zl - zt, vv = zl + zt;
        Sz += uu;
        if (e != 0) Siz = fma((double)e, uu, Siz);
        Szz = fma(uu, vv, Szz);
        const int wl = absent(zl) ? 0 : 1, wt = absent(zt) ? 0 : 1;
        if (init) {
          sv[h] = vv;
          wl_g[h] = wl;
          wt_g[h] = wt;
          el_g[h] = e * wl;
          et_g[h] = e * wt;
        } else {
          sv[h] += vv;
          wl_g[h] += wl;
          wt_g[h] += wt;
          el_g[h] += e * wl;
          et_g[h] += e * wt;
        }
        q_l += e * e * wl;
        q_t += e * e * wt;
      };
      column(std::integral_constant<int, R + d>{}, first);
      if (d != 0) column(std::integral_constant<int, R - d>{}, false);
    });
    double acc = Sjz;
    int dn = 0, di = 0, dj = 0, dij = 0, djj = 0;
    static_for<R + 1>([&](auto dc) __attribute__((always_inline)) {
      constexpr int d = decltype(dc)::value;
      constexpr int h = Shape<Q>::hw(d);
      constexpr bool first = d == 0 || Shape<Q>::hw(d > 0 ? d - 1 : 0) != h;
      if (first) {
        acc = fma((double)h + 0.5, sv[h], acc);
        dn += wl_g[h] - wt_g[h];
        di += el_g[h] - et_g[h];
        dj += h * wl_g[h] + (h + 1) * wt_g[h];
        dij += h * el_g[h] + (h + 1) * et_g[h];
        djj += h * h * wl_g[h] - (h + 1) * (h + 1) * wt_g[h];
      }
    });
    Sjz = fma(-0.5, Sz0 + Sz, acc);
    Mjj += djj - 2 * Mj + Mn;
    Mij += dij - Mi;
    Mj += dj - Mn;
    Mn += dn;
    Mi += di;
    Mii += q_l - q_t;
  };

  // ---- the march ---------------------------------------------------------------------------------------------------
  int j = js;
  auto rotate = [&]() __attribute__((always_inline)) {
    if (NC > 1) {  // the chunk that held the oldest rows now holds the newest
      const unsigned v0 = vb[0];
#pragma unroll
      for (int c = 0; c + 1 < NC; ++c) vb[c] = vb[c + 1];
      vb[NC - 1] = v0;
    }
  };
  if constexpr (HOLES == 0) {
    bool aborted = false;
#pragma unroll 1
    while (true) {
      bool leave = false;
      static_for<C>([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        if (leave) return;
        if (__builtin_expect(j >= jend, 0)) {
          leave = true;
          return;
        }
        constexpr bool out = true;
        if (out) {
          if constexpr (TIES) {
            const int ky = GENERAL ? (j < R ? R - j : (a.cols - 1 - j < R ? -(R - (a.cols - 1 - j)) : 0)) : 0;  // uniform
            tail_ties(j, ky, uc);
          } else if (GENERAL) {
            const int ky = j < R ? R - j : (a.cols - 1 - j < R ? -(R - (a.cols - 1 - j)) : 0);  // uniform
            tail_clipped(j, ky);
          } else {
#if TE_N3_WHATIF == 1  // (measurement only: no tail -- what the slide alone costs; the layers hold garbage)
            o_slope = (float)(Sz + Siz);
            o_rough = (float)(Sjz + Szz);
#else
            tail(j);
#endif
          }
        }
#if TE_N3_WHATIF == 2  // (measurement only: no slide -- the tail, the staging and the stores alone)
        Sz += 1.0;
#else
        if constexpr (SLIM) {
          bool inv = false;
          const double lead_c = own_cell(j + 1 + R, pmq[u], inv);
          slide(uc, lead_c);
          // my own cell of the ring's oldest row, j - R + 1, before row j + 1 + R takes its slot: the next slide's trailing centre cell
          ctr_old = *reinterpret_cast<const double*>(ringb + vb[0] + (u * RB + R * 8));
        } else {
          slide(uc);
        }
#endif
        // row j+2+R replaces row j-R (same slot: LDS operations of a wave execute in order); SLIM: row j+1+R replaces row j-R+1
        stage_row(j + LEAD + R, vb[0], u, pmq[u], phq[u]);
        load_row(j + LEAD + R + C, pmq[u], phq[u]);
        if (out) store_row();
        ++j;
        if (__builtin_expect(row_dirty && j < jend, 0)) {  // an invalid cell: this strip needs the other march
          aborted = true;
          leave = true;
        }
      });
      if (leave) break;
      rotate();
    }
    if (aborted) {
      if constexpr (TIES) {  // the rest of the strip belongs to the fix-up pass
        leave_to_fixup(j);
        write_flags();
        return jend;
      }
      // rows [js, j) are done: the invalid cell is in the map row that the step of row j - 1 staged, j + R (SLIM) or
      // j + 1 + R -- beyond the disc of row j - 1
      if (__builtin_expect(flag_rows != 0, 0)) write_flags();
      return j;
    }
  } else {
    bool done = false;
#pragma unroll 1
    while (!done) {
      static_for<C>([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        if (done) return;
        if (__builtin_expect(j >= jend, 0)) {
          done = true;
          return;
        }
        if (HOLES == 2 && dmask != 0 && !holes) {  // a dirty row has entered the ring (it leads in this step's slide)
          count_moments(u);
          holes = true;
        }
        deferred = false;
        // Inside an unobserved region: not one cell in the window of the ring's rows, j - R .. j + 1 + R.  Every disc of this
        // row is empty -- no normal, slope or roughness (the centre is invalid) -- and the slide would add and take away
        // nothing: the moments (all zero, the z-sums a few markers) are those of the next row's discs as they stand.
        // (One branch around the tail and one around the slide: a step of its own with the staging and the stores a second
        // time cost the speckled maps 5 % -- the kernel's code is larger than the instruction cache as it is.)
        const bool void_step = HOLES == 2 && holes && amask == kAllRows;  // (uniform)
        if (__builtin_expect(void_step, 0)) {
          o_slope = o_rough = fx = fy = fz = __builtin_nanf("");
        } else if ((dmask & kDiscMask) == 0) {
          if (GENERAL) {
            const int ky = j < R ? R - j : (a.cols - 1 - j < R ? -(R - (a.cols - 1 - j)) : 0);  // uniform
            tail_clipped(j, ky);
          } else {
            tail(j);
          }
        } else if constexpr (HOLES == 1) {
          tail_sparse(j, uc);
        } else {
          tail_dense(j, uc);
        }
        if (__builtin_expect(void_step, 0)) {
        } else if (HOLES == 2 && holes) {
          slide_holes(uc);
        } else {
          slide(uc);
        }
        stage_row(j + 2 + R, vb[0], u, pmq[u], phq[u]);
        dmask = (dmask >> 1) | (row_dirty ? kTopBit : 0u);
        if (HOLES == 2) amask = (amask >> 1) | (row_void ? kTopBit : 0u);
        if (HOLES == 2) holes = holes && dmask != 0;  // the last dirty row has left the ring: the table / closed form serves again
        load_row(j + 2 + R + C, pmq[u], phq[u]);
        store_row();
        ++j;
      });
      // (looking at the queue inside the last step of the chunk, before that step's row prefetch is issued -- so that the
      // flush's fence does not wait for a load just sent -- changed nothing: 0.3229 / 0.3257 ms, tools/lab/r05_exp17.sh)
      if constexpr (HOLES == 1 && !GENERAL && !KEEP)
        while (qtail - qhead >= (unsigned)kLanes) flush_queue(kLanes);
      if (!done) rotate();
    }
    if constexpr (HOLES == 1 && !GENERAL && !KEEP)
      while (qtail != qhead) flush_queue(qtail - qhead < (unsigned)kLanes ? qtail - qhead : (unsigned)kLanes);
  }
  if (__builtin_expect(flag_rows != 0, 0)) write_flags();
  return jend;
}

// Which strip a block works on: te_n3_plan.h (the same code runs in the CPU test harness)
__device__ __forceinline__ bool n3_block(const N3Args& a, int& i0, int& own_lo, int& js, int& jend, bool& general) {
  return n3_block_of(a, (int)blockIdx.x, i0, own_lo, js, jend, general);
}

constexpr int kN3TieWaves = 3;  // the TIES march holds 168 registers and 170 bytes of scratch (local copies of the moments, the general
                                // tail on every row); compiled for 2 waves the compiler takes all 256 and spills 500 bytes on top
template <int Q, bool KEEP, int HM, bool TIES = false>
__global__ __launch_bounds__(kLanes) __attribute__((amdgpu_waves_per_eu(TIES ? kN3TieWaves : kN3Waves, TIES ? kN3TieWaves : kN3Waves))) void k_normals3(N3Args a) {
  constexpr int R = Shape<Q>::R;
#if TE_N3_DYN_LDS
  // (a ring whose size the compiler does not see: with the static array it derives 3 waves per SIMD from the LDS size and
  // allocates registers for that, whatever amdgpu_waves_per_eu asks for)
  extern __shared__ double ring[];
#else
  __shared__ double ring[(2 * R + 2) * (kLanes + 2 * R)];
#endif
  __shared__ unsigned long long hmask[2 * R + 2][2];  // invalid cells of the ring rows (HOLES march)
#if TE_N3_PRIO
  __builtin_amdgcn_s_setprio(TE_N3_PRIO);
#endif
  int i0, own_lo, js, jend;
  bool general;
  if (!n3_block(a, i0, own_lo, js, jend, general)) return;
  if constexpr (TIES) {  // (a strip with invalid cells flags its tiles for the fix-up pass itself)
    if (general)
      march3<Q, KEEP, true, 0, true>(a, ring, hmask, i0, own_lo, js, jend);
    else
      march3<Q, KEEP, false, 0, true>(a, ring, hmask, i0, own_lo, js, jend);
    return;
  }
  const int jr = (HM == 1 && a.skip_clean) ? js  // (uniform) nearly every strip holds an invalid cell: no first attempt
                  : general ? march3<Q, KEEP, true, 0>(a, ring, hmask, i0, own_lo, js, jend)
                            : march3<Q, KEEP, false, 0>(a, ring, hmask, i0, own_lo, js, jend);
  if (__builtin_expect(jr < jend, 0)) {  // the strip holds invalid cells: from the row that met the first one on, the march that handles them
    __syncthreads();
    if (general)
      march3<Q, KEEP, true, HM>(a, ring, hmask, i0, own_lo, jr, jend);
    else
      march3<Q, KEEP, false, HM>(a, ring, hmask, i0, own_lo, jr, jend);
  }
}

// The clean march on the slim ring (march3, SLIM), for launches whose elevation layer holds no invalid cell at all -- the
// host counted them at upload (N3Args::no_holes) -- and shapes whose centre column alone reaches rows j +- R.  Should an
// invalid cell turn up all the same (the layer was written behind the library's back), the rest of the strip goes to
// the fix-up pass, which is correct for any input.
constexpr int n3_blocks_per_cu(int lds_bytes) {  // single-wave blocks the LDS admits (2 KiB granules, tools/census.hip), at most 3 waves per SIMD
  const int n = (160 * 1024) / (((lds_bytes + 2047) / 2048) * 2048);
  return n > kN3Waves * 4 ? kN3Waves * 4 : n;
}
// ... and only where the two rows saved buy a resident block: R = 9 (11 -> 12 per CU) and R = 10 (10 -> 11).  At smaller
// radii both rings admit the 12 blocks the registers allow, and the slim march -- chunks of 2 or 3 rows instead of 4, more
// chunk rotations -- measured 2 % slower there (8192^2 at R = 5: 1.156 -> 1.18 ms per launch).
template <int Q>
constexpr bool slim_shape() {
  constexpr int R = Shape<Q>::R, W = kLanes + 2 * R;
  return R >= 2 && Shape<Q>::hw(1) < R && n3_blocks_per_cu(2 * R * W * 8) > n3_blocks_per_cu((2 * R + 2) * W * 8 + (2 * R + 2) * 16);
}
template <int Q>
__global__ __launch_bounds__(kLanes) __attribute__((amdgpu_waves_per_eu(kN3Waves, kN3Waves))) void k_normals3s(N3Args a) {
  constexpr int R = Shape<Q>::R;
  if constexpr (slim_shape<Q>()) {
    __shared__ double ring[(2 * R) * (kLanes + 2 * R)];
    int i0, own_lo, js, jend;
    bool general;
    if (!n3_block(a, i0, own_lo, js, jend, general)) return;
    const int jr = general ? march3<Q, false, true, 0, false, true>(a, ring, nullptr, i0, own_lo, js, jend)
                           : march3<Q, false, false, 0, false, true>(a, ring, nullptr, i0, own_lo, js, jend);
    if (__builtin_expect(jr < jend, 0)) {
      // an invalid cell after all: the rest of the strip goes to the fix-up pass (NaN in my cells, every tile those rows
      // touch flagged -- the pass takes the valid cells of a flagged tile whose slope is NaN; rows of such a tile that the
      // march had already stored are simply computed again)
      const int lane = threadIdx.x;
      const size_t mo = (size_t)(a.map >= 0 ? a.map : (int)blockIdx.z) * (size_t)a.map_cells;
      js = jr;
      if (i0 + lane >= own_lo)
        for (int jj = js; jj < jend; ++jj) {
          const size_t o = mo + (size_t)jj * a.rows + (size_t)(i0 + lane);
          a.slope[o] = a.rough[o] = __builtin_nanf("");
        }
      if (lane == 0) {
        const int tile_base = (a.map >= 0 ? 0 : (int)blockIdx.z) * a.ntx * a.nty;
        const int tc0 = (i0 - a.fi0) >> 6, tc1 = (i0 + kLanes - 1 - a.fi0) >> 6;  // a shifted block straddles two tiles
        for (int tr = (js - a.fj0) >> 4; tr <= (jend - 1 - a.fj0) >> 4; ++tr) {
          const int t0 = tile_base + tr * a.ntx + tc0, t1 = tile_base + tr * a.ntx + tc1;
          a.tile_flags[(t0 % a.fix_groups) * kFixTiles + t0 / a.fix_groups] = 1;
          a.tile_flags[(t1 % a.fix_groups) * kFixTiles + t1 / a.fix_groups] = 1;
        }
      }
    }
  }
}

template <int Q>
int resident_blocks_slim() {
  constexpr int R = Shape<Q>::R;
  constexpr int lds = (2 * R) * (kLanes + 2 * R) * 8;
  int per_cu = (160 * 1024) / (((lds + 2047) / 2048) * 2048);
  if (per_cu > kN3Waves * 4) per_cu = kN3Waves * 4;
  static const int ov = lab_int("TE_N3_BLOCKS_PER_CU", 0);  // measurement aid
  if (ov > 0) per_cu = ov < kN3Waves * 4 ? ov : kN3Waves * 4;
  return per_cu * device_cus();
}

// Resident single-wave blocks per CU and CUs of the current device.  The LDS allocation granularity decides: at R = 9
// (13 120 B) 11 blocks fit, not 12 (tools/census.hip), and a grid of 12 per CU runs in two rounds -- twice the time.
template <int Q, bool KEEP>
int resident_blocks(bool ties = false) {
  // (hipOccupancyMaxActiveBlocksPerMultiprocessor answers 12 for 13 120 B; the hardware admits 11: the census fits an
  // allocation granule of 1280..2048 bytes, the conservative end is used here)
  constexpr int R = Shape<Q>::R;
  constexpr int lds = (2 * R + 2) * (kLanes + 2 * R) * 8 + (2 * R + 2) * 16;  // ring + hole masks
  int per_cu = (160 * 1024) / (((lds + 2047) / 2048) * 2048);
  if (per_cu > kN3Waves * 4) per_cu = kN3Waves * 4;
  if (ties && per_cu > kN3TieWaves * 4) per_cu = kN3TieWaves * 4;
  static const int ov = lab_int("TE_N3_BLOCKS_PER_CU", 0);  // measurement aid
  if (ov > 0) per_cu = ov < kN3Waves * 4 ? ov : kN3Waves * 4;  // (the hole queues are allocated for kN3Waves * 4 per CU)
  return per_cu * device_cus();
}

template <int Q>
bool launch3(const Geo& g, const N3Args& a0, bool keep, int maps, hipStream_t s) {
  N3Args a = a0;
  // As many blocks as fill the resident wave slots in ONE round.  Edge block columns run the general tail on every row
  // (about 1.5x the time of an interior row): their strips are 1.5x shorter so that all blocks finish together.
  static const bool no_slim = lab_flag("TE_N3_NO_SLIM");  // measurement aid
  const bool slim = slim_shape<Q>() && a.no_holes && !keep && a.n_ties == 0 && !no_slim;
  const int resident = slim ? resident_blocks_slim<Q>() : keep ? resident_blocks<Q, true>(a.n_ties != 0) : resident_blocks<Q, false>(a.n_ties != 0);
  constexpr int R = Shape<Q>::R;
  static const int pct_env = lab_int("TE_N3_EDGE_PERCENT", 50);  // measurement aid: strip height of the edge columns in percent
  static const int rows_env = lab_int("TE_N3_STRIP_ROWS", 0);    // measurement aid
  // strips of 32 rows only when the upload counted invalid cells and the dense march serves them (Layers::short_strips): a
  // clean map would pay the extra strip starts for nothing (profiles/r05_experiments.json, exp13)
  bool fits = false;
  const int nblocks = n3_plan_strips(a, g.cols, R, resident, maps, a.short_strips && !slim && a.n_ties == 0, pct_env, rows_env, &fits);
  (void)fits;
  if (nblocks <= 0) return true;
  const dim3 grid((unsigned)nblocks, 1, (unsigned)maps);
  constexpr unsigned kDyn = TE_N3_DYN_LDS ? (unsigned)((2 * R + 2) * (kLanes + 2 * R) * 8) : 0u;  // the ring, when the kernel declares it extern
  if (a.n_ties != 0) {  // tie radius: the whole-cell shapes only
    if constexpr (R * R == Q && R >= 3) {  // (tie radii of one and two cells: k_normals_small, te_normals_small.hip)
      // the kernel knows the circle's cells from R (te_tie_triple.h): the disc's own table must say the same
      if (a.n_gen != tie_triple_cells(R) || a.n_ties != 4 + a.n_gen) return false;
      if (keep)
        hipLaunchKernelGGL((k_normals3<Q, true, 2, true>), grid, dim3(kLanes), kDyn, s, a);
      else
        hipLaunchKernelGGL((k_normals3<Q, false, 2, true>), grid, dim3(kLanes), kDyn, s, a);
      return true;
    }
    return false;
  }
  if (slim) {
    hipLaunchKernelGGL((k_normals3s<Q>), grid, dim3(kLanes), 0, s, a);
    return true;
  }
  // (the kernel that keeps the normals -- the plugin path -- exists with the dense march only)
  // The sparse march indexes its queue scratch by block (blockIdx.x + gridDim.x * blockIdx.z) and the scratch holds one
  // queue per RESIDENT block of the device (normals_hole_queue_bytes): a grid that did not fit one round -- no strip
  // height up to 512 rows fits `capacity` on a large batch, a very large map or a small / partitioned device -- has more
  // blocks than queues and takes the dense march, which needs no scratch.
  const bool queues_fit = (long long)nblocks * (long long)(maps > 0 ? maps : 1) <= (long long)(kN3Waves * 4) * (long long)device_cus();
  if (keep)
    hipLaunchKernelGGL((k_normals3<Q, true, 2>), grid, dim3(kLanes), kDyn, s, a);
  else if (a.sparse_holes && queues_fit)
    hipLaunchKernelGGL((k_normals3<Q, false, 1>), grid, dim3(kLanes), kDyn, s, a);
  else
    hipLaunchKernelGGL((k_normals3<Q, false, 2>), grid, dim3(kLanes), kDyn, s, a);
  return true;
}

}  // namespace

// Shapes this file is instantiated for: every disc shape up to radius 10 (te_march.h) except the single cell.  The file
// is compiled in TE_PARTS parts (build.py: -DTE_PARTS=6 -DTE_PART=k, one object each, side by side) -- the 43 shapes x
// 2 x 4 marches take 13 minutes in one translation unit.  Part k instantiates the shapes of its list and exports one
// function that launches them; part 0 also holds normals_fast3.  -DTE_N3_SHAPES=... (tools) compiles one part with that list.
#define TE_N3_P0(X) X(9) X(20) X(32) X(52) X(61) X(72) X(100)
#define TE_N3_P1(X) X(8) X(18) X(29) X(50) X(58) X(82) X(98)
#define TE_N3_P2(X) X(5) X(17) X(26) X(49) X(53) X(81) X(97)
#define TE_N3_P3(X) X(4) X(16) X(37) X(45) X(68) X(80) X(90)
#define TE_N3_P4(X) X(10) X(13) X(36) X(41) X(65) X(74) X(89)
#define TE_N3_P5(X) X(1) X(2) X(25) X(34) X(40) X(64) X(73) X(85)
#if !defined(TE_PARTS) || defined(TE_N3_SHAPES)
#undef TE_PARTS
#undef TE_PART
#define TE_PARTS 1
#define TE_PART 0
#endif
#if TE_PARTS != 1 && TE_PARTS != 6
#error "te_normals3.hip is cut into 1 or 6 parts"
#endif
#ifndef TE_N3_SHAPES
#if TE_PARTS == 1
#define TE_N3_SHAPES(X) TE_N3_P0(X) TE_N3_P1(X) TE_N3_P2(X) TE_N3_P3(X) TE_N3_P4(X) TE_N3_P5(X)
#else
#define TE_N3_CAT2(a, b) a##b
#define TE_N3_CAT(a, b) TE_N3_CAT2(a, b)
#define TE_N3_SHAPES(X) TE_N3_CAT(TE_N3_P, TE_PART)(X)
#endif
#endif
#define TE_N3_NAME2(k) n3_launch_part##k
#define TE_N3_NAME(k) TE_N3_NAME2(k)

// launches shape Q if it belongs to this part (args: the N3Args block of part 0 -- the same struct in every part)
bool TE_N3_NAME(TE_PART)(int Q, const Geo& g, const void* args, bool keep_normals, int maps, hipStream_t s, bool* ok) {
  const N3Args& a = *static_cast<const N3Args*>(args);
  switch (Q) {
#define X(q)                                      \
  case q:                                         \
    *ok = launch3<q>(g, a, keep_normals, maps, s); \
    return true;
    TE_N3_SHAPES(X)
#undef X
    default:
      return false;
  }
}

#if TE_PART == 0
#if TE_PARTS > 1
bool n3_launch_part1(int Q, const Geo& g, const void* args, bool keep_normals, int maps, hipStream_t s, bool* ok);
bool n3_launch_part2(int Q, const Geo& g, const void* args, bool keep_normals, int maps, hipStream_t s, bool* ok);
bool n3_launch_part3(int Q, const Geo& g, const void* args, bool keep_normals, int maps, hipStream_t s, bool* ok);
bool n3_launch_part4(int Q, const Geo& g, const void* args, bool keep_normals, int maps, hipStream_t s, bool* ok);
bool n3_launch_part5(int Q, const Geo& g, const void* args, bool keep_normals, int maps, hipStream_t s, bool* ok);
#endif

// Scratch of the sparse-hole march: one queue per resident block (the grids are sized to one round of resident blocks,
// whatever the map and the batch).
size_t normals_hole_queue_bytes() { return (size_t)(kN3Waves * 4) * (size_t)device_cus() * (size_t)kHoleQueueBytes; }

// The normals / slope / roughness pass over region r (discs clipped by the map border included); returns false
// if this kernel does not take the case (the caller then uses the sliding kernel of te_slide_normals.hip for everything).
// On success the caller still owes the fix-up pass for the tiles this kernel flagged.
bool normals_fast3(const Geo& g, const ChainParams& p, const Layers& L, bool keep_normals, const Region& r, int* flags,
                   FastGrid* fgp, hipStream_t s) {
  FastGrid& fg = *fgp;
  const Disc& d = p.normals;
  // The shape the kernel slides: the disc, or for a tie radius the disc with its circle (whole-cell radii: every cell
  // on the circle has the norm reach^2, the runs plus the circle are the shape reach^2)
  int shape = d.Q, R = d.R;
  if (d.n_ties != 0) {
    static const bool no_ties = lab_flag("TE_N3_NO_TIES");  // measurement aid: tie radii to the generic kernel as before
    R = d.reach;
    shape = R * R;
    if (no_ties) return false;
    for (int t = 0; t < d.n_ties; ++t)
      if ((int)d.tie_di[t] * d.tie_di[t] + (int)d.tie_dj[t] * d.tie_dj[t] != shape) return false;
  }
  // cells and sum(di^2) of the shape the kernel slides (a tie disc WITH its circle: at a radius of exactly one cell the runs
  // hold the centre alone and the four edge neighbours are all ties -- robot_filter_parameter.yaml's 0.05 m on a 0.05 m map)
  int shape_points = d.npoints;
  long long sii = 0;  // of the runs (the constants of the closed-form tail below)
  for (int dj = -d.R; dj <= d.R; ++dj) {
    const int hw = d.hw[dj < 0 ? -dj : dj];
    for (int di = -hw; di <= hw; ++di) sii += di * di;
  }
  long long shape_sii = sii;
  for (int t = 0; t < d.n_ties; ++t) {
    ++shape_points;
    shape_sii += (int)d.tie_di[t] * d.tie_di[t];
  }
  if (R < 1 || shape < 1 || shape_points < 3) return false;
  N3Args a;
  a.i_lo = r.i0;
  a.i_hi = r.i1;
  a.j_lo = r.j0;
  a.j_hi = r.j1;
  if (a.i_hi - a.i_lo < kLanes || a.j_hi <= a.j_lo || !L.clip_table) return false;
  if (g.rows < 2 * R + 1 || g.cols < 2 * R + 1) return false;  // both borders inside one disc: the clip codes do not cover that
  if ((double)g.rows * (double)g.cols * 4.0 >= 4294967296.0) return false;  // 32-bit byte offsets inside a map
  if (!(g.res * g.res * ((double)shape_sii / (double)shape_points) > 1e-8)) return false;  // NormalVectorsFilter's eigenvalue test would fail everywhere
  // (N, sii of the runs: the constants of the closed-form tail of the tie-free marches; the TIES march takes every disc's
  // x/y moments from its clip table and the general tail)
  const double N = (double)(d.npoints > 1 ? d.npoints : 2);
  a.elev = L.elev;
  a.slope = L.slope;
  a.rough = L.rough;
  a.nx = L.nx;
  a.ny = L.ny;
  a.nz = L.nz;
  a.tile_flags = flags;
  a.rows = g.rows;
  a.cols = g.cols;
  a.map_cells = (long long)g.rows * g.cols;
  a.map = r.map;
  n3_plan_edges(a, g.rows, R);
  a.gtab = d.n_ties ? L.clip_table + kClipInts : L.clip_table;  // (ties: the table of the shape with its circle)
  a.n_ties = d.n_ties;
  a.n_gen = 0;
  for (int t = 0; t < d.n_ties; ++t) a.n_gen += (d.tie_di[t] != 0 && d.tie_dj[t] != 0) ? 1 : 0;
  a.gen_tab = L.clip_table + 2 * kClipInts;
  a.r2 = d.r2;
  a.ax = g.ax;
  a.ay = g.ay;
  a.res = g.res;
  a.Nd = N;
  a.Ni = d.npoints;
  a.sparse_holes = L.sparse_holes && L.hole_queue ? 1 : 0;
  a.no_holes = L.no_holes;
  a.skip_clean = L.skip_clean;
  a.short_strips = L.short_strips;
  a.hole_queue = L.hole_queue;
  a.SIIi = (int)sii;
  a.K1h = 0.5 * N * g.res * g.res * (double)sii;
  a.Kr2 = (N * g.res) * (N * g.res);
  a.kinv = 1.0 / (N * (N - 1.0));
  a.inv_slope_crit = (float)(1.0 / p.slope_crit);
  a.inv_rough_crit = (float)(1.0 / p.rough_crit);
  a.band_slope = clip_band_slope(p.slope_crit);
  a.band_rough = clip_band_rough(p.rough_crit);
  a.Krf = (float)(N * g.res);
  a.fi0 = r.i0;
  a.fj0 = r.j0;
  a.ntx = fg.ntx;
  a.nty = fg.nty;
  a.fix_groups = fix_groups(fg.ntx * fg.nty * fg.nbz);
  const int maps = r.map >= 0 ? 1 : g.batch;
  bool ok = false;
  if (n3_launch_part0(shape, g, &a, keep_normals, maps, s, &ok)) return ok;
#if TE_PARTS > 1
  if (n3_launch_part1(shape, g, &a, keep_normals, maps, s, &ok) || n3_launch_part2(shape, g, &a, keep_normals, maps, s, &ok) ||
      n3_launch_part3(shape, g, &a, keep_normals, maps, s, &ok) || n3_launch_part4(shape, g, &a, keep_normals, maps, s, &ok) ||
      n3_launch_part5(shape, g, &a, keep_normals, maps, s, &ok))
    return ok;
#endif
  return false;
}
#endif  // TE_PART == 0

}  // namespace fast
}  // namespace te
