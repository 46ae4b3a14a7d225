// te_footprint4.hip -- the sliding-sum kernel of the circular footprint pass on 32-bit fixed point FOR TIE RADII (radius a
// whole number of cells: the reference's own 0.45 m at 0.03 m), and k_fp_blocked, the second half of every fixed-point pass.
// (Tie-free radii take the scatter-form sum of te_footprint5.hip since round 4; this kernel's tie-free instantiations were
// retired in round 5.)
//
//   TraversabilityMap::traversabilityFootprint(radius, offset)   traversability_estimation/src/TraversabilityMap.cpp:307-318
//     -> isTraversable(center, radiusMax, traversability, radiusMin)              :654-746
//
// k_fp_slide3 (te_footprint3.hip) slides one DOUBLE per cell and is bound by the instructions one wave can issue: 38
// ds_read_b64 + 38 v_add_f64 + scalar and wait instructions per row, 3 waves per SIMD, 11 blocks per CU (13 KB rings).
// The footprint value is a mean of at most a few hundred traversability values in [0, 1] compared at 1e-5 -- it does
// not need 53 bits.  Here a cell is ONE 32-bit word
//     cell = round(T' * 2^k) | U << 24        T' = traversability (NaN -> default), U = 1: fails isTraversableForFilters
// with k chosen by the host such that the sum of one disc edge (2R+1 cells) stays below 2^24: the leading and the
// trailing edge are each added up with v_add3_u32 in the packed form (T in bits 0..23, the number of untraversable
// cells in bits 24..31, neither field can overflow), then  X += lead - trail  (mod 2^32) and  n_U += lead>>24 - trail>>24;
// sum(T) = X - (n_U << 24) exactly.  Integer sums do not drift, so the slide is exact in the fixed-point values;
// the only error is the rounding of T' to 2^-k (k = 19 at R = 9: 9.5e-7 per cell, hence for the mean).
// Half the LDS (20 rows x 82 words = 6.5 KB: 16 blocks per CU, 4 waves per SIMD), half the LDS instructions
// (ds_read2_b32 takes the cells e and -e of a row together), a third fewer vector instructions.
// A disc that holds an untraversable cell: 0 if that is its own centre (:694-704).  Otherwise the disc is not walked
// here: its cell goes onto a list and k_fp_blocked takes the list afterwards, one disc per wavefront at a time, from
// the layers (L2) in double like isTraversable() does, on the whole GPU.  Walking inside the march serialises exactly
// where the work is: a strip that runs along a kerb was busy for a millisecond while the rest of the GPU had finished
// (4096^2 with 3000 boxes: 1.12 ms in this kernel).  The list is filled in chunks of kF4Chunk entries a block reserves
// with one atomic (one atomic per blocked row made 10^5 of them queue on one address: 1.6 ms); the unused tail of a
// block's last chunk holds kF4NoCell.
// TIE RADII (radius / resolution a whole number of cells -- the reference's own 0.45 m at 0.03 m): the cells exactly on
// the circle belong to a disc or not as SpiralIterator::isInside() decides from rounded positions, per centre.  The
// kernel (TIES = true, instantiated for the whole-cell shapes Q = R^2) slides the full shape, circle included, and every
// row takes the rejected circle cells of its centre out again: one ring read and the reference's own test per cell
// on the circle (12 at 15 cells: the four axis cells and the eight of the triple 9-12-15, known from R at compile time;
// the test's dx * dx is a lane's constant and its dy * dy a row's -- round 6: 401 -> see DESIGN.md us at 15 cells).  k_fp_blocked applies the same test to the table entries that carry the tie flag.
// Used when the host can bound the traversability values (layer written by the chain with non-negative weights);
// otherwise, and for radii whose k would drop below 17, k_fp_slide3 serves.
#include "te_internal.h"
#include "te_march.h"
#include "te_tie_triple.h"

#include <cstdlib>
#include <type_traits>

namespace te {
namespace fast {

namespace {

constexpr int kF4Waves = 4;

struct F4Args {
  const float* trav;
  const uint8_t* untrav;
  float* footprint;
  int rows, cols;
  long long map_cells;
  int nbx, strip_rows;
  // the part of the map this launch covers: block columns [bx0, bx0 + nbx_l), output rows [j_lo, j_hi), map (< 0: blockIdx.z)
  int bx0, nbx_l, j_lo, j_hi, map;
  const int* gtab;  // clip table of the disc: {n, ...} per (ky, kx)
  double rmin;      // inner radius: 0 makes every disc with an untraversable cell 0 (:694-704), no walk needed
  float def, scale;    // cell = (unsigned)(T' * scale + 0.5) | U << 24, scale = 2^k
  double inv_scale;    // 2^-k
  // tie radii (see the header): the circle of a whole-cell radius R holds (+-R, 0), (0, +-R) and n_gen offsets with both
  // parts non-zero (gen_tab: di & 0xff | (dj & 0xff) << 8); r2, ax, ay, res: what isInside() needs.  n_ties = 0 otherwise
  int n_ties, n_gen;
  const int* gen_tab;
  double r2, ax, ay, res;
  unsigned* blocked_list;   // cells (index into the layer, all maps) whose disc holds an untraversable cell ...
  unsigned* blocked_count;  // ... [0] how many entries are reserved, [1] how many hold a cell (k_fp_mask resets both: it runs
                            // before this kernel in every footprint pass), [3] see k_fp_blocked
  int chunk;                // entries a block reserves at a time: kF4Chunk, less for strips shorter than four rows
  size_t list_cap;          // entries the list holds (host side: launch_f4 refuses a grid whose unfinished chunks might not fit)
};

constexpr int f4_chunk_rows(int NR) {
  const int pref[] = {4, 5, 6, 3, 7, 8, 9, 10, 11, 13, 17, 2};
  for (int c : pref)
    if (NR % c == 0) return c;
  return 1;
}

template <int Q, bool TIES>
__global__ __launch_bounds__(kLanes) __attribute__((amdgpu_waves_per_eu(kF4Waves, kF4Waves))) void k_fp_slide4(F4Args a) {
  constexpr int R = Shape<Q>::R;
  constexpr int W = kLanes + 2 * R;
  constexpr int NR = 2 * R + 2;
  constexpr int C = f4_chunk_rows(NR);
  constexpr int NC = NR / C;
  constexpr int RB = W * 4;
  __shared__ unsigned ring[NR * W];
  char* const ringb = reinterpret_cast<char*>(ring);
  typedef const float __attribute__((address_space(1))) cgfloat;
  typedef const uint8_t __attribute__((address_space(1))) cgbyte;
  typedef float __attribute__((address_space(1))) gfloat;

  const int lane = threadIdx.x;
  const int bx = a.bx0 + (int)blockIdx.x % a.nbx_l, strip = (int)blockIdx.x / a.nbx_l;
  int i0 = bx * kLanes;
  i0 = i0 + kLanes > a.rows ? a.rows - kLanes : i0;  // the last block ends at the map edge (rows >= 64)
  const int js = a.j_lo + strip * a.strip_rows;
  if (js >= a.j_hi) return;
  const int jend = js + a.strip_rows < a.j_hi ? js + a.strip_rows : a.j_hi;
  const size_t mo = (size_t)(a.map >= 0 ? a.map : (int)blockIdx.z) * (size_t)a.map_cells;
  if (blockIdx.x == 0 && blockIdx.z == 0 && lane == 0) a.blocked_count[3] = 0u;  // for k_fp_blocked: any spiral entry can be untraversable (k_fp_slide5: beyond its inner disc)

  unsigned vb[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) vb[c] = (unsigned)(c * C * RB + lane * 4);
  const int hl = lane < 2 * R ? lane : 2 * R - 1;
  const int hcol = hl < R ? hl : kLanes + hl;
  const int vhd = hcol * 4 - lane * 4;
  const bool halo_in = i0 - R + hcol >= 0 && i0 - R + hcol < a.rows;
  const int lhalo = halo_in ? hcol - R : lane;
  const int icol = i0 + lane;
  const int kx = icol < R ? R - icol : (a.rows - 1 - icol < R ? -(R - (a.rows - 1 - icol)) : 0);
  const int nt_mid = a.gtab[((0 + R) * (2 * R + 1) + (kx + R)) * 6];  // cells of my disc on a row away from the top / bottom

  // the last block of a row of blocks is shifted left to end at the map edge: the columns it shares with its neighbour
  // are the neighbour's (one store, one list entry per cell)
  const bool own = icol >= bx * kLanes;
  // TIES: the lanes for which isInside() rejects (icol + R, j) / (icol - R, j) -- the same for every j (dy = 0 exactly)
  unsigned long long xfail_p = 0ull, xfail_m = 0ull;
  if constexpr (TIES) {
    const double xi = a.ax + a.res * (double)(-icol);  // cell_x (te_geom.h)
    const double dxp = (a.ax + a.res * (double)(-(icol + R))) - xi, dxm = (a.ax + a.res * (double)(-(icol - R))) - xi;
    xfail_p = __ballot(!(dxp * dxp + 0.0 <= a.r2) && icol + R < a.rows);
    xfail_m = __ballot(!(dxm * dxm + 0.0 <= a.r2) && icol - R >= 0);
  }
  // TIES, the triple's cells: isInside() tests dx * dx + dy * dy <= r2 with dx a function of the lane and dy of the row --
  // dx * dx once per lane (here), dy * dy once per row (tail); a cell outside the map is never taken out: -inf passes the test
  constexpr int TA = TIES ? tie_triple_a(R) : 0, TB = TIES ? tie_triple_b(R) : 0;
  static_assert(!TIES || tie_triples(R) <= 1, "one Pythagorean triple per radius (whole-cell radii up to 16 cells)");
  double dxsq[4] = {0.0, 0.0, 0.0, 0.0};  // di = -TB, -TA, +TA, +TB
  if constexpr (TA != 0) {
    const double xi = a.ax + a.res * (double)(-icol);
    constexpr int di4[4] = {-TB, -TA, TA, TB};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double dx = (a.ax + a.res * (double)(-(icol + di4[q]))) - xi;
      dxsq[q] = (unsigned)(icol + di4[q]) < (unsigned)a.rows ? dx * dx : -__builtin_inf();
    }
  }

  // rows are loaded C steps before they are staged (a queue slot per unrolled position): with one step of lead the
  // wave waited for memory 38 % of its time (SQ_WAIT_ANY, profiles/r02_sq_counters.json)
  float pmq[C], phq[C];
  unsigned umq[C], uhq[C];
#pragma unroll
  for (int k = 0; k < C; ++k) {
    pmq[k] = phq[k] = 0.0f;
    umq[k] = uhq[k] = 0;
  }
  cgfloat* ldt = (cgfloat*)(a.trav + mo + ((long long)(js - R) * a.rows + i0));
  cgbyte* ldu = (cgbyte*)(a.untrav + mo + ((long long)(js - R) * a.rows + i0));
  auto load_row = [&](int r, float& pm, float& ph, unsigned& um, unsigned& uh) __attribute__((always_inline)) {
    if ((unsigned)r < (unsigned)a.cols) {  // 0 <= r < cols in one compare
      pm = ldt[lane];
      ph = ldt[lhalo];
      um = ldu[lane];
      uh = ldu[lhalo];
    }
    ldt += a.rows;
    ldu += a.rows;
  };
  auto stage_row = [&](int r, unsigned vbase, int ro, float pm, float ph, unsigned um, unsigned uh) __attribute__((always_inline)) {
    const bool rin = (unsigned)r < (unsigned)a.cols;
    const float tm = __builtin_isfinite(pm) ? pm : a.def;  // :719-724
    const float th = __builtin_isfinite(ph) ? ph : a.def;
    // round(T' * 2^k): the product is exact, + 0.5 is exact below 2^23, the conversion truncates (and clamps at 0)
    const unsigned vm = (unsigned)__builtin_fmaf(tm, a.scale, 0.5f) | (um << 24);
    const unsigned vh = (unsigned)__builtin_fmaf(th, a.scale, 0.5f) | (uh << 24);
    *reinterpret_cast<unsigned*>(ringb + vbase + (ro * RB + R * 4)) = rin ? vm : 0u;  // cells outside the map: nothing
    *reinterpret_cast<unsigned*>(ringb + (vbase + vhd) + ro * RB) = (rin && halo_in) ? vh : 0u;
  };
  // The strip starts with its first disc summed directly: rows js-R .. js+R+1 go into ring rows 0 .. 2R+1 (the layout
  // step j = js, u = 0 expects) C at a time, then every lane adds the cells of its disc, column by column (a column has
  // at most 2R+1 cells, so its packed sum cannot overflow a field).
  __syncthreads();
  static_for<NC>([&](auto cc) __attribute__((always_inline)) {
    constexpr int c = decltype(cc)::value;
#pragma unroll
    for (int k = 0; k < C; ++k) load_row(js - R + c * C + k, pmq[k], phq[k], umq[k], uhq[k]);
#pragma unroll
    for (int k = 0; k < C; ++k) stage_row(js - R + c * C + k, vb[c], k, pmq[k], phq[k], umq[k], uhq[k]);
  });
#pragma unroll
  for (int k = 0; k < C; ++k) load_row(js + R + 2 + k, pmq[k], phq[k], umq[k], uhq[k]);  // rows j+2+R of the first C steps

  unsigned X = 0;   // sum over the disc of the packed cells, mod 2^32
  unsigned NU = 0;  // untraversable cells in the disc (exact)
  static_for<R + 1>([&](auto dc) __attribute__((always_inline)) {
    constexpr int d = decltype(dc)::value;
    constexpr int h = Shape<Q>::hw(d);
    unsigned colp = 0, colm = 0;
    static_for<2 * h + 1>([&](auto rc) __attribute__((always_inline)) {
      constexpr int p = R - h + decltype(rc)::value;  // ring row of map row js - h + rc
      const char* row = ringb + vb[p / C] + (p % C) * RB;
      colp += *reinterpret_cast<const unsigned*>(row + (R + d) * 4);
      if (d != 0) colm += *reinterpret_cast<const unsigned*>(row + (R - d) * 4);
    });
    X += colp + colm;
    NU += (colp >> 24) + (colm >> 24);
  });
  gfloat* p_out = (gfloat*)(a.footprint + mo + (size_t)js * a.rows + i0);
  float out = 0.0f;
  bool skip = false;  // this lane's cell of the current row is not stored here (not its own column, or on the list)
  const float rnt = (float)(a.inv_scale / (double)nt_mid);
  const double drmin = a.rmin;
  unsigned chunk_at = 0;  // my chunk of the list: next free entry ...
  int chunk_left = 0;     // ... and how many are left (uniform)
  int listed_total = 0;   // cells this block has put onto the list
  auto fill_chunk = [&]() __attribute__((always_inline)) {
    for (int q = lane; q < chunk_left; q += kLanes) a.blocked_list[chunk_at + (unsigned)q] = kF4NoCell;
  };

  // TIES: what depends on the ROW alone -- isInside() for (0, +-R) (dx = 0 exactly) and dy * dy of the triple's cells -- is
  // evaluated for 64 rows at once, lane l taking row ybase + l, and read back row by row (a ballot bit; v_readlane)
  int ybase = js - kLanes;  // rows [ybase, ybase + 64) are at hand: none yet
  unsigned long long yfail_p = 0ull, yfail_m = 0ull;  // bit l: (0, +R) / (0, -R) of row ybase + l is rejected (and lies in the map)
  double dysq_l[4] = {0.0, 0.0, 0.0, 0.0};            // dy * dy for dj = -TB, -TA, +TA, +TB of row ybase + lane (-inf: outside the map)
  auto rows_ahead = [&](int j0) __attribute__((always_inline)) {
    ybase = j0;
    const int jr = j0 + lane;
    const double yr = a.ay + a.res * (double)(-jr);  // cell_y
    const double dyp = (a.ay + a.res * (double)(-(jr + R))) - yr, dym = (a.ay + a.res * (double)(-(jr - R))) - yr;
    yfail_p = __ballot(!(0.0 + dyp * dyp <= a.r2) && (unsigned)(jr + R) < (unsigned)a.cols);
    yfail_m = __ballot(!(0.0 + dym * dym <= a.r2) && (unsigned)(jr - R) < (unsigned)a.cols);
    if constexpr (TA != 0) {
      constexpr int d4[4] = {-TB, -TA, TA, TB};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double dy = (a.ay + a.res * (double)(-(jr + d4[q]))) - yr;
        dysq_l[q] = (unsigned)(jr + d4[q]) < (unsigned)a.cols ? dy * dy : -__builtin_inf();
      }
    }
  };
  auto row_value = [&](double v, int l) __attribute__((always_inline)) {  // v of lane l (uniform l)
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
  };

  auto tail = [&](int j, int u) __attribute__((always_inline)) {
    int nt = nt_mid;
    float rn = rnt;
    if (__builtin_expect((unsigned)(j - R) >= (unsigned)(a.cols - 2 * R), 0)) {  // a row of the top / bottom frame (uniform)
      const int ky = j < R ? R - j : -(R - (a.cols - 1 - j));
      nt = a.gtab[((ky + R) * (2 * R + 1) + (kx + R)) * 6];
      rn = (float)(a.inv_scale / (double)nt);
    }
    unsigned Xc = X, NUc = NU;
    if constexpr (TIES) {
      // the cells on the circle that isInside() rejects for this centre (grid_map_core SpiralIterator) leave the sum
      // again; a rejected cell outside the map was never in it (0 in the ring, not counted in nt).
      const int slot0 = (int)((unsigned)__builtin_amdgcn_readfirstlane((int)vb[0]) / (unsigned)RB) + u;  // ring slot of map row j - R
      auto wrap = [&](int sl) __attribute__((always_inline)) {
        sl = sl >= NR ? sl - NR : sl;
        return sl >= NR ? sl - NR : sl;
      };
      int nfail = 0;
      auto take_out = [&](bool fail, unsigned wd) __attribute__((always_inline)) {
        Xc -= fail ? wd : 0u;
        NUc -= fail ? wd >> 24 : 0u;
        nfail += fail ? 1 : 0;
      };
      // (+-R, 0): decided per lane before the march
      const unsigned* crow = ring + wrap(slot0 + R) * W + lane + R;
      take_out(((xfail_p >> lane) & 1ull) != 0ull, crow[R]);
      take_out(((xfail_m >> lane) & 1ull) != 0ull, crow[-R]);
      // (0, +-R): dx = 0 exactly, the same answer for every lane of the row -- from rows_ahead
      if (__builtin_expect(j - ybase >= kLanes, 0)) rows_ahead(j);  // (uniform)
      const int yl = __builtin_amdgcn_readfirstlane(j - ybase);
      if ((yfail_m >> yl) & 1ull) take_out(true, ring[wrap(slot0 + R - R) * W + lane + R]);
      if ((yfail_p >> yl) & 1ull) take_out(true, ring[wrap(slot0 + R + R) * W + lane + R]);
      // the others (3-4-5 radii: 5, 10, 13, 15 cells): the reference's test per cell
      if constexpr (TA != 0) {
        constexpr int d4[4] = {-TB, -TA, TA, TB};
        double dysq[4];
        const unsigned* rrow[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          dysq[q] = row_value(dysq_l[q], yl);
          rrow[q] = ring + wrap(slot0 + d4[q] + R) * W + lane + R;
        }
        // (di, dj) = (+-TA, +-TB) and (+-TB, +-TA): index q of d4 pairs with 3 - q' ... spelled out: |di| = TA <-> |dj| = TB
        unsigned sub = 0;  // packed sum of the cells taken out (eight cells: neither field overflows)
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
          constexpr int lo_hi[4][2] = {{1, 2}, {0, 3}, {0, 3}, {1, 2}};  // |d4[qi]| = TB: dj = -+TA (slots 1, 2); = TA: dj = -+TB (slots 0, 3)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const int qj = lo_hi[qi][s2];
            const bool fail = !(dxsq[qi] + dysq[qj] <= a.r2);
            const unsigned wd = rrow[qj][d4[qi]];
            sub += fail ? wd : 0u;
            nfail += fail ? 1 : 0;
          }
        }
        Xc -= sub;
        NUc -= sub >> 24;
      }
      if (__builtin_expect(__any(nfail != 0), 0)) rn = nfail ? (float)(a.inv_scale / (double)(nt - nfail)) : rn;
    }
    const unsigned T = Xc - (NUc << 24);  // sum of the fixed-point traversabilities of the disc, exact
    out = (float)T * rn;  // :732-735 no untraversable cell in the footprint: the mean (T < 2^29: the conversion is good to 2^-25)
    const bool blocked = NUc != 0;
    skip = !own;
    if (__builtin_expect(__any(blocked), 0)) {
      if (drmin == 0.0) {
        out = blocked ? 0.0f : out;  // :694-704 radiusMin = 0: the first untraversable cell, wherever it lies, gives 0
      } else {
        // An untraversable centre cell is the spiral's first cell: 0 (ring 0 lies within any inner radius > 0, :694-704).
        // Logical row j sits R rows below the oldest row of the ring, which is row u of the chunk vb[0] points to.
        // Nothing else is decided here -- a search of the inner disc on the ring (29 reads at 3 cells) made a strip that
        // runs along a kerb take 160 us more than its neighbours, and the kernel ends with its last strip.
        int sl = (int)((unsigned)__builtin_amdgcn_readfirstlane((int)vb[0]) / (unsigned)RB) + u + R;
        sl = sl >= NR ? sl - NR : sl;
        sl = sl >= NR ? sl - NR : sl;
        const bool self = (ring[sl * W + lane + R] >> 24) != 0;
        out = (blocked && self) ? 0.0f : out;
        const bool listed = blocked && own && !self;
        // the others onto the list: k_fp_blocked walks their spirals (and stores their values)
        const unsigned long long bm = __ballot(listed);
        if (bm != 0ull) {
          const int n = __popcll(bm);
          const int rank = __popcll(bm & ((1ull << lane) - 1ull));
          // A row that does not fit the rest of the chunk fills it to the last entry and continues in a new one: every
          // closed chunk is full, so a launch reserves at most (listed cells + one chunk per block) entries -- the bound
          // footprint_slide4 checks against the capacity of the list before it launches.
          unsigned at = chunk_at + (unsigned)rank;
          if (n > chunk_left) {
            const int old_left = chunk_left;
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(a.blocked_count, (unsigned)a.chunk);
            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
            if (rank >= old_left) at = base + (unsigned)(rank - old_left);
            chunk_at = base - (unsigned)old_left;  // (+ n below: the entries of this row that went into the new chunk)
            chunk_left = a.chunk + old_left;
          }
          if (listed) a.blocked_list[at] = (unsigned)(mo + (size_t)j * a.rows + icol);
          chunk_at += (unsigned)n;
          chunk_left -= n;
          listed_total += n;
        }
        skip = skip || listed;
      }
    }
  };

  auto slide = [&](auto uc) __attribute__((always_inline)) {
    constexpr int u = decltype(uc)::value;
    // all ring reads of the step first, then ONE wait for all of them, then the sums (the compiler's own placement waits
    // before every add3 for its two operands: 13 s_waitcnt per step -- with four waves per SIMD it is the instruction
    // count that matters, the latency is covered by the other waves)
    unsigned zl[2 * R + 1], zt[2 * R + 1];
    static_for<R + 1>([&](auto dc) __attribute__((always_inline)) {
      constexpr int d = decltype(dc)::value;
      constexpr int h = Shape<Q>::hw(d);
      constexpr int pl = u + R + 1 + h, pt = u + R - h;
      constexpr int al = (pl / C) % NC, ol = pl % C, at = (pt / C) % NC, ot = pt % C;
      const char* rl = ringb + vb[al];
      const char* rt = ringb + vb[at];
      zl[R + d] = *reinterpret_cast<const unsigned*>(rl + (ol * RB + (R + d) * 4));
      zt[R + d] = *reinterpret_cast<const unsigned*>(rt + (ot * RB + (R + d) * 4));
      if (d != 0) {
        zl[R - d] = *reinterpret_cast<const unsigned*>(rl + (ol * RB + (R - d) * 4));
        zt[R - d] = *reinterpret_cast<const unsigned*>(rt + (ot * RB + (R - d) * 4));
      }
    });
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the compiler would otherwise wait before every add3, for its two operands
    unsigned lead = 0, trail = 0;  // packed sums of the 2R+1 cells of the leading / trailing edge (no field overflows)
    static_for<R + 1>([&](auto dcr) __attribute__((always_inline)) {
      constexpr int d = R - decltype(dcr)::value;
      if (d == 0) {
        lead += zl[R];
        trail += zt[R];
      } else {
        lead += zl[R + d] + zl[R - d];
        trail += zt[R + d] + zt[R - d];
      }
    });
    X += lead - trail;
    NU += (lead >> 24) - (trail >> 24);
  };

  int j = js;
#pragma unroll 1
  while (true) {
    bool finished = false;
    static_for<C>([&](auto uc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value;
      if (finished) return;
      if (__builtin_expect(j >= jend, 0)) {
        finished = true;
        return;
      }
      tail(j, u);
      slide(uc);
      stage_row(j + 2 + R, vb[0], u, pmq[u], phq[u], umq[u], uhq[u]);
      load_row(j + 2 + R + C, pmq[u], phq[u], umq[u], uhq[u]);
      if (!skip) p_out[lane] = out;
      p_out += a.rows;
      ++j;
    });
    if (finished) break;
    if (NC > 1) {
      const unsigned v0 = vb[0];
#pragma unroll
      for (int c = 0; c + 1 < NC; ++c) vb[c] = vb[c + 1];
      vb[NC - 1] = v0;
    }
  }
  fill_chunk();
  if (listed_total != 0 && lane == 0) atomicAdd(a.blocked_count + 1, (unsigned)listed_total);
}

template <int Q>
bool launch_f4(const F4Args& a0, int batch, hipStream_t s) {
  F4Args a = a0;
  constexpr int R = Shape<Q>::R;
  static_assert(R * R == Q, "instantiated for the shapes a tie radius can have: its circle passes through (R, 0)");
  if (a.n_ties == 0) return false;  // (tie-free discs: k_fp_slide5, or the double kernel -- footprint_slide4 does not ask)
  // the kernel knows the circle's cells from R (axis cells + one Pythagorean triple): the disc's own table must say the same
  if (a.n_gen != tie_triple_cells(R) || a.n_ties != 4 + a.n_gen) return false;
  constexpr int lds = (2 * R + 2) * (kLanes + 2 * R) * 4;
  int per_cu = (160 * 1024) / (((lds + 2047) / 2048) * 2048);  // see te_normals3.hip (resident_blocks)
  if (per_cu > kF4Waves * 4) per_cu = kF4Waves * 4;
  static const int per_cu_env = lab_int("TE_F4_BLOCKS_PER_CU", 0);  // measurement aid
  if (per_cu_env > 0 && per_cu_env < kF4Waves * 4) per_cu = per_cu_env;
  const int capacity = per_cu * device_cus();
  const int nz = a.map >= 0 ? 1 : (batch > 0 ? batch : 1);
  const int H = a.j_hi - a.j_lo;
  const int per_row = a.nbx_l * nz;
  int strips = capacity / per_row;
  strips = strips < 1 ? 1 : strips;
  int sr = (H + strips - 1) / strips;
  // small maps cannot fill the wave slots: every resident block runs at once, so the launch takes one warm-up plus the
  // rows of one strip -- the shortest strips win (the spiral walks of a row are serial within its wavefront)
  static const int min_strip = lab_int("TE_F4_MIN_STRIP", 1);
  sr = sr < min_strip ? min_strip : (sr > 512 ? 512 : sr);
  sr = sr < 1 ? 1 : sr;
  a.strip_rows = sr;
  a.chunk = sr >= 4 ? kF4Chunk : (sr * kLanes >= kF4Chunk / 2 ? kF4Chunk / 2 : kLanes);  // (a strip of one row lists at most 64 cells)
  const int nstrips = (H + sr - 1) / sr;
  // every listed cell takes one entry and every block may leave one chunk unfinished (closed chunks are full): the list
  // must hold both, whatever the grid (strips clamped to 512 rows on a very tall map or a small device: more blocks
  // than one round of resident ones).  false: the double kernel serves.
  if ((double)a.nbx_l * (double)nstrips * (double)nz * (double)a.chunk + (double)a.map_cells * (double)nz > (double)a.list_cap) return false;
  const dim3 grid((unsigned)(a.nbx_l * nstrips), 1, (unsigned)nz);
  hipLaunchKernelGGL((k_fp_slide4<Q, true>), grid, dim3(kLanes), 0, s, a);
  return true;
}

}  // namespace

// Shapes: the whole-cell radii 1 .. 16 (Q = R^2), TIES march only.  Until round 5 the kernel was also instantiated for
// every tie-free shape up to radius 16 (97 shapes x 2, five translation units): k_fp_slide5 has served those since round 4
// (bit-identical results, 51 against 61 us on the bench map), and the one tie-free radius it does not take -- 16 cells --
// goes to the double kernel like every unbounded layer.
#define TE_F4_SHAPES_ALL(X) X(1) X(4) X(9) X(16) X(25) X(36) X(49) X(64) X(81) X(100) X(121) X(144) X(169) X(196) X(225) X(256)
#undef TE_PARTS
#undef TE_PART
#define TE_PARTS 1
#define TE_PART 0
#ifndef TE_F4_SHAPES
#define TE_F4_SHAPES(X) TE_F4_SHAPES_ALL(X)
#endif
#define TE_F4_NAME2(k) f4_launch_part##k
#define TE_F4_NAME(k) TE_F4_NAME2(k)

// launches shape Q if it belongs to this part (args: the F4Args block of part 0 -- the same struct in every part)
bool TE_F4_NAME(TE_PART)(int Q, const void* args, int batch, hipStream_t s) {
  const F4Args& a = *static_cast<const F4Args*>(args);
  switch (Q) {
#define X(q) \
  case q:    \
    return launch_f4<q>(a, batch, s);
    TE_F4_SHAPES(X)
#undef X
    default:
      return false;
  }
}

#if TE_PART == 0

namespace {

constexpr int kFBTab = 4;    // chunks of 64 spiral entries a lane keeps in registers (256 entries: radii up to 8 cells)
#ifndef TE_FB_TRIP
#define TE_FB_TRIP 8
#endif
constexpr int kFBTrip = TE_FB_TRIP;   // entries per trip of the per-lane walks, their loads issued together
constexpr int kFBDense = 16;  // cells per wavefront of the launch from which every lane walks a disc of its own (round 4: 8; the lists are
                              // now made of discs walked to their rim -- bench map, 30 / 300 / 1000 / 3000 boxes: per wavefront 0.462 / 0.588 /
                              // 0.843 / 1.449 ms per launch, per lane -- / 0.577 / 0.748 / 1.147, profiles/r05_experiments.json)
static_assert(kMaxSpiral % kFBTrip == 0, "k_fp_blocked reads whole trips of the table");

struct FBArgs {
  const float* trav;
  const uint8_t* untrav;
  float* footprint;
  const unsigned* list;
  const unsigned* count;
  const unsigned* ptab;  // packed spiral entries: di | dj << 8 | ring << 16 | tie << 24 (kMaxSpiral words)
  int n_spiral, rows, cols, reach;
  unsigned map_cells;
  double rmin, rmax, def, res;
  double r2, ax, ay;  // SpiralIterator::isInside for the entries on the circle itself (tie flag)
  int path;           // 0: by the length of the list; 1: one disc per wavefront; 2: one disc per lane (TE_FB_PATH, tests)
};

// SpiralIterator::isInside (grid_map_core) for the cell (i + di, j + dj) of the disc around (i, j): cell centres as
// cell_x / cell_y (te_geom.h) compute them
__device__ __forceinline__ bool fb_on_circle_inside(const FBArgs& a, int i, int j, int di, int dj) {
  const double dx = (a.ax + a.res * (double)(-(i + di))) - (a.ax + a.res * (double)(-i));
  const double dy = (a.ay + a.res * (double)(-(j + dj))) - (a.ay + a.res * (double)(-j));
  return dx * dx + dy * dy <= a.r2;
}

// isTraversable(center, radiusMax, traversability, radiusMin) :654-746 for the cells on the list, straight from the
// layers (L2), in double like the reference.  A wavefront takes `group` consecutive list entries per trip, as few as
// keep every wavefront of the launch busy, and there are two ways to walk the SpiralIterator order (host-built table):
//   short list (a lone kerb: fewer than kFBDense cells per wavefront of the launch) -- ONE DISC PER WAVEFRONT AT A TIME: lane q takes entry 64 ch + q (the
//     first kFBTab chunks stay in registers), a ballot finds the first untraversable entry, the sum of the cells
//     before it -- needed only beyond the inner radius -- is one reduction.  Two or three dependent loads per disc:
//     three boxes on a 4096^2 map take 7 us (50 us with one disc per lane: 15 dependent trips);
//   long list -- ONE DISC PER LANE: all lanes step through the table together (scalar loads), first for the index of
//     their first untraversable cell (a byte load and four instructions per entry), then, if any lane's lies beyond the
//     inner radius, for the sum of the cells before it in the iterator's order.  About 25 instructions per disc; the
//     wave-wide walk needs 250, and 3 million discs (3000 boxes) took it 1.5 ms.
__global__ __launch_bounds__(kLanes) void k_fp_blocked(FBArgs a) {
  const unsigned n = a.count[0], n_cells = a.count[1];  // entries (some hold kF4NoCell: the unused tail of a block's last chunk) / cells
  if (n == 0) return;
  const int lane = threadIdx.x;
  const unsigned nwaves = gridDim.x;
  unsigned group = 1;
  while (group < (unsigned)kLanes && group * nwaves < n) group *= 2;
  const bool per_lane = a.path ? a.path == 2 : n_cells >= (unsigned)kFBDense * nwaves;
  // my entries of the table, and their offsets from the top left corner of the disc's bounding square
  unsigned tw[kFBTab];
  int toff[kFBTab];
#pragma unroll
  for (int ch = 0; ch < kFBTab; ++ch) {
    const int k = ch * kLanes + lane;
    tw[ch] = k < a.n_spiral ? a.ptab[k] : 0u;
    const int di = (int)(signed char)(tw[ch] & 0xffu), dj = (int)(signed char)((tw[ch] >> 8) & 0xffu);
    toff[ch] = (dj + a.reach) * a.rows + (di + a.reach);
  }
  const int ctr = a.reach * a.rows + a.reach;
  const int nch = (a.n_spiral + kLanes - 1) / kLanes;
  // one group of up to 64 list entries (kF4NoCell: none in this lane)
  auto take = [&](const unsigned mycell) __attribute__((always_inline)) {
    const unsigned long long actm = __ballot(mycell != kF4NoCell);
    if (actm == 0ull) return;
    // (cell -> map, column j, row i) once per lane; a lane without a cell takes the first one's: a valid address, and a
    // neighbour's.  (__shfl, not readlane: with readlane the compiler dropped the select and the empty lanes read
    // untrav[0xffffffff] -- a memory fault on the first obstacle map.)
    const unsigned first_cell = (unsigned)__shfl((int)mycell, __builtin_ctzll(actm));
    const unsigned cellv = mycell != kF4NoCell ? mycell : first_cell;
    const unsigned mapv = cellv / a.map_cells, remv = cellv - mapv * a.map_cells;
    const unsigned jv = remv / (unsigned)a.rows, iv = remv - jv * (unsigned)a.rows;
    float myout = 0.0f;
    if (per_lane) {
      // ---- one disc per lane
      const bool act = mycell != kF4NoCell;
      const int i = (int)iv, j = (int)jv;
      const bool all_in = __all(i >= a.reach && i < a.rows - a.reach && j >= a.reach && j < a.cols - a.reach);
      const uint8_t* up = a.untrav + cellv;
      const float* tp = a.trav + cellv;
      // The walk comes in two versions: every disc of the group inside the map (the offset of an entry is the same scalar
      // for all lanes: no bounds, no selects), or not.  ONE pass (round 5; rounds 3-4 walked twice, first for the index of
      // the first untraversable cell, then for the sum before it -- but the lists k_fp_slide5 leaves hold only discs that need
      // the sum, see its inner disc): every lane adds up the cells it passes until it meets its first untraversable one.
      // The finite cells add up in double, the others are counted and enter as n * default (the reference adds them in
      // the iterator's order; at most a few hundred terms in [0, 1]: the order shows in the 16th digit).
      const int N = a.n_spiral;
      const int kstart = (int)a.count[3];  // entries before it cannot be untraversable (the producer saw to that): their mask bytes are not fetched
      int kmin = N;  // index of the first untraversable cell in the iterator's order :690 (N: none yet)
      double sum = 0.0;
      int cnt = 0, ndef = 0;
      auto walk = [&](auto fast) __attribute__((always_inline)) {
        constexpr bool kFast = decltype(fast)::value;
        for (int k0 = 0; k0 < N; k0 += kFBTrip) {
          if (__all(kmin < N)) break;
          uint8_t u[kFBTrip];
          float t[kFBTrip];
          bool in[kFBTrip];
#pragma unroll
          for (int q = 0; q < kFBTrip; ++q) {
            const bool valid = k0 + q < N;                   // uniform
            const unsigned w = valid ? a.ptab[k0 + q] : 0u;  // uniform: a scalar load (past the end: the centre)
            const int di = (int)(signed char)(w & 0xffu), dj = (int)(signed char)((w >> 8) & 0xffu);
            in[q] = valid;
            if (!kFast) in[q] = in[q] && i + di >= 0 && i + di < a.rows && j + dj >= 0 && j + dj < a.cols;
            if ((w >> 24) != 0u) in[q] = in[q] && fb_on_circle_inside(a, i, j, di, dj);  // (uniform branch)
            const int off = kFast ? dj * a.rows + di : (in[q] ? dj * a.rows + di : 0);
            u[q] = k0 + q >= kstart ? up[off] : (uint8_t)0;  // (uniform)
            t[q] = tp[off];
          }
#pragma unroll
          for (int q = 0; q < kFBTrip; ++q) {
            const bool before = kmin == N;
            const bool hit = before && in[q] && u[q] != 0;
            kmin = hit ? k0 + q : kmin;
            const bool add = before && !hit && in[q];
            const bool fin = __builtin_isfinite(t[q]);  // :719-724
            sum += (double)((add && fin) ? t[q] : 0.0f);
            cnt += add ? 1 : 0;
            ndef += (add && !fin) ? 1 : 0;
          }
        }
      };
      if (all_in)
        walk(std::true_type{});
      else
        walk(std::false_type{});
      // its ring decides :694-711: within the inner radius 0, beyond it the weighted mean of the cells before it
      const bool found = kmin < N;
      const int ring_no = found ? (int)((a.ptab[kmin] >> 16) & 0xffu) : 0;
      const double ru = (double)ring_no * a.res;  // getCurrentRadius()
      if (act && !(found && (a.rmin == 0.0 || ru <= a.rmin))) {  // (not found: cannot happen for a listed cell; the mean then, :732-735)
        sum += (double)ndef * a.def;
        const double factor = found ? ((ru - a.rmin) / (a.rmax - a.rmin) + 1.0) / 2.0 : 1.0;  // :705-711
        myout = (float)(sum * (factor / cnt));
      }
      if (act) a.footprint[mycell] = myout;
      return;
    }
    // ---- one disc per wavefront at a time, the next disc's loads in flight while this one is evaluated
    // (a disc that is on the list at all is walked far -- k_fp_slide5 lists only discs without an untraversable cell inside
    // the inner radius -- so the loads of all register-resident chunks, discs of up to 8 cells radius: all of them, are
    // issued together; chunk by chunk every step waited for its own two loads: 10 us per disc)
    struct Disc1 {
      unsigned cell;
      int i, j;
      bool inside;  // the whole bounding square lies in the map
      bool qin[kFBTab];
      uint8_t qu[kFBTab];
      float qt[kFBTab];
    };
    // which of my entries' cells exist (inside the map, on the accepted side of the circle)
    auto entry_in = [&](const Disc1& d, unsigned w, int k0) __attribute__((always_inline)) {
      bool in = k0 + lane < a.n_spiral;
      if (!d.inside) {
        const int ii = d.i + (int)(signed char)(w & 0xffu), jj = d.j + (int)(signed char)((w >> 8) & 0xffu);
        in = in && ii >= 0 && ii < a.rows && jj >= 0 && jj < a.cols;
      }
      if (in && (w >> 24) != 0u) in = fb_on_circle_inside(a, d.i, d.j, (int)(signed char)(w & 0xffu), (int)(signed char)((w >> 8) & 0xffu));
      return in;
    };
    auto issue = [&](int l, Disc1& d) __attribute__((always_inline)) {
      d.cell = (unsigned)__builtin_amdgcn_readlane((int)cellv, l);
      d.i = __builtin_amdgcn_readlane((int)iv, l);
      d.j = __builtin_amdgcn_readlane((int)jv, l);
      d.inside = d.i >= a.reach && d.i < a.rows - a.reach && d.j >= a.reach && d.j < a.cols - a.reach;
      // (a pointer before the layer for a cell near the border: never dereferenced, those lanes read the centre)
      const float* tb = a.trav + ((long long)d.cell - ctr);
      const uint8_t* ub = a.untrav + ((long long)d.cell - ctr);
      static_for<kFBTab>([&](auto chc) __attribute__((always_inline)) {
        constexpr int ch = decltype(chc)::value;
        d.qin[ch] = false;
        d.qu[ch] = 0;
        d.qt[ch] = 0.0f;
        if (ch >= nch) return;  // uniform
        d.qin[ch] = entry_in(d, tw[ch], ch * kLanes);
        const int o = d.qin[ch] ? toff[ch] : ctr;
        d.qu[ch] = ub[o];
        d.qt[ch] = tb[o];
      });
    };
    auto finish = [&](int l, const Disc1& d) __attribute__((always_inline)) {
      const float* tb = a.trav + ((long long)d.cell - ctr);
      const uint8_t* ub = a.untrav + ((long long)d.cell - ctr);
      double acc = 0.0;
      int cnt = 0;
      float oc = 0.0f;
      bool done = false;
      auto chunk = [&](unsigned w, bool in, uint8_t u, float t) __attribute__((always_inline)) {
        const unsigned long long bm = __ballot(in && u != 0);
        const double v = __builtin_isfinite(t) ? (double)t : a.def;  // :719-724
        if (bm != 0ull) {  // the first untraversable cell :690-717
          const int first = __builtin_ctzll(bm);
          const int ring_first = __builtin_amdgcn_readlane((int)((w >> 16) & 0xffu), first);
          const double ru = (double)ring_first * a.res;  // getCurrentRadius()
          done = true;
          if (a.rmin == 0.0 || ru <= a.rmin) return;  // :694-704: 0, no sum needed
          const bool before = in && lane < first;
          acc += before ? v : 0.0;
          cnt += __popcll(__ballot(before));
#pragma unroll
          for (int dd = 32; dd >= 1; dd >>= 1) acc += __shfl_xor(acc, dd);
          const double factor = ((ru - a.rmin) / (a.rmax - a.rmin) + 1.0) / 2.0;  // :705-711
          oc = (float)(acc * (factor / cnt));
          return;
        }
        acc += in ? v : 0.0;
        cnt += __popcll(__ballot(in));
      };
      static_for<kFBTab>([&](auto chc) __attribute__((always_inline)) {
        constexpr int ch = decltype(chc)::value;
        if (done || ch >= nch) return;  // uniform
        chunk(tw[ch], d.qin[ch], d.qu[ch], d.qt[ch]);
      });
      for (int ch = kFBTab; ch < nch && !done; ++ch) {
        const int k = ch * kLanes + lane;
        const unsigned w = k < a.n_spiral ? a.ptab[k] : 0u;
        const int di = (int)(signed char)(w & 0xffu), dj = (int)(signed char)((w >> 8) & 0xffu);
        const bool in = entry_in(d, w, ch * kLanes);
        const int o = in ? (dj + a.reach) * a.rows + (di + a.reach) : ctr;
        chunk(w, in, ub[o], tb[o]);
      }
      if (!done) {  // (no untraversable cell after all: the mean :732-735)
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) acc += __shfl_xor(acc, dd);
        oc = (float)(acc / cnt);
      }
      if (lane == l) myout = oc;
    };
    unsigned long long rest = actm;
    Disc1 cur, nxt;
    int lc = __builtin_ctzll(rest);
    rest &= rest - 1ull;
    issue(lc, cur);
    while (true) {
      const bool more = rest != 0ull;  // (uniform)
      int ln = 0;
      if (more) {
        ln = __builtin_ctzll(rest);
        rest &= rest - 1ull;
        issue(ln, nxt);
      }
      finish(lc, cur);
      if (!more) break;
      cur = nxt;
      lc = ln;
    }
    if (mycell != kF4NoCell) a.footprint[mycell] = myout;
  };
  // `group` consecutive entries per trip, as few as keep every wavefront busy (the list is dense: k_fp_slide5 copies a
  // block's cells into it when its strip is done, k_fp_slide4 fills it chunk by chunk)
  for (unsigned c0 = blockIdx.x * group; c0 < n; c0 += nwaves * group)
    take((unsigned)lane < group && c0 + (unsigned)lane < n ? a.list[c0 + (unsigned)lane] : kF4NoCell);
}

}  // namespace

// Entries of the list beyond one per cell: every block of k_fp_slide4 may leave one chunk unfinished, and a launch has
// at most (resident blocks + one row of blocks) of them -- or, when the strips are clamped to 512 rows (a very tall
// map, a small device), one block per 512 rows of every block column (launch_f4 checks the actual grid against it).
size_t f4_list_slack(int rows, int cols, int batch) {
  const size_t nbx = (size_t)(rows + kLanes - 1) / kLanes, nb = (size_t)(batch > 0 ? batch : 1);
  const size_t one_round = (size_t)32 * (size_t)device_cus() + 2 * nbx * nb;  // (up to 8 waves per SIMD: k_fp_slide5 runs at 5)
  const size_t clamped = nbx * nb * ((size_t)(cols + 511) / 512 + 1);
  // k_fp_slide5 reserves 64 entries per row of EVERY block column, the shifted last one included (launch_f5's capacity test):
  // a map whose rows are not a multiple of 64 needs the columns that block shares with its neighbour once more -- without them
  // rows = 65 or 4033 failed the test and fell to the double kernel for no other reason (advisor, round 5)
  const size_t shared = (nbx * (size_t)kLanes - (size_t)rows) * (size_t)cols * nb;
  return (size_t)kF4Chunk * (one_round > clamped ? one_round : clamped) + shared;
}

// The second half of a footprint pass that used k_fp_slide4: the listed cells (see the header).
void footprint_blocked4(const Geo& g, const FootprintParams& p, const Layers& L, const int16_t* spiral_table, hipStream_t s) {
  if (p.rmin == 0.0) return;  // k_fp_slide4 wrote those cells itself (0)
  FBArgs a;
  a.trav = L.trav;
  a.untrav = L.untrav;
  a.footprint = L.footprint;
  a.list = L.fp_blocked;
  a.count = L.fp_blocked_count;
  a.ptab = reinterpret_cast<const unsigned*>(spiral_table + 4 * kMaxSpiral);
  a.n_spiral = p.n_spiral;
  a.rows = g.rows;
  a.cols = g.cols;
  a.reach = p.reach;
  a.map_cells = (unsigned)((size_t)g.rows * g.cols);
  a.rmin = p.rmin;
  a.rmax = p.rmax;
  a.def = p.def;
  a.res = g.res;
  a.r2 = p.fp_disc.r2;
  a.ax = g.ax;
  a.ay = g.ay;
  a.path = L.fb_walk == 1 || L.fb_walk == 2 ? L.fb_walk : 0;  // te_set_option(TE_OPT_FP_BLOCKED_WALK): both walks give the same values
  const int per_cu = L.fb_blocks_per_cu > 0 && L.fb_blocks_per_cu <= 32 ? L.fb_blocks_per_cu : 24;  // 6 waves per SIMD: 77 registers
  hipLaunchKernelGGL(k_fp_blocked, dim3((unsigned)(per_cu * device_cus())), dim3(kLanes), 0, s, a);
}

// The fixed-point sliding-sum kernel of the footprint pass for a tie-free disc of an instantiated shape; false: not
// taken.  tcap: upper bound of the finite values of the traversability layer, as the host can prove it (the layer was
// written by the chain: w_scale * (w_slope + w_step + w_rough) with non-negative weights); < 0: unknown.
bool footprint_slide4(const Geo& g, const FootprintParams& p, const Layers& L, const int16_t* spiral_table, const int* clip_table,
                      double tcap, hipStream_t s, const Region* region, bool finish) {
  const Disc& d = p.fp_disc;
  static const bool off = lab_flag("TE_NO_F4");
  static const bool no_ties = lab_flag("TE_F4_NO_TIES");  // measurement aid: tie radii to the general kernel as before
  // The shape the kernel slides: the disc itself, or for a tie radius the disc with its circle (whole-cell radii only:
  // every cell on the circle has the norm reach^2, and the runs plus the circle are the shape reach^2).
  if (d.n_ties == 0) return false;  // tie-free discs: k_fp_slide5 (te_footprint5.hip), else the double kernel
  const int R = p.reach, shape = R * R;
  if (no_ties) return false;
  for (int t = 0; t < d.n_ties; ++t)
    if ((int)d.tie_di[t] * d.tie_di[t] + (int)d.tie_dj[t] * d.tie_dj[t] != shape) return false;
  if (off || shape < 1 || R < 1 || p.reach != R || g.rows < kLanes || g.rows < 2 * R + 1 || g.cols < 2 * R + 1) return false;
  if ((double)g.rows * (double)g.cols * 4.0 >= 4294967296.0) return false;
  // 32-bit list entries, and room for every block's unfinished chunk
  if (!L.fp_blocked || !L.fp_blocked_count || (double)g.rows * (double)g.cols * (double)g.batch > (double)L.fp_blocked_cap) return false;
  // the fixed-point scale: (2R+1) cells of at most cap * 2^k + 1/2 each must stay below 2^24 (the packed edge sums), and
  // the default value that replaces NaN has to fit as well
  if (!(tcap >= 0.0) || !(p.def >= 0.0)) return false;
  const double cap = (tcap > p.def ? tcap : p.def) * (1.0 + 1e-6) + 1e-12;
  int k = 23;
  while (k >= 0 && (double)(2 * R + 1) * (cap * ldexp(1.0, k) + 1.0) >= 16777216.0) --k;
  if (k < 17) return false;  // rounding each value to 2^-17 could show at the 1e-5 level: the double kernel serves
  F4Args a;
  a.trav = L.trav;
  a.untrav = L.untrav;
  a.footprint = L.footprint;
  a.rows = g.rows;
  a.cols = g.cols;
  a.map_cells = (long long)g.rows * g.cols;
  a.nbx = (g.rows + kLanes - 1) / kLanes;
  a.strip_rows = 0;
  a.bx0 = region ? region->i0 / kLanes : 0;
  a.nbx_l = region ? (region->i1 - 1) / kLanes - a.bx0 + 1 : a.nbx;
  a.j_lo = region ? region->j0 : 0;
  a.j_hi = region ? region->j1 : g.cols;
  a.map = region ? region->map : -1;
  if (a.j_hi <= a.j_lo || a.nbx_l <= 0) return true;
  a.gtab = d.n_ties ? clip_table + kFpClipInts : clip_table;  // (ties: the table of the shape with its circle)
  a.rmin = p.rmin;
  a.n_ties = d.n_ties;
  a.n_gen = 0;
  for (int t = 0; t < d.n_ties; ++t) a.n_gen += (d.tie_di[t] != 0 && d.tie_dj[t] != 0) ? 1 : 0;
  a.gen_tab = clip_table + 2 * kFpClipInts;  // (te_set_params put them there)
  a.r2 = d.r2;
  a.ax = g.ax;
  a.ay = g.ay;
  a.res = g.res;
  a.def = (float)p.def;
  a.scale = (float)ldexp(1.0, k);
  a.inv_scale = ldexp(1.0, -k);
  a.blocked_list = L.fp_blocked;
  a.blocked_count = L.fp_blocked_count;
  a.list_cap = L.fp_blocked_cap;
  const bool launched = f4_launch_part0(shape, &a, g.batch, s);
  if (launched && finish) footprint_blocked4(g, p, L, spiral_table, s);
  return launched;
}
#endif  // TE_PART == 0

}  // namespace fast
}  // namespace te
