// te_footprint3.hip -- the sliding-sum kernel of the circular footprint pass, laid out like k_normals3.
//
//   TraversabilityMap::traversabilityFootprint(radius, offset)   traversability_estimation/src/TraversabilityMap.cpp:307-318
//     -> isTraversable(center, radiusMax, traversability, radiusMin)              :654-746
//
// Same arithmetic as k_fp_slide (te_footprint.hip): one double per cell, T' + 4096 U (T' = traversability, NaN ->
// default; U = 1 for a cell that fails isTraversableForFilters), the disc sum slides one row per step, a disc without
// an untraversable cell gives the mean, otherwise the lane walks the host-built SpiralIterator table over the rows in
// the ring until the first untraversable cell.  What changed is the instruction stream (te_normals3.hip explains why
// that is what counts on gfx950): ring of exactly 2R+2 rows addressed through rotating chunk base registers with
// immediate offsets (no scalar ring bookkeeping: the old kernel issued 65 scalar instructions per row), one running
// scalar row pointer per layer, 11-12 single-wave blocks per CU instead of 8.
#include "te_internal.h"
#include "te_march.h"

#include <cstdlib>

namespace te {
namespace fast {

namespace {

constexpr int kF3Waves = 3;
constexpr int kF3Head = 24;  // spiral entries every lane walks on its own before the wavefront takes the long walks over
constexpr double kUOff3 = 4096.0;  // as in te_footprint.hip: sums of T' stay below it, so sum(T') and sum(U) split exactly

struct F3Args {
  const float* trav;
  const uint8_t* untrav;
  float* footprint;
  int rows, cols;
  long long map_cells;
  int nbx, strip_rows;
  // the part of the map this launch covers: block columns [bx0, bx0 + nbx_l), output rows [j_lo, j_hi), map (< 0: blockIdx.z)
  int bx0, nbx_l, j_lo, j_hi, map;
  int n_spiral;
  const int16_t* table;  // [n_spiral][4]: di, dj, ring (integer norm), tie flag (never set here: tie-free discs only)
  const int* gtab;       // clip table of the disc: {n, ...} per (ky, kx)
  double rmin, rmax, def, res;
  int inner_q;  // see the tail, step (0)
};

constexpr int f3_chunk_rows(int NR) {
  const int pref[] = {4, 5, 6, 3, 7, 8, 9, 10, 11, 13, 17, 2};
  for (int c : pref)
    if (NR % c == 0) return c;
  return 1;
}

template <int Q>
__global__ __launch_bounds__(kLanes) __attribute__((amdgpu_waves_per_eu(kF3Waves, 4))) void k_fp_slide3(F3Args a) {
  constexpr int R = Shape<Q>::R;
  constexpr int W = kLanes + 2 * R;
  constexpr int NR = 2 * R + 2;
  constexpr int C = f3_chunk_rows(NR);
  constexpr int NC = NR / C;
  constexpr int RB = W * 8;
  __shared__ double ring[NR * W];
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

  unsigned vb[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) vb[c] = (unsigned)(c * C * RB + lane * 8);
  const int hl = lane < 2 * R ? lane : 2 * R - 1;
  const int hcol = hl < R ? hl : kLanes + hl;
  const int vhd = hcol * 8 - lane * 8;
  const bool halo_in = i0 - R + hcol >= 0 && i0 - R + hcol < a.rows;
  const int lhalo = halo_in ? hcol - R : lane;
  const int icol = i0 + lane;
  const int kx = icol < R ? R - icol : (a.rows - 1 - icol < R ? -(R - (a.rows - 1 - icol)) : 0);
  const int nt_mid = a.gtab[((0 + R) * (2 * R + 1) + (kx + R)) * 6];  // cells of my disc on a row away from the top / bottom

  // the spiral table, entry ch * 64 + lane in register ch of lane `lane`: what the wavefront-wide walk below hands out
  constexpr int NTAB = (int)(3.2 * (R + 1) * (R + 1) / kLanes) + 1;  // >= cells of a disc of radius R + 1
  unsigned tabreg[NTAB];
  {
    const unsigned* __restrict__ ptab0 = reinterpret_cast<const unsigned*>(a.table + 4 * kMaxSpiral);
#pragma unroll
    for (int ch = 0; ch < NTAB; ++ch) tabreg[ch] = ch * kLanes + lane < a.n_spiral ? ptab0[ch * kLanes + lane] : 0u;
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
    if (r >= 0 && r < a.cols) {
      pm = ldt[lane];
      ph = ldt[lhalo];
      um = ldu[lane];
      uh = ldu[lhalo];
    }
    ldt += a.rows;
    ldu += a.rows;
  };
  auto stage_row = [&](int r, unsigned vbase, int ro, float pm, float ph, unsigned um, unsigned uh) __attribute__((always_inline)) {
    const bool rin = r >= 0 && r < a.cols;
    const double tm = __builtin_isfinite(pm) ? (double)pm : a.def;  // :719-724
    const double th = __builtin_isfinite(ph) ? (double)ph : a.def;
    const double vm = fma((double)um, kUOff3, tm), vh = fma((double)uh, kUOff3, th);
    *reinterpret_cast<double*>(ringb + vbase + (ro * RB + R * 8)) = rin ? vm : 0.0;  // cells outside the map: nothing
    *reinterpret_cast<double*>(ringb + (vbase + vhd) + ro * RB) = (rin && halo_in) ? vh : 0.0;
  };
  // The strip starts with its first disc summed directly: rows js-R .. js+R+1 go into ring rows 0 .. 2R+1 (the layout
  // step j = js, u = 0 expects) C at a time, then every lane adds the cells of its disc, column by column (unrolled; as rolled loops the sum waited for every LDS read).  Sliding in
  // from an empty disc cost 2R+1 full steps per strip (a fifth of the kernel on the 4096^2 map, all of it on a small
  // one); the direct sum is about three steps' worth of instructions.  Every term is exact, so S is the same number.
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

  double S = 0.0;
  static_for<R + 1>([&](auto dc) __attribute__((always_inline)) {
    constexpr int d = decltype(dc)::value;
    constexpr int h = Shape<Q>::hw(d);
    double col = 0.0;
    static_for<2 * h + 1>([&](auto rc) __attribute__((always_inline)) {
      constexpr int p = R - h + decltype(rc)::value;  // ring row of map row js - h + rc
      const char* row = ringb + vb[p / C] + (p % C) * RB;
      col += *reinterpret_cast<const double*>(row + (R + d) * 8);
      if (d != 0) col += *reinterpret_cast<const double*>(row + (R - d) * 8);
    });
    S += col;
  });
  gfloat* p_out = (gfloat*)(a.footprint + mo + (size_t)js * a.rows + i0);
  float out = 0.0f;
  double rnt = 1.0 / (double)nt_mid;
  const double drmin = a.rmin, inv_span = 1.0 / (a.rmax - a.rmin);

  auto tail = [&](int j, int u) __attribute__((always_inline)) {
    int nt = nt_mid;
    double rn = rnt;
    const int ky = j < R ? R - j : (a.cols - 1 - j < R ? -(R - (a.cols - 1 - j)) : 0);  // uniform
    if (__builtin_expect(ky != 0, 0)) {
      nt = a.gtab[((ky + R) * (2 * R + 1) + (kx + R)) * 6];
      rn = 1.0 / (double)nt;
    }
    out = (float)(S * rn);  // :732-735 no untraversable cell in the footprint: the mean
    const bool blocked = S >= 0.5 * kUOff3;
    if (__builtin_expect(__any(blocked), 0)) {
      // walk the spiral until the first untraversable cell :687-717; logical row j+dj sits dj+R rows below the
      // oldest row of the ring, which is row u of the chunk vb[0] points to.
      const int slot0 = (int)((__builtin_amdgcn_readfirstlane(vb[0])) / RB) + u;
      const unsigned* __restrict__ ptab = reinterpret_cast<const unsigned*>(a.table + 4 * kMaxSpiral);  // packed entries
      auto slot_of = [&](int dj) __attribute__((always_inline)) {
        int sl = slot0 + dj + R;
        sl = sl >= NR ? sl - NR : sl;
        return sl >= NR ? sl - NR : sl;
      };
      auto value_at = [&](int ring_no, double t, int ncells) __attribute__((always_inline)) {
        const double ru = (double)ring_no * a.res;  // getCurrentRadius()
        if (drmin == 0.0 || ru <= drmin) return 0.0f;  // :694-704
        const double factor = ((ru - drmin) * inv_span + 1.0) / 2.0;  // :705-711
        t *= factor / ncells;
        return (float)t;
      };
      bool found = !blocked;
      // (0) An untraversable cell within the inner radius makes the footprint 0 whatever comes before it (:694-704; the
      // spiral visits the rings in order), so a disc is first searched for one directly: all lanes at once, a few hundred
      // ring reads, no table.  inner_q: largest di^2 + dj^2 of the rings that lie within the inner radius and are taken
      // whole by the SpiralIterator (-1: none).  On a map full of obstacles most discs end here.
      if (!found && drmin == 0.0) {
        out = 0.0f;
        found = true;
      }
      if (!found && a.inner_q >= 0) {
        const int dm = (int)__builtin_sqrtf((float)a.inner_q);
        int hits = 0;
        for (int dj = -dm; dj <= dm; ++dj) {
          const int hwi = (int)__builtin_sqrtf((float)(a.inner_q - dj * dj));
          const double* row = ring + slot_of(dj) * W + lane + R;
#pragma unroll 8
          for (int di = -hwi; di <= hwi; ++di) hits += row[di] >= 0.5 * kUOff3 ? 1 : 0;
        }
        if (hits > 0) {
          out = 0.0f;
          found = true;
        }
      }
      // (1) every lane walks the head of its own spiral, eight entries per trip with their eight ring cells fetched
      // together (one entry per trip made every lane wait for a table load and an LDS read in turn).  Pointless after (0).
      if (!found && a.inner_q < 8) {
        double t = 0.0;
        int ncells = 0;
        const bool inner = kx == 0 && j >= R && j < a.cols - R;  // my whole disc lies inside the map
        const int n_head = a.n_spiral < kF3Head ? a.n_spiral : kF3Head;
        for (int k0 = 0; k0 < n_head && !found; k0 += 8) {
          double v[8];
          bool in[8];
          int ring_no[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int kk = k0 + q < n_head ? k0 + q : n_head - 1;
            const unsigned w = ptab[kk];  // uniform: a scalar load
            const int di = (int)(signed char)(w & 0xffu), dj = (int)(signed char)((w >> 8) & 0xffu);
            ring_no[q] = (int)((w >> 16) & 0xffu);
            const int ii = icol + di, jj = j + dj;
            in[q] = k0 + q < n_head && (inner || (ii >= 0 && ii < a.rows && jj >= 0 && jj < a.cols));
            v[q] = ring[slot_of(dj) * W + lane + R + (in[q] ? di : 0)];
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            if (found || !in[q]) continue;
            if (v[q] >= 0.5 * kUOff3) {
              out = value_at(ring_no[q], t, ncells);
              found = true;
            } else {
              ncells++;
              t += v[q];
            }
          }
        }
      }
      // (2) the discs whose first untraversable cell lies further out, one at a time with the whole wavefront: lane q
      // takes entry 64 ch + q (held in registers since the kernel started: a table load per chunk was a memory round
      // trip on the critical path), the first untraversable entry comes from a ballot, the sum of the cells before it
      // from one reduction.  A lane walking 700 entries on its own kept the other 63 waiting.
      unsigned long long rest = __ballot(!found);
      while (rest != 0ull) {
        const int l = __builtin_ctzll(rest);
        rest &= rest - 1ull;
        const int ic = i0 + l;
        double acc = 0.0;
        int cnt = 0;
        float oc = __builtin_nanf("");
        bool done = false;
        static_for<NTAB>([&](auto chc) __attribute__((always_inline)) {
          constexpr int ch = decltype(chc)::value;
          if (done || ch * kLanes >= a.n_spiral) return;  // uniform
          const unsigned w = tabreg[ch];
          const bool valid = ch * kLanes + lane < a.n_spiral;
          const int di = (int)(signed char)(w & 0xffu), dj = (int)(signed char)((w >> 8) & 0xffu);
          const int ii = ic + di, jj = j + dj;
          const bool in = valid && ii >= 0 && ii < a.rows && jj >= 0 && jj < a.cols;
          const double v = ring[slot_of(dj) * W + l + R + di];
          const unsigned long long bm = __ballot(in && v >= 0.5 * kUOff3);
          if (bm != 0ull) {
            const int first = __builtin_ctzll(bm);
            const int ring_first = __builtin_amdgcn_readlane((int)((w >> 16) & 0xffu), first);
            done = true;
            if (drmin == 0.0 || (double)ring_first * a.res <= drmin) {  // :694-704: no sum needed
              oc = 0.0f;
              return;
            }
            const bool before = in && lane < first;
            acc += before ? v : 0.0;
            cnt += __popcll(__ballot(before));
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);  // exact terms (see slide): any order
            oc = value_at(ring_first, acc, cnt);
            return;
          }
          acc += in ? v : 0.0;
          cnt += __popcll(__ballot(in));
        });
        if (lane == l) out = oc;  // an untraversable cell is in the disc, so oc was set
      }
    }
  };

  auto slide = [&](auto uc) __attribute__((always_inline)) {
    constexpr int u = decltype(uc)::value;
    double acc = 0.0;
    static_for<R + 1>([&](auto dc) __attribute__((always_inline)) {
      constexpr int d = decltype(dc)::value;
      constexpr int h = Shape<Q>::hw(d);
      constexpr int pl = u + R + 1 + h, pt = u + R - h;
      constexpr int al = (pl / C) % NC, ol = pl % C, at = (pt / C) % NC, ot = pt % C;
      const char* rl = ringb + vb[al];
      const char* rt = ringb + vb[at];
      const double zl = *reinterpret_cast<const double*>(rl + (ol * RB + (R + d) * 8));
      const double zt = *reinterpret_cast<const double*>(rt + (ot * RB + (R + d) * 8));
      if (d == 0) {
        acc = zl - zt;
      } else {
        const double zl2 = *reinterpret_cast<const double*>(rl + (ol * RB + (R - d) * 8));
        const double zt2 = *reinterpret_cast<const double*>(rt + (ot * RB + (R - d) * 8));
        acc += (zl - zt) + (zl2 - zt2);
      }
    });
    S += acc;  // every term is exact (multiples of the float quantum below 2^53), so is any order
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
      p_out[lane] = out;
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
}

template <int Q>
void launch_f3(const F3Args& a0, int batch, hipStream_t s) {
  F3Args a = a0;
  constexpr int R = Shape<Q>::R;
  constexpr int lds = (2 * R + 2) * (kLanes + 2 * R) * 8;
  int per_cu = (160 * 1024) / (((lds + 2047) / 2048) * 2048);  // see te_normals3.hip (resident_blocks)
  if (per_cu > 16) per_cu = 16;
  const int capacity = per_cu * device_cus();
  const int nz = a.map >= 0 ? 1 : (batch > 0 ? batch : 1);
  const int H = a.j_hi - a.j_lo;
  const int per_row = a.nbx_l * nz;
  int strips = capacity / per_row;
  strips = strips < 1 ? 1 : strips;
  int sr = (H + strips - 1) / strips;
  // small maps cannot fill the wave slots: every resident block runs at once, so the launch takes one warm-up plus the
  // rows of one strip -- the shortest strips win (the spiral walks of a row are serial within its wavefront)
  static const int min_strip = lab_int("TE_F3_MIN_STRIP", 1);
  sr = sr < min_strip ? min_strip : (sr > 512 ? 512 : sr);
  sr = sr < 1 ? 1 : sr;
  a.strip_rows = sr;
  const int nstrips = (H + sr - 1) / sr;
  hipLaunchKernelGGL((k_fp_slide3<Q>), dim3((unsigned)(a.nbx_l * nstrips), 1, (unsigned)nz), dim3(kLanes), 0, s, a);
}

}  // namespace

// Shapes: every disc shape up to radius 10 (te_march.h) except the single cell, and for radii 11 .. 16 (the default
// footprint, 0.45 m, is 15 cells at 0.03 m) every sum of two squares up to 256.  Compiled in TE_PARTS parts like
// te_normals3.hip (build.py): part k instantiates its list and exports one launcher, part 0 also holds footprint_slide3.
#define TE_F3_P0(X) X(4) X(16) X(26) X(37) X(50) X(65) X(73) X(85) X(100) X(121) X(136) X(148) X(162) X(178) X(193) X(202) X(212) X(229) X(256)
#define TE_F3_P1(X) X(10) X(13) X(25) X(36) X(49) X(64) X(82) X(98) X(109) X(117) X(130) X(146) X(160) X(173) X(185) X(200) X(226) X(241) X(250)
#define TE_F3_P2(X) X(9) X(20) X(34) X(45) X(58) X(61) X(81) X(97) X(106) X(116) X(128) X(145) X(157) X(170) X(181) X(197) X(225) X(234) X(245)
#define TE_F3_P3(X) X(2) X(8) X(18) X(32) X(41) X(53) X(72) X(80) X(90) X(104) X(113) X(125) X(144) X(153) X(169) X(196) X(208) X(221) X(233) X(244)
#define TE_F3_P4(X) X(1) X(5) X(17) X(29) X(40) X(52) X(68) X(74) X(89) X(101) X(122) X(137) X(149) X(164) X(180) X(194) X(205) X(218) X(232) X(242)
#if !defined(TE_PARTS) || defined(TE_F3_SHAPES)
#undef TE_PARTS
#undef TE_PART
#define TE_PARTS 1
#define TE_PART 0
#endif
#if TE_PARTS != 1 && TE_PARTS != 5
#error "te_footprint3.hip is cut into 1 or 5 parts"
#endif
#ifndef TE_F3_SHAPES
#if TE_PARTS == 1
#define TE_F3_SHAPES(X) TE_F3_P0(X) TE_F3_P1(X) TE_F3_P2(X) TE_F3_P3(X) TE_F3_P4(X)
#else
#define TE_F3_CAT2(a, b) a##b
#define TE_F3_CAT(a, b) TE_F3_CAT2(a, b)
#define TE_F3_SHAPES(X) TE_F3_CAT(TE_F3_P, TE_PART)(X)
#endif
#endif
#define TE_F3_NAME2(k) f3_launch_part##k
#define TE_F3_NAME(k) TE_F3_NAME2(k)

// launches shape Q if it belongs to this part (args: the F3Args block of part 0 -- the same struct in every part)
bool TE_F3_NAME(TE_PART)(int Q, const void* args, int batch, hipStream_t s) {
  const F3Args& a = *static_cast<const F3Args*>(args);
  switch (Q) {
#define X(q)                  \
  case q:                     \
    launch_f3<q>(a, batch, s); \
    return true;
    TE_F3_SHAPES(X)
#undef X
    default:
      return false;
  }
}

#if TE_PART == 0
#if TE_PARTS > 1
bool f3_launch_part1(int Q, const void* args, int batch, hipStream_t s);
bool f3_launch_part2(int Q, const void* args, int batch, hipStream_t s);
bool f3_launch_part3(int Q, const void* args, int batch, hipStream_t s);
bool f3_launch_part4(int Q, const void* args, int batch, hipStream_t s);
#endif

// Largest di^2 + dj^2 of the spiral rings that lie within the inner radius (ring * res <= rmin, getCurrentRadius()'s
// integer norm) and that SpiralIterator takes without its circle test (all but the two outermost); -1: no such ring.
int footprint_inner_q(double res, double rmin, double rmax) {
  const int nrings = (int)ceil(rmax / res);
  int d = -1;
  while (d + 1 <= nrings - 2 && (double)(d + 1) * res <= rmin) ++d;
  return d < 0 ? -1 : (d + 1) * (d + 1) - 1;
}

// The sliding-sum kernel of the footprint pass for a tie-free disc of an instantiated shape; false: not taken.
bool footprint_slide3(const Geo& g, const FootprintParams& p, const Layers& L, const int16_t* spiral_table, const int* clip_table,
                      hipStream_t s, const Region* region) {
  const Disc& d = p.fp_disc;
  static const bool off = lab_flag("TE_NO_F3");
  if (off || d.n_ties != 0 || d.Q < 1 || d.R < 1 || p.reach != d.R || g.rows < kLanes || g.rows < 2 * d.R + 1 || g.cols < 2 * d.R + 1)
    return false;
  if ((double)g.rows * (double)g.cols * 4.0 >= 4294967296.0) return false;
  if (p.n_spiral > ((int)(3.2 * (d.R + 1) * (d.R + 1) / kLanes) + 1) * kLanes) return false;  // the kernel's table registers
  F3Args a;
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
  a.n_spiral = p.n_spiral;
  a.table = spiral_table;
  a.gtab = clip_table;
  a.rmin = p.rmin;
  a.rmax = p.rmax;
  a.def = p.def;
  a.res = g.res;
  a.inner_q = footprint_inner_q(g.res, p.rmin, p.rmax);
  if (f3_launch_part0(d.Q, &a, g.batch, s)) return true;
#if TE_PARTS > 1
  if (f3_launch_part1(d.Q, &a, g.batch, s) || f3_launch_part2(d.Q, &a, g.batch, s) || f3_launch_part3(d.Q, &a, g.batch, s) ||
      f3_launch_part4(d.Q, &a, g.batch, s))
    return true;
#endif
  return false;
}
#endif  // TE_PART == 0

}  // namespace fast
}  // namespace te
