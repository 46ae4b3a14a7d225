// n3_plan_check.cpp -- CPU harness over traversability_estimation_amd/csrc/te_n3_plan.h: the strip plan and the block ->
// strip map that k_normals3 / k_normals3s and their launch code are compiled from (the SAME header: a change of the
// partition arithmetic in the kernel changes what this program checks).  Built and run by tests/test_n3_plan.py.
//
// For every case (map, radius, region, resident slots, short strips): every cell of the region is owned by exactly one
// block; a block's lanes stay inside the region; a strip never exceeds its planned height; blocks that take the
// closed-form tail hold no lane whose disc leaves the map; short strips only shorten.  Exit code 0: all cases hold.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "te_n3_plan.h"

using namespace te::fast;

struct Plan {  // the members te_n3_plan.h's templates use (N3Args in the kernel)
  int i_lo, i_hi, j_lo, j_hi;
  int nbx, edge0, edge1, n_int, s_int, s_edge, rows_int, rows_edge, n_top, jf_lo, jf_hi;
};

struct Case {
  int rows, cols, R, i_lo, i_hi, j_lo, j_hi, slots, maps;
};

static int fails = 0;
#define CHECK(cond, ...)                                   \
  do {                                                     \
    if (!(cond)) {                                         \
      if (++fails <= 20) {                                 \
        std::fprintf(stderr, "FAILED %s: ", #cond);        \
        std::fprintf(stderr, __VA_ARGS__);                 \
        std::fprintf(stderr, "\n");                        \
      }                                                    \
    }                                                      \
  } while (0)

static Plan make(const Case& c, bool short_strips, int* nblocks, bool* fits) {
  Plan a{};
  a.i_lo = c.i_lo;
  a.i_hi = c.i_hi;
  a.j_lo = c.j_lo;
  a.j_hi = c.j_hi;
  n3_plan_edges(a, c.rows, c.R);
  *nblocks = n3_plan_strips(a, c.cols, c.R, c.slots, c.maps, short_strips, 50, 0, fits);
  return a;
}

static void check_case(const Case& c, bool short_strips) {
  int nblocks = 0;
  bool fits = false;
  const Plan a = make(c, short_strips, &nblocks, &fits);
  const int W = c.i_hi - c.i_lo, H = c.j_hi - c.j_lo;
  std::vector<int> owner((size_t)W * H, 0);
  long long closed = 0;
  const int nb_fast = a.n_int * a.s_int, nb_edge = (a.edge0 + a.edge1) * a.s_edge;
  for (int b = 0; b < nblocks; ++b) {
    int i0, own_lo, js, jend;
    bool general;
    if (!n3_block_of(a, b, i0, own_lo, js, jend, general)) continue;
    CHECK(c.i_lo <= i0 && i0 + kN3Lanes <= c.i_hi && c.j_lo <= js && js < jend && jend <= c.j_hi, "block %d outside the region (%dx%d R=%d)", b, c.rows, c.cols, c.R);
    const int limit = b < nb_fast ? a.rows_int : (b < nb_fast + nb_edge ? a.rows_edge : c.R);  // interior / edge column / frame rows
    CHECK(jend - js <= limit, "block %d: %d rows, planned %d", b, jend - js, limit);
    const int lo = i0 > own_lo ? i0 : own_lo;  // the lanes of a shifted block that its neighbour owns store nothing
    for (int j = js; j < jend; ++j)
      for (int i = lo; i < i0 + kN3Lanes; ++i) ++owner[(size_t)(j - c.j_lo) * W + (i - c.i_lo)];
    if (!general) {
      closed += (long long)(jend - js) * (i0 + kN3Lanes - lo);
      // every lane of the block, owned or not, runs the closed form: none of their discs may leave the map
      CHECK(i0 >= c.R && i0 + kN3Lanes - 1 <= c.rows - 1 - c.R && js >= c.R && jend - 1 <= c.cols - 1 - c.R, "closed-form block %d reaches the frame", b);
    }
  }
  long long zero = 0, many = 0;
  for (int v : owner) {
    zero += v == 0;
    many += v > 1;
  }
  CHECK(zero == 0 && many == 0, "%dx%d R=%d region [%d,%d)x[%d,%d) slots %d short %d: %lld cells without owner, %lld with several", c.rows, c.cols, c.R, c.i_lo,
        c.i_hi, c.j_lo, c.j_hi, c.slots, (int)short_strips, zero, many);
  if (c.rows >= 1024 && c.cols >= 1024 && W == c.rows && H == c.cols)
    CHECK(closed * 10 > (long long)W * H * 8, "closed-form blocks are not the bulk of a large map (%lld of %lld)", closed, (long long)W * H);
  if (short_strips) {
    int n0 = 0;
    bool f0 = false;
    const Plan p0 = make(c, false, &n0, &f0);
    CHECK(a.rows_int <= p0.rows_int && nblocks >= n0, "short strips lengthen: %d > %d rows or %d < %d blocks", a.rows_int, p0.rows_int, nblocks, n0);
    if (p0.rows_int <= kN3ShortStripRows) CHECK(a.rows_int == p0.rows_int && nblocks == n0, "short strips changed a plan that was short already");
    if (f0) CHECK(a.rows_int <= (kN3ShortStripRows > 8 ? kN3ShortStripRows : 8), "a plan that fitted one round keeps strips of %d rows", a.rows_int);
  }
}

int main() {
  const Case cases[] = {
      {4096, 4096, 9, 0, 4096, 0, 4096, 11 * 256, 1},
      {4096, 4096, 9, 0, 4096, 0, 4096, 12 * 256, 1},
      {1024, 1024, 5, 0, 1024, 0, 1024, 12 * 256, 1},
      {100, 133, 2, 0, 100, 0, 133, 12 * 256, 1},
      {700, 333, 10, 0, 700, 0, 333, 300, 1},          // more blocks than slots whatever the height
      {512, 512, 5, 0, 512, 0, 512, 12 * 256, 512},    // a batch's share of the slots (6 per map)
      {521, 481, 4, 0, 521, 0, 481, 12 * 256, 1},      // the last block of a row of blocks shifted left
      {64, 64, 3, 0, 64, 0, 64, 12 * 256, 1},          // one block column, both borders in it
      {2048, 2048, 9, 300, 900, 100, 700, 11 * 256, 1},     // a region in the interior
      {2048, 2048, 9, 0, 200, 0, 50, 11 * 256, 1},          // a region in the corner, shorter than the frame is wide
      {2048, 2048, 9, 1900, 2048, 2000, 2048, 11 * 256, 1},
      {4096, 4096, 1, 0, 4096, 0, 4096, 12 * 256, 1},
      {8192, 8192, 5, 0, 8192, 0, 8192, 12 * 256, 1},
      {330, 210, 1, 0, 330, 0, 210, 12 * 256, 3},
  };
  int n = 0;
  for (const Case& c : cases)
    for (int s = 0; s < 2; ++s) {
      check_case(c, s != 0);
      ++n;
    }
  // a sweep of odd sizes and regions (deterministic)
  unsigned long long x = 88172645463325252ull;
  auto rnd = [&](int lo, int hi) {
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    return lo + (int)(x % (unsigned long long)(hi - lo + 1));
  };
  for (int k = 0; k < 400; ++k) {
    Case c;
    c.R = rnd(1, 10);
    c.rows = rnd(64 > 2 * c.R + 1 ? 64 : 2 * c.R + 1, 1500);
    c.cols = rnd(2 * c.R + 1, 1500);
    if (k % 3 == 0) {
      c.i_lo = 0, c.i_hi = c.rows, c.j_lo = 0, c.j_hi = c.cols;
    } else {
      c.i_lo = rnd(0, c.rows - 64);
      c.i_hi = rnd(c.i_lo + 64, c.rows);
      c.j_lo = rnd(0, c.cols - 1);
      c.j_hi = rnd(c.j_lo + 1, c.cols);
    }
    c.slots = rnd(1, 4) == 1 ? rnd(4, 200) : rnd(9, 12) * 256;
    c.maps = rnd(1, 5) == 1 ? rnd(2, 40) : 1;
    check_case(c, (k & 1) != 0);
    ++n;
  }
  std::printf("%d cases, %d failed checks\n", n, fails);
  return fails ? 1 : 0;
}
