// The circle cells the tie marches address at compile time (te_tie_triple.h), against a brute-force enumeration of the
// lattice points on the circle and against the arithmetic claim the kernels rest on: dx*dx computed once per lane and
// dy*dy once per row, then added, is bit-identical to dx*dx + dy*dy evaluated in place (cell positions as te_geom.h
// forms them; compiled with -ffp-contract=off like the kernels).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <utility>

#include "te_tie_triple.h"

using namespace te::fast;

template <int R>
static int check_radius() {
  int failed = 0;
  std::set<std::pair<int, int>> on_circle;
  for (int di = -R; di <= R; ++di)
    for (int dj = -R; dj <= R; ++dj)
      if (di * di + dj * dj == R * R) on_circle.insert({di, dj});
  constexpr int A = tie_triple_a(R), B = tie_triple_b(R), NT = tie_triples(R);
  std::set<std::pair<int, int>> mine = {{R, 0}, {-R, 0}, {0, R}, {0, -R}};
  if (A != 0)
    for (int sa = -1; sa <= 1; sa += 2)
      for (int sb = -1; sb <= 1; sb += 2) {
        mine.insert({sa * A, sb * B});
        mine.insert({sa * B, sb * A});
      }
  if (NT <= 1 && mine != on_circle) {
    std::printf("R = %d: the kernel's cells differ from the circle's (%zu against %zu)\n", R, mine.size(), on_circle.size());
    ++failed;
  }
  if (NT > 1 && on_circle.size() != 4 + 8 * (size_t)NT) {
    std::printf("R = %d: %d triples but %zu circle cells\n", R, NT, on_circle.size());
    ++failed;
  }
  if ((int)mine.size() != 4 + tie_triple_cells(R)) ++failed;
  return failed;
}

int main() {
  int failed = 0;
  failed += check_radius<1>() + check_radius<2>() + check_radius<3>() + check_radius<4>() + check_radius<5>() + check_radius<6>() +
            check_radius<7>() + check_radius<8>() + check_radius<9>() + check_radius<10>() + check_radius<11>() + check_radius<12>() +
            check_radius<13>() + check_radius<14>() + check_radius<15>() + check_radius<16>() + check_radius<17>() + check_radius<20>();
  static_assert(tie_triple_a(5) == 3 && tie_triple_b(5) == 4 && tie_triple_a(10) == 6 && tie_triple_b(10) == 8, "3-4-5");
  static_assert(tie_triple_a(13) == 5 && tie_triple_b(13) == 12 && tie_triple_a(15) == 9 && tie_triple_b(15) == 12, "5-12-13, 9-12-15");
  static_assert(tie_triples(9) == 0 && tie_triples(16) == 0 && tie_triples(25) == 2, "none at 9 and 16 cells; two at 25");
  // the split evaluation: positions as cell_x / cell_y form them (ax + res * (double)(-i)), many origins and resolutions
  unsigned long long tested = 0;
  std::srand(7);
  for (int trial = 0; trial < 4000; ++trial) {
    const double res = (trial % 3 == 0) ? 0.05 : (trial % 3 == 1 ? 0.03 : 0.1);
    const int n = 64 + std::rand() % 4000;
    const double pos = (std::rand() % 40001 - 20000) * 1e-3;
    const double ax = pos + (0.5 * (n * res) - 0.5 * res), ay = -pos + (0.5 * (n * res) - 0.5 * res);
    const int R = (trial % 4 == 0) ? 5 : (trial % 4 == 1 ? 10 : (trial % 4 == 2 ? 13 : 15));
    const int A = tie_triple_a(R), B = tie_triple_b(R);
    const double r2 = (R * res) * (R * res);
    for (int k = 0; k < 64; ++k) {
      const int i = std::rand() % n, j = std::rand() % n;
      const double xi = ax + res * (double)(-i), yj = ay + res * (double)(-j);
      const int d4[4] = {-B, -A, A, B};
      for (int qi = 0; qi < 4; ++qi)
        for (int qj = 0; qj < 4; ++qj) {
          const int di = d4[qi], dj = d4[qj];
          if (di * di + dj * dj != R * R) continue;
          const double dx = (ax + res * (double)(-(i + di))) - xi, dy = (ay + res * (double)(-(j + dj))) - yj;
          const bool in_place = dx * dx + dy * dy <= r2;
          volatile double dxsq = dx * dx, dysq = dy * dy;  // a lane's constant, a row's
          const bool split = dxsq + dysq <= r2;
          // an axis cell's test without its exactly-zero term (k_normals_small)
          const double dx0 = (ax + res * (double)(-i)) - xi;
          const bool axis_full = dx0 * dx0 + dy * dy <= r2, axis_short = dy * dy <= r2;
          if (in_place != split || axis_full != axis_short || dx0 != 0.0) {
            std::printf("trial %d: R %d cell (%d, %d) at (%d, %d): in place %d, split %d, axis %d / %d\n", trial, R, di, dj, i, j, in_place, split, axis_full, axis_short);
            ++failed;
          }
          ++tested;
        }
    }
  }
  std::printf("%llu tests of the split evaluation, %d failed checks\n", tested, failed);
  return failed != 0;
}
