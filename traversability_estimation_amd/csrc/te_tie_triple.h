// te_tie_triple.h -- the cells on the circle of a whole-cell radius, known at compile time.
//
// CircleIterator / SpiralIterator::isInside (un-vendored grid_map_core; call sites StepFilter.cpp:124,157,
// RoughnessFilter.cpp:96, TraversabilityMap.cpp:687) accept a cell from the rounded double positions of centre and cell:
// (p - c).squaredNorm() <= r^2.  At a radius of exactly R cells the cells with di^2 + dj^2 = R^2 are rounding ties the
// kernels decide centre by centre with the reference's own arithmetic: (+-R, 0), (0, +-R), and the eight cells
// (+-A, +-B), (+-B, +-A) of a Pythagorean triple A^2 + B^2 = R^2.  Up to 24 cells a radius has at most one triple:
// 5 (3, 4), 10 (6, 8), 13 (5, 12), 15 (9, 12), 17 (8, 15), 20 (12, 16).  With R a template parameter the marching kernels
// address these cells at immediate offsets, and since dx depends on the lane only and dy on the row only, dx * dx is a
// lane's constant and dy * dy a row's: the test per cell is one addition and one comparison (bit-identical to
// dx * dx + dy * dy <= r2 evaluated in place: the same three operations, -ffp-contract=off).
#pragma once

namespace te {
namespace fast {

constexpr int tie_triples(int R) {
  int n = 0;
  for (int a = 1; a < R; ++a)
    for (int b = a + 1; b < R; ++b) n += (a * a + b * b == R * R) ? 1 : 0;
  return n;
}
constexpr int tie_triple_a(int R) {
  for (int a = 1; a < R; ++a)
    for (int b = a + 1; b < R; ++b)
      if (a * a + b * b == R * R) return a;
  return 0;
}
constexpr int tie_triple_b(int R) {
  for (int a = 1; a < R; ++a)
    for (int b = a + 1; b < R; ++b)
      if (a * a + b * b == R * R) return b;
  return 0;
}
// cells of the circle besides the four on the axes
constexpr int tie_triple_cells(int R) { return tie_triple_a(R) != 0 ? 8 : 0; }

}  // namespace fast
}  // namespace te
