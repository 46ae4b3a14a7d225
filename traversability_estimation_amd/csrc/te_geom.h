// te_geom.h -- grid_map_core geometry on the device: cell centres, checkIfPositionWithinMap,
// getIndexFromPosition and the Bresenham LineIterator, in the reference's own double arithmetic.
#pragma once
#include "te_internal.h"

namespace te {

__device__ __forceinline__ double cell_x(const Geo& g, int i) { return g.ax + g.res * (double)(-i); }
__device__ __forceinline__ double cell_y(const Geo& g, int j) { return g.ay + g.res * (double)(-j); }

// checkIfPositionWithinMap (grid_map_core)
__device__ __forceinline__ bool pos_inside(const Geo& g, double x, double y) {
  const double tx = -((x - g.pos_x) - 0.5 * g.len_x);
  const double ty = -((y - g.pos_y) - 0.5 * g.len_y);
  return tx >= 0.0 && ty >= 0.0 && tx < g.len_x && ty < g.len_y;
}
// getIndexFromPosition (grid_map_core)
__device__ __forceinline__ bool pos_to_index(const Geo& g, double x, double y, int& i, int& j) {
  const double vx = ((x - 0.5 * g.len_x) - g.pos_x) / g.res;
  const double vy = ((y - 0.5 * g.len_y) - g.pos_y) / g.res;
  i = (int)(-vx);
  j = (int)(-vy);
  return pos_inside(g, x, y) && i >= 0 && j >= 0 && i < g.rows && j < g.cols;
}

// grid_map::LineIterator (Bresenham from (si,sj) to (ei,ej), both included)
struct LineIt {
  int i, j, inc1i, inc1j, inc2i, inc2j, den, num, numadd, ncells, icell;
  __device__ __forceinline__ void init(int si, int sj, int ei, int ej) {
    icell = 0;
    i = si;
    j = sj;
    const int dx = ei > si ? ei - si : si - ei, dy = ej > sj ? ej - sj : sj - ej;
    inc1i = inc2i = (ei >= si) ? 1 : -1;
    inc1j = inc2j = (ej >= sj) ? 1 : -1;
    if (dx >= dy) {
      inc1i = 0;
      inc2j = 0;
      den = dx;
      num = dx / 2;
      numadd = dy;
      ncells = dx + 1;
    } else {
      inc2i = 0;
      inc1j = 0;
      den = dy;
      num = dy / 2;
      numadd = dx;
      ncells = dy + 1;
    }
  }
  __device__ __forceinline__ bool past_end() const { return icell >= ncells; }
  __device__ __forceinline__ void next() {
    num += numadd;
    if (num >= den) {
      num -= den;
      i += inc1i;
      j += inc1j;
    }
    i += inc2i;
    j += inc2j;
    icell++;
  }
};

}  // namespace te
