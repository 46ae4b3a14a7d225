// te_n3_plan.h -- how k_normals3 / k_normals3s (te_normals3.hip) cut a region of a map into strips: the plan the host makes
// per launch and the block -> strip map every workgroup evaluates.  In a header of its own so that the SAME code is
// compiled into the kernels, into the launch code and into the CPU test harness (tests/cpu/n3_plan_check.cpp: every cell
// of the region has exactly one owner, closed-form blocks hold no cell whose disc leaves the map, strips never exceed the
// planned height) -- a Python restatement of this arithmetic cannot fail when the arithmetic changes.
//
// The fields live in the kernel's argument block (N3Args) and in the harness's plain struct alike: everything here is a
// template over "a struct with these members":
//   rows (cells along i, the lane axis), i_lo, i_hi, j_lo, j_hi (the region), nbx, edge0, edge1, n_int, s_int, s_edge,
//   rows_int, rows_edge, n_top, jf_lo, jf_hi.
#pragma once

#if defined(__HIPCC__)
#define TE_N3_HD __host__ __device__ __forceinline__
#else
#define TE_N3_HD inline
#endif

namespace te {
namespace fast {

constexpr int kN3Lanes = 64;           // cells of a block along i (one wavefront)
constexpr int kN3ShortStripRows = 32;  // strip height of the dense march on maps with counted invalid cells

// Block columns whose lanes include cells of the left / right map frame (the first edge0 and the last edge1 block columns:
// the last block is shifted left to end at the region's edge, so its neighbour can reach the frame too).
template <class A>
inline void n3_plan_edges(A& a, int map_rows, int R) {
  a.nbx = (a.i_hi - a.i_lo + kN3Lanes - 1) / kN3Lanes;
  a.edge0 = a.edge1 = 0;
  auto is_edge = [&](int bx) {
    int i0 = a.i_lo + bx * kN3Lanes;
    i0 = i0 + kN3Lanes > a.i_hi ? a.i_hi - kN3Lanes : i0;
    return i0 < R || i0 + kN3Lanes - 1 > map_rows - 1 - R;
  };
  while (a.edge0 < a.nbx && is_edge(a.edge0) && (a.i_lo + a.edge0 * kN3Lanes < R)) ++a.edge0;  // left: columns that reach i < R
  while (a.edge1 < a.nbx - a.edge0 && is_edge(a.nbx - 1 - a.edge1)) ++a.edge1;
}

// Strip heights and counts.  As many blocks as fill the resident wave slots in ONE round (`resident` single-wave blocks
// on the device, shared by `maps` maps); edge block columns run the general tail on every row (about 1.5x the time of an
// interior row): their strips are shorter (edge_percent of the interior height) so that all blocks finish together.
// Returns the number of blocks per map; *fits: one round holds them.
template <class A>
inline int n3_plan_strips(A& a, int map_cols, int R, int resident, int maps, bool short_strips, int edge_percent, int rows_override, bool* fits_out) {
  const int H = a.j_hi - a.j_lo;
  const int capacity = resident / (maps > 0 ? maps : 1);
  const double capacity_f = (double)resident / (double)(maps > 0 ? maps : 1);  // slots per map
  const int ne = a.edge0 + a.edge1;
  a.n_int = a.nbx - ne;
  a.jf_lo = a.j_lo > R ? a.j_lo : (R < a.j_hi ? R : a.j_hi);                                  // first row below the top frame
  a.jf_hi = a.j_hi < map_cols - R ? a.j_hi : (map_cols - R > a.jf_lo ? map_cols - R : a.jf_lo);  // one past the last above the bottom frame
  a.n_top = a.jf_lo > a.j_lo ? a.n_int : 0;
  const int n_bottom = a.j_hi > a.jf_hi ? a.n_int : 0;
  const int Hf = a.jf_hi - a.jf_lo;
  int rows_int = 512;
  const int pct = edge_percent > 0 ? edge_percent : 50;
  auto edge_rows = [&](int h) { return (pct * h + 99) / 100; };
  bool fits = false;
  for (int h = 8; h <= 512; ++h) {  // smallest strip height whose block count fits (small maps: short strips, low latency)
    const int he = edge_rows(h);
    const int blocks = a.n_int * ((Hf + h - 1) / h) + ne * ((H + he - 1) / he) + a.n_top + n_bottom;
    if (blocks <= capacity) {
      rows_int = h;
      fits = true;
      break;
    }
  }
  if (!fits) {
    // More blocks than resident slots whatever the strip height (a large batch -- 512 maps of 512^2: 22 blocks per map
    // against 5.5 slots --, a very large map, a small device): the launch runs in waves of blocks and its last blocks run
    // on a nearly empty device.  Shorter strips make that tail shorter and pay the strip start (staging 2R+2 rows and the
    // direct sums of the first disc: about R + 6 row steps) more often; the height that minimises
    //   (row steps of all blocks) / slots  +  half a block
    // is taken (512 x 512^2 at R = 5: 512 -> 176 rows; normals pass 1.16 -> 0.97 ms).
    const double c0 = (double)(R + 6);
    double best = 0.0;
    for (int h = 16; h <= 512; h += 8) {
      const int he = edge_rows(h);
      const double si = (double)((Hf + h - 1) / h), se = (double)((H + he - 1) / he);
      const double work = (double)a.n_int * (si > 0 ? (double)Hf + si * c0 : 0.0) + 1.5 * (double)ne * ((double)H + se * c0) +
                          (double)(a.n_top + n_bottom) * ((double)R + c0);
      const double t = work / capacity_f + 0.5 * ((double)h + c0);
      if (best == 0.0 || t < best) {
        best = t;
        rows_int = h;
      }
    }
  }
  // UNOBSERVED REGIONS (short_strips: the upload counted invalid cells in runs and the dense march serves them): strips of
  // 32 rows, three times as many blocks as slots -- a strip along a region's edge costs 2.6x a clean one and the pass
  // lasted as long as its slowest strip (te_normals3.hip, launch3)
  if (short_strips && fits && rows_int > kN3ShortStripRows) rows_int = kN3ShortStripRows;
  if (rows_override > 0) rows_int = rows_override;
  a.rows_int = rows_int;
  a.rows_edge = edge_rows(rows_int);
  a.s_int = (a.n_int > 0 && Hf > 0) ? (Hf + a.rows_int - 1) / a.rows_int : 0;
  a.s_edge = ne > 0 ? (H + a.rows_edge - 1) / a.rows_edge : 0;
  if (fits_out) *fits_out = fits;
  return a.n_int * a.s_int + ne * a.s_edge + a.n_top + n_bottom;
}

// Which strip block b works on (uniform): blocks [0, nb_fast) are the interior columns x interior rows, then come the edge
// block columns over all rows, then the top and the bottom frame rows of the interior columns.  false: nothing to do.
// i0: first column of the block's 64 lanes (the last block of a row of blocks is shifted left to end at the region's
// edge), own_lo: first column the block OWNS (it stores only cells at or beyond it); rows [js, jend); general: every row
// takes the x/y moments of its (possibly clipped) disc from the table and the general tail.
template <class A>
TE_N3_HD bool n3_block_of(const A& a, int b, int& i0, int& own_lo, int& js, int& jend, bool& general) {
  int bx;
  general = true;
  const int nb_fast = a.n_int * a.s_int, ne = a.edge0 + a.edge1;
  if (b < nb_fast) {
    general = false;
    bx = a.edge0 + b % a.n_int;
    js = a.jf_lo + (b / a.n_int) * a.rows_int;
    jend = js + a.rows_int < a.jf_hi ? js + a.rows_int : a.jf_hi;
  } else if ((b -= nb_fast) < ne * a.s_edge) {
    const int q = b % ne;
    bx = q < a.edge0 ? q : a.nbx - ne + q;
    js = a.j_lo + (b / ne) * a.rows_edge;
    jend = js + a.rows_edge < a.j_hi ? js + a.rows_edge : a.j_hi;
  } else {
    b -= ne * a.s_edge;
    const bool bottom = b >= a.n_top;  // n_top = n_int if the region has top frame rows, else 0
    bx = a.edge0 + (bottom ? b - a.n_top : b);
    js = bottom ? a.jf_hi : a.j_lo;
    jend = bottom ? a.j_hi : a.jf_lo;
  }
  own_lo = a.i_lo + bx * kN3Lanes;
  i0 = own_lo + kN3Lanes > a.i_hi ? a.i_hi - kN3Lanes : own_lo;  // the last block ends at the edge
  return js < jend;
}

}  // namespace fast
}  // namespace te
