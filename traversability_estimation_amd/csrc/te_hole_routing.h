// te_hole_routing.h -- which march of k_normals3 a launch takes for the invalid cells of its elevation layer, decided on the
// host from two numbers counted at upload (k_count_invalid, te_shim.hip): the invalid cells and their RUNS in memory order.
// Plain functions of those numbers, in a header of their own: te_shim.hip routes with them and the CPU test
// (tests/cpu/hole_routing_check.cpp, tests/test_hole_routing.py) runs the same code on the counts of the hole benches' maps.
#pragma once

namespace te {

struct HoleCounts {
  long long cells;    // cells of the layer (all maps)
  long long invalid;  // non-finite cells; < 0: unknown (tile uploads, device pointers)
  long long runs;     // runs of invalid cells in memory order; meaningful with invalid >= 0
};

// Unobserved REGIONS rather than scattered cells: the invalid cells come in runs of eight and more on average (speckle: runs
// of one; a region 100 cells wide: runs of 100).
inline bool holes_clustered(const HoleCounts& h) { return h.invalid > 0 && h.runs >= 0 && h.runs * 8 <= h.invalid; }

// The sparse march (HOLES = 1): at most 2 per mille of the cells, scattered.  (A region, however small, is not its
// business: it walks every invalid cell of a disc -- 5x the dense march inside a region.)
inline bool holes_sparse(const HoleCounts& h) {
  return h.invalid > 0 && (double)h.invalid <= 0.002 * (double)h.cells && !holes_clustered(h);
}

// Sparse holes, and so many of them that hardly a strip is free of them (a strip's window is some 8 000 cells: from three
// expected invalid cells per window on): k_normals3's clean first attempt would be given up within its first rows on
// nearly every strip (0.1 % speckle: 99.99 % of them) -- it is skipped.  (Unobserved regions take the dense march and keep
// the attempt: most of their strips ARE clean.)
inline bool holes_skip_clean_march(const HoleCounts& h) { return holes_sparse(h) && (double)h.invalid * 8000.0 >= 3.0 * (double)h.cells; }

// Unobserved regions: the dense march on strips of 32 rows (te_n3_plan.h).  Scattered invalid cells keep the long strips --
// every strip costs alike there, and the extra strip starts and the second round of blocks cost the launch 10 % (1 %
// speckle: 0.61 -> 0.69 ms) --, and so does an unknown count: a map without invalid cells would pay for nothing.
inline bool holes_short_strips(const HoleCounts& h) { return holes_clustered(h) && !holes_sparse(h); }

}  // namespace te
