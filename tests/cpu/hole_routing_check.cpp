// hole_routing_check.cpp -- prints the march te_shim.hip would pick for an elevation layer of <cells> cells with <invalid>
// invalid ones in <runs> runs, from the header the shim itself routes with (te_hole_routing.h).  tests/test_hole_routing.py.
#include <cstdio>
#include <cstdlib>

#include "te_hole_routing.h"

int main(int argc, char** argv) {
  if (argc != 4) return 2;
  const te::HoleCounts h{std::atoll(argv[1]), std::atoll(argv[2]), std::atoll(argv[3])};
  const char* march = h.invalid == 0 ? "clean" : te::holes_sparse(h) ? "sparse" : te::holes_short_strips(h) ? "dense, short strips" : "dense";
  std::printf("%s%s\n", march, te::holes_skip_clean_march(h) ? ", clean attempt skipped" : "");
  return 0;
}
