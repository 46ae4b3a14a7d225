// Corruption fuzzer for the grid_map_msgs/GridMap and rosbag V2.0 parsers (te_gridmap_msg.hip is plain host C++):
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -Iinclude -Itraversability_estimation_amd/csrc \
//       -x c++ traversability_estimation_amd/csrc/te_gridmap_msg.hip tools/fuzz_msg.cpp -o /tmp/fuzz_msg && /tmp/fuzz_msg
// 400 000 mutated inputs (byte flips, extreme length fields, truncation, spliced garbage) in exact-size heap buffers:
// the parsers must reject or accept them without an ASan / UBSan report and never hand out an offset past the buffer.
#include <cstdio>
#include <cstring>
#include <random>
#include "te_msg.h"
using namespace te::msg;
int main() {
  te_msg_info info;
  memset(&info, 0, sizeof(info));
  info.seq = 7; info.stamp_sec = 12; info.stamp_nsec = 34; strcpy(info.frame_id, "odom");
  info.rows = 7; info.cols = 5; info.resolution = 0.05; info.length_x = 7 * 0.05; info.length_y = 5 * 0.05;
  info.pose[6] = 1; info.start_row = 3; info.start_col = 1;
  const char* names[2] = {"elevation", "variance"};
  const char* basic[1] = {"elevation"};
  Names ln = {2, names}, bn = {1, basic};
  size_t need = message_size(info, ln, bn);
  std::vector<uint8_t> msg(need);
  std::vector<size_t> off;
  std::string err;
  if (!write_skeleton(info, ln, bn, msg.data(), need, off, err)) { printf("skeleton: %s\n", err.c_str()); return 1; }
  for (size_t o : off) for (int k = 0; k < 35; ++k) { float v = (float)k; memcpy(msg.data() + o + 4 * k, &v, 4); }
  size_t bneed = bag_size(need, "grid_map");
  std::vector<uint8_t> bag(bneed);
  size_t written;
  if (!bag_write(msg.data(), need, "grid_map", 12, 34, bag.data(), bneed, written, err)) { printf("bag: %s\n", err.c_str()); return 1; }
  std::mt19937 rng(1);
  long ok = 0, bad = 0;
  for (int which = 0; which < 2; ++which) {
    const std::vector<uint8_t>& src = which ? bag : msg;
    for (int trial = 0; trial < 200000; ++trial) {
      // exact-size heap copy so that ASan sees any overrun
      std::vector<uint8_t> b(src);
      int kind = trial % 4;
      if (kind == 0) { for (int k = 0; k < 1 + (int)(rng() % 3); ++k) b[rng() % b.size()] = (uint8_t)rng(); }
      else if (kind == 1) { size_t at = rng() % (b.size() - 4); uint32_t vals[] = {0u, 1u, 0x7FFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFF0u, (uint32_t)b.size(), (uint32_t)(b.size() - at), (uint32_t)rng()}; uint32_t v = vals[rng() % 8]; memcpy(&b[at], &v, 4); }
      else if (kind == 2) { b.resize(rng() % b.size()); }
      else { size_t at = rng() % b.size(); size_t n = 1 + rng() % 40; std::vector<uint8_t> g(n); for (auto& x : g) x = (uint8_t)rng(); b.insert(b.begin() + at, g.begin(), g.end()); }
      uint8_t* heap = (uint8_t*)malloc(b.size() ? b.size() : 1);
      memcpy(heap, b.data(), b.size());
      bool r;
      if (which) { size_t o, l; r = bag_find(heap, b.size(), "grid_map", o, l, err); if (r && o + l > b.size()) { printf("OOB bag\n"); return 2; } }
      else { View v; r = parse(heap, b.size(), v, err); if (r) for (auto& l : v.layers) if (l.data_off + (size_t)4 * v.info.rows * v.info.cols > b.size()) { printf("OOB msg\n"); return 2; } }
      free(heap);
      (r ? ok : bad)++;
    }
  }
  printf("ok=%ld rejected=%ld\n", ok, bad);
  return 0;
}
