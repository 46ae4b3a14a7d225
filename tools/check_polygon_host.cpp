// Host-side checks of te_polygon.hip that need no GPU (run under ASan + UBSan):
//   cd traversability_estimation_amd
//   hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -Xarch_host -fsanitize=address,undefined \
//         -Xarch_host -fno-sanitize-recover=all -I../include -Icsrc -c csrc/te_polygon.hip -o /tmp/te_polygon_host.o
//   hipcc --cuda-host-only -x hip -O1 -g -std=c++17 -ffp-contract=off -fsanitize=address,undefined -fno-sanitize-recover=all \
//         -I../include -Icsrc -c ../tools/check_polygon_host.cpp -o /tmp/check_polygon_host.o
//   hipcc -fsanitize=address,undefined /tmp/check_polygon_host.o /tmp/te_polygon_host.o -o /tmp/check_polygon_host && /tmp/check_polygon_host
// (1) build_path_polygons / build_polygon_table on degenerate input (identical, collinear and duplicate points, empty and
//     300-pose paths, tilted orientations, footprints from 1e-3 to 1e9 cells): structure of every output is validated.
// (2) the offset table against the per-cell crossing-number expression (the arithmetic of k_polygon_footprint and of the
//     oracle): an offset the table calls inside (outside) must be inside (outside) for every centre cell -- 1.3e8 cell tests
//     over 3000 polygons (edges on cell centres, turned by pi/2, random, snapped to half cells; map origins up to 5 km).
#include <cstdio>
#include <cmath>
#include <cstring>
#include <map>
#include <random>
#include "te_internal.h"
// The offset table against the per-cell crossing-number test (the arithmetic of k_polygon_footprint / the oracle):
// an offset the table calls inside (outside) must be inside (outside) for every centre cell.
static bool inside_exact(int n, const double* vx, const double* vy, double px, double py) {
  int cross = 0;
  for (int i = 0, j = n - 1; i < n; j = i++)
    if (((vy[i] > py) != (vy[j] > py)) && (px < (vx[j] - vx[i]) * (py - vy[i]) / (vy[j] - vy[i]) + vx[i])) cross++;
  return cross & 1;
}
static int check_table_classification() {
  std::mt19937_64 rng(11);
  std::uniform_real_distribution<double> u(-1.0, 1.0);
  long checked = 0, uncertain = 0, tables = 0;
  for (int trial = 0; trial < 3000; ++trial) {
    te::Geo g;
    memset((void*)&g, 0, sizeof(g));
    g.rows = 50 + (int)(rng() % 4000); g.cols = 50 + (int)(rng() % 4000); g.batch = 1;
    const double ress[] = {0.01, 0.02, 0.03, 0.04, 0.05, 0.1, 0.25, 0.0373};
    g.res = ress[rng() % 8];
    g.len_x = g.rows * g.res; g.len_y = g.cols * g.res;
    const double big = (trial % 4 == 0) ? 5000.0 : 20.0;
    g.pos_x = u(rng) * big; g.pos_y = u(rng) * big;
    g.ax = g.pos_x + (0.5 * g.len_x - 0.5 * g.res);
    g.ay = g.pos_y + (0.5 * g.len_y - 0.5 * g.res);
    const int n = 3 + (int)(rng() % 6);
    double off[64];
    const int mode = trial % 3;
    if (mode == 0) {  // rectangle aligned with the cell centres (every edge on cell centres), maybe turned like yaw = pi/2
      const double hx = g.res * (1 + rng() % 12), hy = g.res * (1 + rng() % 9);
      double pts[8] = {hx, hy, hx, -hy, -hx, -hy, -hx, hy};
      te::rotate_footprint(4, pts, (trial % 2) ? 1.5707963267948966 : 0.0, off);
    } else {
      std::vector<double> ang(n);
      for (auto& a : ang) a = u(rng) * 3.14159;
      std::sort(ang.begin(), ang.end());
      for (int k = 0; k < n; ++k) {
        double r = g.res * (2 + (rng() % 100) / 10.0);
        off[2 * k] = r * std::cos(ang[k]);
        off[2 * k + 1] = r * std::sin(ang[k]);
        if (mode == 2) { off[2 * k] = std::round(off[2 * k] / (0.5 * g.res)) * 0.5 * g.res; off[2 * k + 1] = std::round(off[2 * k + 1] / (0.5 * g.res)) * 0.5 * g.res; }
      }
    }
    const int nn = mode == 0 ? 4 : n;
    std::vector<unsigned> stream;
    te::PolygonTable tb;
    memset((void*)&tb, 0, sizeof(tb));
    if (!te::build_polygon_table(g, nn, off, stream, tb)) continue;
    ++tables;
    std::map<std::pair<int, int>, int> cls;  // 1 inside, 2 uncertain
    size_t p = tb.first;
    for (int r = 0; r < tb.n_rows; ++r) {
      const unsigned hdr = stream[p++];
      for (unsigned k = 0; k < (hdr >> 8); ++k) {
        const unsigned item = stream[p++];
        for (unsigned m = 0; m < ((item >> 8) & 0xff); ++m)
          cls[{tb.di_min + (int)(hdr & 0xff), tb.dj_min + (int)(item & 0xff) + (int)m}] = (item >> 31) ? 2 : 1;
      }
    }
    for (int c = 0; c < 40; ++c) {
      const int i = (int)(rng() % g.rows), j = (int)(rng() % g.cols);
      const double cx = g.ax + g.res * (double)(-i), cy = g.ay + g.res * (double)(-j);
      double vx[8], vy[8];
      for (int k = 0; k < nn; ++k) { vx[k] = off[2 * k] + cx; vy[k] = off[2 * k + 1] + cy; }
      for (int di = -16; di <= 16; ++di)
        for (int dj = -16; dj <= 16; ++dj) {
          const double px = g.ax + g.res * (double)(-(i + di)), py = g.ay + g.res * (double)(-(j + dj));
          const bool in = inside_exact(nn, vx, vy, px, py);
          auto it = cls.find({di, dj});
          const int cl = it == cls.end() ? 0 : it->second;
          ++checked;
          if (cl == 2) { ++uncertain; continue; }
          if ((cl == 1) != in) {
            printf("MISMATCH trial %d mode %d res %g centre (%d,%d) offset (%d,%d): table %d exact %d\n", trial, mode, g.res, i, j, di, dj, cl, (int)in);
            return 1;
          }
        }
    }
  }
  printf("tables=%ld cell tests=%ld (left to the per-cell expression: %ld)\n", tables, checked, uncertain);
  return 0;
}

static int check_builders() {
  std::mt19937_64 rng(3);
  std::uniform_real_distribution<double> u(-1.0, 1.0);
  long polys = 0, tables = 0, fallbacks = 0;
  // path polygons: degenerate inputs
  for (int trial = 0; trial < 20000; ++trial) {
    const int n_paths = 1 + (int)(rng() % 6), n_points = 1 + (int)(rng() % 8);
    std::vector<int> off(1, 0);
    std::vector<double> poses, pts;
    std::vector<unsigned char> cons;
    const int mode = trial % 5;
    for (int k = 0; k < n_points; ++k) {
      double x = u(rng), y = u(rng);
      if (mode == 1) { x = 0.3; y = 0.3; }              // all points identical
      if (mode == 2) y = x;                              // collinear
      if (mode == 3) { x = std::round(x * 4) / 4; y = std::round(y * 4) / 4; }  // many duplicates / ties
      pts.insert(pts.end(), {x, y, u(rng) * 0.1});
    }
    for (int p = 0; p < n_paths; ++p) {
      int k = (int)(rng() % 400 == 0 ? 300 : rng() % 6);  // sometimes empty, sometimes very long (conservative growth limit)
      double x = u(rng) * 5, y = u(rng) * 5;
      for (int i = 0; i < k; ++i) {
        double yaw = u(rng) * 3.14;
        double v[7] = {x, y, 0, mode == 4 ? u(rng) : 0, mode == 4 ? u(rng) : 0, std::sin(yaw / 2), std::cos(yaw / 2)};
        poses.insert(poses.end(), v, v + 7);
        if (mode != 3 || (i & 1)) { x += u(rng); y += u(rng); }   // repeated poses too
      }
      off.push_back((int)poses.size() / 7);
      cons.push_back(rng() & 1);
    }
    if (poses.empty()) poses.resize(7);
    te::PathPolygons pp;
    te::build_path_polygons(n_paths, off.data(), poses.data(), n_points, pts.data(), (trial & 1) ? cons.data() : nullptr, pp);
    if ((int)pp.first.size() != n_paths || pp.vertex_offset.size() != pp.area.size() + 1) { printf("bad sizes\n"); return 1; }
    for (size_t q = 0; q < pp.area.size(); ++q) {
      if (pp.vertex_offset[q + 1] <= pp.vertex_offset[q]) { printf("empty polygon\n"); return 1; }
      if (!(pp.area[q] >= 0.0) || !(pp.area_previous[q] >= 0.0)) { printf("bad area\n"); return 1; }
    }
    if ((size_t)pp.vertex_offset.back() * 2 != pp.vertex_xy.size()) { printf("bad vertex count\n"); return 1; }
    polys += (long)pp.area.size();
  }
  // offset tables: random polygons and resolutions
  for (int trial = 0; trial < 20000; ++trial) {
    te::Geo g;
    memset((void*)&g, 0, sizeof(g));
    g.rows = 100; g.cols = 80; g.batch = 1;
    g.res = std::pow(10.0, u(rng) * 1.5 - 1.5);   // 0.001 .. 0.03..1
    g.len_x = g.rows * g.res; g.len_y = g.cols * g.res; g.pos_x = u(rng) * 100; g.pos_y = u(rng) * 100;
    const int n = 1 + (int)(rng() % 8);
    double off[64];
    const double scale = (trial % 7 == 0) ? 1e9 : std::pow(10.0, u(rng) * 2);   // tiny .. huge footprints
    for (int k = 0; k < 2 * n; ++k) off[k] = (trial % 3 == 0 ? std::round(u(rng) * 8) * g.res : u(rng) * scale * g.res * 5);
    std::vector<unsigned> stream;
    te::PolygonTable tb;
    memset((void*)&tb, 0, sizeof(tb));
    if (!te::build_polygon_table(g, n, off, stream, tb)) { ++fallbacks; continue; }
    ++tables;
    // walk the stream like the kernel does and check every index
    size_t p = tb.first;
    if ((size_t)(255 + tb.di_span) * tb.dj_span * 8 > 65536) { printf("tile too large\n"); return 1; }
    int unc = 0;
    for (int r = 0; r < tb.n_rows; ++r) {
      if (p >= stream.size()) { printf("stream overrun\n"); return 1; }
      const unsigned hdr = stream[p++];
      const int di_idx = hdr & 0xff, items = hdr >> 8;
      if (di_idx >= tb.di_span || items <= 0) { printf("bad row\n"); return 1; }
      int last = -1;
      for (int k = 0; k < items; ++k) {
        if (p >= stream.size()) { printf("stream overrun\n"); return 1; }
        const unsigned item = stream[p++];
        const int dj_idx = item & 0xff, len = (item >> 8) & 0xff;
        if (len < 1 || dj_idx <= last || dj_idx + len > tb.dj_span) { printf("bad item\n"); return 1; }
        last = dj_idx + len - 1;
        unc += item >> 31;
      }
    }
    if (p != stream.size() || unc != tb.n_uncertain) { printf("stream length / uncertain count\n"); return 1; }
  }
  printf("path polygons=%ld tables=%ld fallbacks=%ld\n", polys, tables, fallbacks);
  return 0;
}

int main() { return check_builders() || check_table_classification(); }
