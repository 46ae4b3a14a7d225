// plugin_chain_test.cpp -- exercises the drop-in plugin adapters the way filters::FilterChain would
// (configure from a parameter map, then update(in, out) plugin after plugin) and checks every output
// layer against the CPU oracle.  TEST ONLY: this is the single place where the oracle is linked.
//
//   plugin_chain_test --no-device   configure()/validation behaviour, graceful failure without a GPU
//   plugin_chain_test --device      full parity of Slope/Step/Roughness plugins and FusedChainFilter
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include <filters/filter_base.h>
#include <grid_map_core/GridMap.hpp>
#include <pluginlib/class_list_macros.h>

#include "te_oracle.h"

typedef filters::FilterBase<grid_map::GridMap> Filter;
using filters::ParamMap;

static int g_fail = 0;
#define CHECK(cond)                                                         \
  do {                                                                      \
    if (!(cond)) {                                                          \
      std::fprintf(stderr, "CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      ++g_fail;                                                             \
    }                                                                       \
  } while (0)

static std::unique_ptr<Filter> make(const std::string& type) {
  auto it = pluginlib_stub::registry().find(type);
  if (it == pluginlib_stub::registry().end()) {
    std::fprintf(stderr, "class %s not exported\n", type.c_str());
    ++g_fail;
    return nullptr;
  }
  return std::unique_ptr<Filter>(static_cast<Filter*>(it->second()));
}

static const char* kSlope = "filters::SlopeFilter<grid_map::GridMap>";
static const char* kStep = "filters::StepFilter<grid_map::GridMap>";
static const char* kRough = "filters::RoughnessFilter<grid_map::GridMap>";
static const char* kFused = "filters::FusedChainFilter<grid_map::GridMap>";

static void test_configure() {
  // same acceptance rules as the reference's configure()s
  auto s = make(kSlope);
  CHECK(s && s->configure("slopeFilter", ParamMap{{"critical_value", 1.0}, {"map_type", "traversability_slope"}}));
  CHECK(s && !s->configure("slopeFilter", ParamMap{{"map_type", "traversability_slope"}}));              // missing
  CHECK(s && !s->configure("slopeFilter", ParamMap{{"critical_value", 2.0}, {"map_type", "x"}}));         // > pi/2
  CHECK(s && !s->configure("slopeFilter", ParamMap{{"critical_value", -0.1}, {"map_type", "x"}}));
  CHECK(s && !s->configure("slopeFilter", ParamMap{{"critical_value", 1.0}}));                           // no map_type
  auto t = make(kStep);
  ParamMap good{{"critical_value", 0.12}, {"first_window_radius", 0.04}, {"second_window_radius", 0.04},
                {"critical_cell_number", 4}, {"map_type", "traversability_step"}};
  CHECK(t && t->configure("stepFilter", good));
  ParamMap bad = good;
  bad["critical_cell_number"] = 0;
  CHECK(t && !t->configure("stepFilter", bad));
  bad = good;
  bad["first_window_radius"] = -1.0;
  CHECK(t && !t->configure("stepFilter", bad));
  bad = good;
  bad.erase("second_window_radius");
  CHECK(t && !t->configure("stepFilter", bad));
  auto r = make(kRough);
  CHECK(r && r->configure("roughnessFilter", ParamMap{{"critical_value", 0.05}, {"estimation_radius", 0.05}, {"map_type", "traversability_roughness"}}));
  CHECK(r && !r->configure("roughnessFilter", ParamMap{{"critical_value", -1.0}, {"estimation_radius", 0.05}, {"map_type", "x"}}));
  CHECK(r && !r->configure("roughnessFilter", ParamMap{{"critical_value", 0.05}, {"map_type", "x"}}));
  auto f = make(kFused);
  CHECK(f && f->configure("fused", ParamMap{}));
  CHECK(f && !f->configure("fused", ParamMap{{"slope_critical_value", 3.0}}));
}

static grid_map::GridMap make_map(int rows, int cols, double res) {
  grid_map::GridMap m;
  m.setGeometry(grid_map::Vec2d{{rows * res, cols * res}}, res, grid_map::Vec2d{{1.25, -0.5}});
  m.add("elevation");
  grid_map::Matrix& e = m["elevation"];
  for (int j = 0; j < cols; ++j)
    for (int i = 0; i < rows; ++i) {
      double z = 0.15 * std::sin(0.11 * i) * std::cos(0.07 * j) + 0.002 * ((i * 131 + j * 71) % 17);
      if (i > 40 && i < 60 && j > 30 && j < 50) z += 0.25;  // a box
      e(i, j) = (float)z;
    }
  e(10, 10) = std::nanf("");
  e(11, 10) = std::nanf("");
  e(100, 90) = std::nanf("");
  return m;
}

static int compare(const char* name, const grid_map::Matrix& got, const std::vector<float>& want) {
  int bad = 0;
  double mx = 0;
  const size_t n = (size_t)got.rows() * got.cols();
  for (size_t k = 0; k < n; ++k) {
    const float a = got.data()[k], b = want[k];
    if (std::isnan(a) != std::isnan(b)) {
      ++bad;
      continue;
    }
    if (std::isnan(a)) continue;
    const double d = std::fabs((double)a - (double)b);
    if (d > mx) mx = d;
    if (d > 1e-5) ++bad;
  }
  std::printf("  %-28s mismatches=%d max|d|=%.3g\n", name, bad, mx);
  return bad;
}

static void test_device() {
  const int rows = 150, cols = 120;
  const double res = 0.04;
  grid_map::GridMap map0 = make_map(rows, cols, res);
  const size_t n = (size_t)rows * cols;
  teo_geom g;
  teo_geom_init(&g, rows, cols, res, 1.25, -0.5);
  teo_params p;
  teo_params_default(&p);
  p.normals_radius = 0.09;
  p.rough_radius = 0.13;
  p.step_radius1 = 0.1;
  p.step_radius2 = 0.07;
  std::vector<float> nx(n), ny(n), nz(n), sl(n), st(n), ro(n), tr(n);
  const float* elev = map0["elevation"].data();
  teo_chain(&g, &p, elev, sl.data(), st.data(), ro.data(), tr.data(), nx.data(), ny.data(), nz.data());
  // gridMapFilters/NormalVectorsFilter runs upstream of the three plugins (on the host): hand its layers in
  map0.add("surface_normal_x");
  map0.add("surface_normal_y");
  map0.add("surface_normal_z");
  std::memcpy(map0["surface_normal_x"].data(), nx.data(), n * 4);
  std::memcpy(map0["surface_normal_y"].data(), ny.data(), n * 4);
  std::memcpy(map0["surface_normal_z"].data(), nz.data(), n * 4);

  auto s = make(kSlope), t = make(kStep), r = make(kRough);
  CHECK(s->configure("slopeFilter", ParamMap{{"critical_value", p.slope_critical}, {"map_type", "traversability_slope"}}));
  CHECK(t->configure("stepFilter", ParamMap{{"critical_value", p.step_critical}, {"first_window_radius", p.step_radius1},
                                            {"second_window_radius", p.step_radius2}, {"critical_cell_number", p.step_ncrit},
                                            {"map_type", "traversability_step"}}));
  CHECK(r->configure("roughnessFilter", ParamMap{{"critical_value", p.rough_critical}, {"estimation_radius", p.rough_radius},
                                                 {"map_type", "traversability_roughness"}}));
  grid_map::GridMap m1, m2, m3;
  CHECK(s->update(map0, m1));
  CHECK(t->update(m1, m2));
  CHECK(r->update(m2, m3));
  std::printf("drop-in plugins (Slope -> Step -> Roughness):\n");
  CHECK(compare("traversability_slope", m3["traversability_slope"], sl) == 0);
  CHECK(compare("traversability_step", m3["traversability_step"], st) == 0);
  CHECK(compare("traversability_roughness", m3["traversability_roughness"], ro) == 0);
  CHECK(!m3.exists("step_height"));                                   // StepFilter.cpp:180
  CHECK(m3.exists("elevation") && m3.exists("surface_normal_x"));     // mapOut = mapIn keeps every layer
  // a missing input layer makes update() return false (the reference would throw out of GridMap::at)
  grid_map::GridMap bare = make_map(rows, cols, res), out;
  CHECK(!s->update(bare, out));

  auto f = make(kFused);
  CHECK(f->configure("fused", ParamMap{{"normals_radius", p.normals_radius}, {"estimation_radius", p.rough_radius},
                                       {"first_window_radius", p.step_radius1}, {"second_window_radius", p.step_radius2}}));
  grid_map::GridMap fin = make_map(rows, cols, res), fout;
  CHECK(f->update(fin, fout));
  std::printf("FusedChainFilter:\n");
  CHECK(compare("traversability_slope", fout["traversability_slope"], sl) == 0);
  CHECK(compare("traversability_step", fout["traversability_step"], st) == 0);
  CHECK(compare("traversability_roughness", fout["traversability_roughness"], ro) == 0);
  CHECK(compare("traversability", fout["traversability"], tr) == 0);
  CHECK(!fout.exists("surface_normal_x"));                             // DeletionFilter
}

// the same map as a circular buffer: logical (i, j) stored at ((i + si) % rows, (j + sj) % cols)
static grid_map::GridMap rolled(const grid_map::GridMap& src, int si, int sj) {
  const int rows = src.getSize()(0), cols = src.getSize()(1);
  grid_map::GridMap m;
  m.setGeometry(src.getLength(), src.getResolution(), src.getPosition());
  m.setStartIndex(grid_map::Arr2i{{si, sj}});
  for (const std::string& name : src.getLayers()) {
    m.add(name);
    for (int j = 0; j < cols; ++j)
      for (int i = 0; i < rows; ++i) m[name]((i + si) % rows, (j + sj) % cols) = src.get(name)(i, j);
  }
  return m;
}

static void test_device_circular() {
  // a map that has been move()d: the reference's iterators hide the start index (StepFilter.cpp:112,124)
  const int rows = 150, cols = 120, si = 37, sj = 101;
  const double res = 0.04;
  grid_map::GridMap flat = make_map(rows, cols, res);
  auto f = make(kFused);
  CHECK(f->configure("fused", ParamMap{{"normals_radius", 0.09}, {"estimation_radius", 0.13},
                                       {"first_window_radius", 0.1}, {"second_window_radius", 0.07}}));
  grid_map::GridMap want, got, in = rolled(flat, si, sj);
  CHECK(f->update(flat, want));
  CHECK(f->update(in, got));
  CHECK(got.getStartIndex()(0) == si && got.getStartIndex()(1) == sj);
  std::printf("FusedChainFilter on a circular-buffer map (start index %d,%d):\n", si, sj);
  grid_map::GridMap expect = rolled(want, si, sj);
  const size_t n = (size_t)rows * cols;
  for (const char* name : {"traversability_slope", "traversability_step", "traversability_roughness", "traversability"}) {
    std::vector<float> w(expect[name].data(), expect[name].data() + n);
    CHECK(compare(name, got[name], w) == 0);
  }
  // the single plugins take the same route
  auto t = make(kStep);
  CHECK(t->configure("stepFilter", ParamMap{{"critical_value", 0.12}, {"first_window_radius", 0.1}, {"second_window_radius", 0.07},
                                            {"critical_cell_number", 4}, {"map_type", "traversability_step"}}));
  grid_map::GridMap tw, tg;
  CHECK(t->update(flat, tw));
  CHECK(t->update(in, tg));
  grid_map::GridMap te = rolled(tw, si, sj);
  std::vector<float> w(te["traversability_step"].data(), te["traversability_step"].data() + n);
  CHECK(compare("traversability_step (plugin)", tg["traversability_step"], w) == 0);
}

static void test_no_device() {
  auto t = make(kStep);
  CHECK(t->configure("stepFilter", ParamMap{{"critical_value", 0.12}, {"first_window_radius", 0.04},
                                            {"second_window_radius", 0.04}, {"critical_cell_number", 4},
                                            {"map_type", "traversability_step"}}));
  grid_map::GridMap in = make_map(40, 30, 0.05), out;
  CHECK(!t->update(in, out));  // no GPU: a clean `false` (the chain is marked failed), never a CPU fallback
}

int main(int argc, char** argv) {
  const bool device = argc > 1 && std::strcmp(argv[1], "--device") == 0;
  test_configure();
  if (device) {
    test_device();
    test_device_circular();
  } else {
    test_no_device();
  }
  std::printf("%s (%d failures)\n", g_fail ? "FAILED" : "OK", g_fail);
  return g_fail ? 1 : 0;
}
