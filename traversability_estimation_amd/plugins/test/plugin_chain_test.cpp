// plugin_chain_test.cpp -- exercises the drop-in plugin adapters the way filters::FilterChain would
// (configure from a parameter map, then update(in, out) plugin after plugin) and checks every output
// layer against the CPU oracle.  TEST ONLY: this is the single place where the oracle is linked.
//
//   plugin_chain_test --no-device   configure()/validation behaviour, graceful failure without a GPU
//   plugin_chain_test --device      full parity of Slope/Step/Roughness plugins and FusedChainFilter
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include <filters/filter_base.h>
#include <grid_map_core/GridMap.hpp>
#include <pluginlib/class_list_macros.h>

#include "te_oracle.h"
#include "travgpu_plugins/DeviceMap.hpp"
#include "traversability_estimation_gpu/TraversabilityMap.hpp"

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
static const char* kNormals = "filters::SurfaceNormalsFilter<grid_map::GridMap>";

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
  auto nf = make(kNormals);
  CHECK(!nf->configure("normals", ParamMap{}));                                                    // radius is required
  CHECK(!nf->configure("normals", ParamMap{{"radius", -0.1}}));
  CHECK(!nf->configure("normals", ParamMap{{"radius", 0.05}, {"normal_vector_positive_axis", "w"}}));
  CHECK(nf->configure("normals", ParamMap{{"input_layer", "elevation"}, {"output_layers_prefix", "surface_normal_"}, {"radius", 0.05},
                                          {"normal_vector_positive_axis", "z"}}));                 // robot_filter_parameter.yaml:3-9
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
  map0.setTimestamp(1529564943122772932ull);
  travgpu_plugins::DeviceMap& dev = travgpu_plugins::DeviceMap::instance();
  const unsigned long up0 = dev.uploads(), sk0 = dev.uploadsSkipped();
  CHECK(s->update(map0, m1));
  CHECK(t->update(m1, m2));
  CHECK(r->update(m2, m3));
  // every plugin gets a deep copy of the whole map, but a layer crosses PCIe once: surface_normal_z for SlopeFilter,
  // elevation for StepFilter, surface_normal_x / _y for RoughnessFilter (its elevation and _z are already resident)
  CHECK(dev.uploads() - up0 == 4);
  CHECK(dev.uploadsSkipped() - sk0 == 2);
  {  // the same map again (the node re-filters on a parameter update): nothing is uploaded
    grid_map::GridMap a, b, c;
    CHECK(s->update(map0, a) && t->update(a, b) && r->update(b, c));
    CHECK(dev.uploads() - up0 == 4);
    CHECK(dev.uploadsSkipped() - sk0 == 8);
    CHECK(compare("traversability_roughness (resident inputs)", c["traversability_roughness"], ro) == 0);
    // a new elevation map (new stamp, new content) is uploaded again
    grid_map::GridMap next = map0;
    next.setTimestamp(map0.getTimestamp() + 250000000ull);
    next["elevation"](7, 9) += 0.25f;
    // (StepFilter also sends the normals RoughnessFilter will read next, when the map brings them under a new stamp)
    const bool prefetching = !(getenv("TRAVGPU_PLUGIN_PREFETCH") && atoi(getenv("TRAVGPU_PLUGIN_PREFETCH")) == 0);
    const unsigned long nrm = (prefetching && map0.exists("surface_normal_x")) ? 3 : 0;
    CHECK(t->update(next, b));
    CHECK(dev.uploads() - up0 == 5 + nrm);
    CHECK(t->update(map0, b));  // and back: the device layer is identified, not assumed
    CHECK(dev.uploads() - up0 == 6 + 2 * nrm);
    // one cell edited in place under the SAME stamp (inpainting, a local update): it must be uploaded, not taken for
    // the resident layer, wherever the cell lies
    grid_map::GridMap edited = map0;
    edited["elevation"](rows - 2, cols / 2 + 1) += 0.5f;
    grid_map::GridMap e1, e2;
    CHECK(t->update(edited, e1));
    CHECK(dev.uploads() - up0 == 7 + 2 * nrm);
    CHECK(t->update(map0, e2));
    CHECK(dev.uploads() - up0 == 8 + 2 * nrm);
    {
      const grid_map::Matrix &a = e1["traversability_step"], &b = e2["traversability_step"];
      size_t differ = 0;
      for (size_t k = 0; k < (size_t)a.rows() * a.cols(); ++k)
        if (memcmp(a.data() + k, b.data() + k, 4) != 0) ++differ;
      CHECK(differ > 0);  // the edited cell reached the device
    }
  }
  std::printf("drop-in plugins (Slope -> Step -> Roughness):\n");
  CHECK(compare("traversability_slope", m3["traversability_slope"], sl) == 0);
  CHECK(compare("traversability_step", m3["traversability_step"], st) == 0);
  CHECK(compare("traversability_roughness", m3["traversability_roughness"], ro) == 0);
  CHECK(!m3.exists("step_height"));                                   // StepFilter.cpp:180
  CHECK(m3.exists("elevation") && m3.exists("surface_normal_x"));     // mapOut = mapIn keeps every layer
  // a missing input layer makes update() return false (the reference would throw out of GridMap::at)
  grid_map::GridMap bare = make_map(rows, cols, res), out;
  CHECK(!s->update(bare, out));

  {  // the whole unchanged-parameter chain on the device: SurfaceNormalsFilter in place of the host NormalVectorsFilter.
     // One upload (elevation) serves all four plugins: the normals the first one downloads are what the device holds.
    auto nf = make(kNormals);
    CHECK(nf->configure("normals", ParamMap{{"radius", p.normals_radius}, {"normal_vector_positive_axis", "z"}}));
    grid_map::GridMap fresh = make_map(rows, cols, res), n1, n2, n3, n4;
    fresh.setTimestamp(map0.getTimestamp() + 1000000000ull);
    const unsigned long up1 = dev.uploads();
    CHECK(nf->update(fresh, n1));
    CHECK(s->update(n1, n2) && t->update(n2, n3) && r->update(n3, n4));
    CHECK(dev.uploads() - up1 == 1);
    std::printf("SurfaceNormalsFilter -> Slope -> Step -> Roughness:\n");
    CHECK(compare("surface_normal_x", n4["surface_normal_x"], nx) == 0);
    CHECK(compare("surface_normal_y", n4["surface_normal_y"], ny) == 0);
    CHECK(compare("surface_normal_z", n4["surface_normal_z"], nz) == 0);
    CHECK(compare("traversability_slope", n4["traversability_slope"], sl) == 0);
    CHECK(compare("traversability_step", n4["traversability_step"], st) == 0);
    CHECK(compare("traversability_roughness", n4["traversability_roughness"], ro) == 0);
  }
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

// the device-backed TraversabilityMap against the oracle: chain, both footprint passes, path checks
static void test_traversability_map() {
  using traversability_estimation_gpu::TraversabilityMap;
  const int rows = 150, cols = 120, si = 61, sj = 7;
  const double res = 0.04;
  grid_map::GridMap flat = make_map(rows, cols, res);
  const size_t n = (size_t)rows * cols;
  teo_geom g;
  teo_geom_init(&g, rows, cols, res, 1.25, -0.5);
  teo_params p;
  teo_params_default(&p);
  p.normals_radius = 0.09;
  p.rough_radius = 0.13;
  p.step_radius1 = 0.1;
  p.step_radius2 = 0.07;
  p.fp_radius = 0.2;
  p.fp_offset = 0.1;
  std::vector<float> sl(n), st(n), ro(n), tr(n), fp(n), fx(n), fr(n);
  const float* elev = flat["elevation"].data();
  teo_chain(&g, &p, elev, sl.data(), st.data(), ro.data(), tr.data(), nullptr, nullptr, nullptr);
  teo_footprint(&g, &p, elev, sl.data(), st.data(), ro.data(), tr.data(), fp.data(), nullptr, nullptr, nullptr);
  const double foot[8] = {0.45, 0.30, 0.45, -0.30, -0.45, -0.30, -0.45, 0.30};
  const double yaw = 1.5707963267948966;
  teo_polygon_footprint(&g, &p, elev, sl.data(), st.data(), ro.data(), tr.data(), 4, foot, yaw, fx.data(), fr.data());

  TraversabilityMap tm;
  te_params tp = tm.getParameters();
  tp.normals_radius = p.normals_radius;
  tp.rough_radius = p.rough_radius;
  tp.step_radius1 = p.step_radius1;
  tp.step_radius2 = p.step_radius2;
  CHECK(tm.setParameters(tp));
  traversability_msgs::TraversabilityResult r1;
  traversability_msgs::FootprintPath one;
  one.poses.poses.resize(1);
  one.radius = 0.2;
  CHECK(tm.checkFootprintPath(one, r1) && !r1.is_safe);  // map not initialised: true, unsafe (:323-327)
  CHECK(!tm.computeTraversability());                    // no elevation map yet (:228-231)
  CHECK(!tm.traversabilityFootprint(0.2, 0.1) && !tm.traversabilityFootprint(yaw));
  grid_map::GridMap bare;
  bare.setGeometry(flat.getLength(), res, flat.getPosition());
  CHECK(!tm.setElevationMap(bare));  // no "elevation" layer (:145-150)
  grid_map::GridMap in = rolled(flat, si, sj);
  CHECK(tm.setElevationMap(in));
  CHECK(tm.computeTraversability());
  CHECK(tm.traversabilityFootprint(0.2, 0.1));
  std::vector<geometry_msgs::Point32> pts(4);
  for (int k = 0; k < 4; ++k) {
    pts[k].x = (float)foot[2 * k];
    pts[k].y = (float)foot[2 * k + 1];
  }
  // Point32 is float: the oracle gets the same rounded values
  double footf[8], foot3[12];
  for (int k = 0; k < 4; ++k) {
    footf[2 * k] = pts[k].x;
    footf[2 * k + 1] = pts[k].y;
    foot3[3 * k] = pts[k].x;
    foot3[3 * k + 1] = pts[k].y;
    foot3[3 * k + 2] = 0.0;
  }
  teo_polygon_footprint(&g, &p, elev, sl.data(), st.data(), ro.data(), tr.data(), 4, footf, yaw, fx.data(), fr.data());
  tm.setFootprintPolygon(pts);
  CHECK(tm.traversabilityFootprint(yaw));
  grid_map::GridMap out = tm.getTraversabilityMap();
  CHECK(out.getStartIndex()(0) == si && out.getStartIndex()(1) == sj);
  std::printf("TraversabilityMap on a circular-buffer map (start index %d,%d):\n", si, sj);
  grid_map::GridMap want;
  want.setGeometry(flat.getLength(), res, flat.getPosition());
  struct Named {
    const char* name;
    const std::vector<float>* v;
  };
  const Named layers[] = {{"traversability_slope", &sl}, {"traversability_step", &st}, {"traversability_roughness", &ro},
                          {"traversability", &tr},       {"traversability_footprint", &fp}, {"traversability_x", &fx},
                          {"traversability_rot", &fr}};
  for (const Named& l : layers) {
    want.add(l.name);
    std::memcpy(want[l.name].data(), l.v->data(), n * 4);
  }
  grid_map::GridMap expect = rolled(want, si, sj);
  for (const Named& l : layers) {
    CHECK(out.exists(l.name));
    if (!out.exists(l.name)) continue;
    std::vector<float> w(expect[l.name].data(), expect[l.name].data() + n);
    CHECK(compare(l.name, out[l.name], w) == 0);
  }

  // a CheckFootprintPath request: circular footprints of two radii and polygonal ones, mixed
  std::vector<traversability_msgs::FootprintPath> paths;
  unsigned seed = 12345;
  auto rnd = [&]() {
    seed = seed * 1664525u + 1013904223u;
    return (double)(seed >> 8) / (double)(1u << 24);
  };
  for (int k = 0; k < 90; ++k) {
    traversability_msgs::FootprintPath path;
    const int np = 1 + (int)(rnd() * 4.0);
    double x = 1.25 + (rnd() - 0.5) * rows * res * 0.9, y = -0.5 + (rnd() - 0.5) * cols * res * 0.9;
    for (int m = 0; m < np; ++m) {
      geometry_msgs::Pose pose;
      pose.position.x = x;
      pose.position.y = y;
      const double a = rnd() * 6.0;
      pose.orientation.z = std::sin(a / 2);
      pose.orientation.w = std::cos(a / 2);
      path.poses.poses.push_back(pose);
      x += (rnd() - 0.5) * 0.8;
      y += (rnd() - 0.5) * 0.8;
    }
    if (k % 3 == 0) {
      path.footprint.polygon.points = pts;
      path.conservative = (k % 2) ? 1 : 0;
    } else {
      path.radius = (k % 3 == 1) ? 0.2 : 0.3;
    }
    paths.push_back(path);
  }
  std::vector<traversability_msgs::TraversabilityResult> results;
  CHECK(tm.checkFootprintPaths(paths, results));
  CHECK(results.size() == paths.size());
  int n_safe = 0, n_bad = 0;
  for (size_t k = 0; k < paths.size() && k < results.size(); ++k) {
    const int np = (int)paths[k].poses.poses.size();
    const int off[2] = {0, np};
    unsigned char safe = 0;
    double trav = 0, area = 0;
    int status = 0;
    if (paths[k].footprint.polygon.points.empty()) {
      teo_params q = p;
      q.fp_radius = paths[k].radius;
      q.fp_offset = 0.15;
      std::vector<float> layer(n);
      teo_footprint(&g, &q, elev, sl.data(), st.data(), ro.data(), tr.data(), layer.data(), nullptr, nullptr, nullptr);
      std::vector<double> xy;
      for (const auto& pose : paths[k].poses.poses) {
        xy.push_back(pose.position.x);
        xy.push_back(pose.position.y);
      }
      teo_check_circular_paths(&g, layer.data(), q.fp_default, 1, off, xy.data(), &safe, &trav, &status);
    } else {
      std::vector<double> poses;
      for (const auto& pose : paths[k].poses.poses) {
        const double v[7] = {pose.position.x,    pose.position.y,    pose.position.z,   pose.orientation.x,
                             pose.orientation.y, pose.orientation.z, pose.orientation.w};
        poses.insert(poses.end(), v, v + 7);
      }
      teo_check_polygon_paths(&g, &p, elev, sl.data(), st.data(), ro.data(), tr.data(), 1, off, poses.data(), 4, foot3,
                              &paths[k].conservative, &safe, &trav, &area, &status);
    }
    // (the traversability of a circular path is a mean of footprint values, which the fixed-point footprint kernel
    // delivers within 1e-6 of the double sum, not bit for bit; the decision and the area are exact)
    if (results[k].is_safe != safe || std::fabs(results[k].traversability - trav) > 1e-5 || results[k].area != area) {
      ++n_bad;
      std::fprintf(stderr, "path %zu: got (%d, %.17g, %.17g) want (%d, %.17g, %.17g)\n", k, results[k].is_safe,
                   results[k].traversability, results[k].area, safe, trav, area);
    }
    n_safe += safe;
  }
  std::printf("  checkFootprintPaths: %zu paths, %d safe, %d mismatches\n", paths.size(), n_safe, n_bad);
  CHECK(n_bad == 0);
  CHECK(n_safe > 5 && n_safe < (int)paths.size() - 5);
  // footprint/check_robot_inclination (:114): the layer robot_slope travels with the elevation map
  {
    std::vector<float> rs(n, 1.0f);
    for (size_t k = 0; k < n; ++k) {
      const double u = rnd();
      if (u < 0.004) rs[k] = 0.0f;
      else if (u < 0.05) rs[k] = NAN;
    }
    grid_map::GridMap flat2 = flat;
    flat2.add("robot_slope");
    std::memcpy(flat2["robot_slope"].data(), rs.data(), n * 4);
    CHECK(tm.setCheckRobotInclination(true));
    CHECK(!tm.checkFootprintPaths(paths, results));  // no such layer on the device yet: refused, not skipped
    CHECK(tm.setElevationMap(rolled(flat2, si, sj)) && tm.computeTraversability());
    CHECK(tm.checkFootprintPaths(paths, results) && results.size() == paths.size());
    int n_safe_incl = 0, n_bad_incl = 0;
    for (size_t k = 0; k < paths.size() && k < results.size(); ++k) {
      const int np = (int)paths[k].poses.poses.size();
      const int off[2] = {0, np};
      unsigned char safe = 0;
      double trav = 0, area = 0;
      int status = 0;
      if (paths[k].footprint.polygon.points.empty()) {
        teo_params q = p;
        q.fp_radius = paths[k].radius;
        q.fp_offset = 0.15;
        std::vector<float> layer(n);
        teo_footprint(&g, &q, elev, sl.data(), st.data(), ro.data(), tr.data(), layer.data(), nullptr, nullptr, nullptr);
        std::vector<double> xy;
        for (const auto& pose : paths[k].poses.poses) {
          xy.push_back(pose.position.x);
          xy.push_back(pose.position.y);
        }
        teo_check_circular_paths_incl(&g, layer.data(), q.fp_default, rs.data(), 1, off, xy.data(), &safe, &trav, &status);
      } else {
        std::vector<double> poses;
        for (const auto& pose : paths[k].poses.poses) {
          const double v[7] = {pose.position.x,    pose.position.y,    pose.position.z,   pose.orientation.x,
                               pose.orientation.y, pose.orientation.z, pose.orientation.w};
          poses.insert(poses.end(), v, v + 7);
        }
        teo_check_polygon_paths_incl(&g, &p, elev, sl.data(), st.data(), ro.data(), tr.data(), rs.data(), 1, off, poses.data(),
                                     4, foot3, &paths[k].conservative, &safe, &trav, &area, &status);
      }
      if (results[k].is_safe != safe || std::fabs(results[k].traversability - trav) > 1e-5 || results[k].area != area) ++n_bad_incl;
      n_safe_incl += safe;
    }
    std::printf("  checkFootprintPaths with check_robot_inclination: %d safe (of %d without), %d mismatches\n", n_safe_incl,
                n_safe, n_bad_incl);
    CHECK(n_bad_incl == 0);
    CHECK(n_safe_incl > 0 && n_safe_incl < n_safe);
    // the next elevation map comes WITHOUT the layer (same geometry): the previous map's inclinations must not be used
    CHECK(tm.setElevationMap(rolled(flat, si, sj)) && tm.computeTraversability());
    CHECK(!tm.checkFootprintPaths(paths, results));
    CHECK(tm.setCheckRobotInclination(false));
    CHECK(tm.checkFootprintPaths(paths, results) && results.size() == paths.size());
  }
  // a request with a path without poses stops there (TraversabilityEstimation.cpp:290)
  paths[4].poses.poses.clear();
  CHECK(!tm.checkFootprintPaths(paths, results) && results.size() == 4);
  CHECK(!tm.checkFootprintPath(paths[4], r1) && !r1.is_safe);
  paths.clear();
  CHECK(!tm.checkFootprintPaths(paths, results));
}

static void test_no_device() {
  auto t = make(kStep);
  CHECK(t->configure("stepFilter", ParamMap{{"critical_value", 0.12}, {"first_window_radius", 0.04},
                                            {"second_window_radius", 0.04}, {"critical_cell_number", 4},
                                            {"map_type", "traversability_step"}}));
  grid_map::GridMap in = make_map(40, 30, 0.05), out;
  CHECK(!t->update(in, out));  // no GPU: a clean `false` (the chain is marked failed), never a CPU fallback
  traversability_estimation_gpu::TraversabilityMap tm;  // same for the device-backed map: every entry point refuses
  CHECK(!tm.setElevationMap(in) && !tm.computeTraversability() && !tm.traversabilityFootprint(0.3, 0.15));
  CHECK(!tm.error().empty());
}

int main(int argc, char** argv) {
  const bool device = argc > 1 && std::strcmp(argv[1], "--device") == 0;
  test_configure();
  if (device) {
    test_device();
    test_device_circular();
    test_traversability_map();
  } else {
    test_no_device();
  }
  std::printf("%s (%d failures)\n", g_fail ? "FAILED" : "OK", g_fail);
  return g_fail ? 1 : 0;
}
