// te_paths_api.hip -- C-ABI of libtravgpu.so, SURVEY.md 8(f) N2 / N3: batched checkFootprintPath (circular and polygonal,
// TraversabilityMap.cpp:320-645), checkInclination (:748-762), the polygon footprint layers (:239-305).  Kernels:
// te_paths.hip, te_polygon.hip.  The context and the helpers shared with the other parts: te_ctx.h.
#include "te_ctx.h"

using namespace te;
using namespace te::shim;

extern "C" {

int te_check_footprint_paths(te_ctx* c, int map, int n_paths, const int* pose_offset, const double* pose_xy,
                             unsigned char* is_safe, double* traversability, int* status) {
  if (!c || n_paths < 0 || (n_paths > 0 && (!pose_offset || !pose_xy || !is_safe || !traversability || !status)))
    return fail(TE_ERR_INVALID_ARG, "te_check_footprint_paths: NULL argument");
  CtxLock lk(c);
  if (!c->have_geo || !c->footprint_done)
    return fail(TE_ERR_NOT_READY, "te_check_footprint_paths: run the chain with the footprint pass first");
  if (map < 0 || map >= c->geo.batch) return fail(TE_ERR_INVALID_ARG, "te_check_footprint_paths: map %d of %d", map, c->geo.batch);
  if (c->check_inclination && !c->have_robot_slope)
    return fail(TE_ERR_NOT_READY, "te_check_footprint_paths: check_robot_inclination is set but the layer robot_slope was never uploaded");
  if (n_paths == 0) return TE_OK;
  const int n_poses = pose_offset[n_paths];
  if (pose_offset[0] != 0 || n_poses < 0) return fail(TE_ERR_INVALID_ARG, "te_check_footprint_paths: bad pose offsets");
  for (int k = 0; k < n_paths; ++k)
    if (pose_offset[k + 1] < pose_offset[k]) return fail(TE_ERR_INVALID_ARG, "te_check_footprint_paths: bad pose offsets");
  HIP_TRY(hipSetDevice(c->device));
  // staging buffers for this call (paths are small: a few KB .. MB)
  const size_t b_off = (size_t)(n_paths + 1) * sizeof(int), b_xy = (size_t)2 * (n_poses > 0 ? n_poses : 1) * sizeof(double);
  const size_t b_safe = (size_t)n_paths, b_trav = (size_t)n_paths * sizeof(double), b_st = (size_t)n_paths * sizeof(int);
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  char* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, up(b_off) + up(b_xy) + up(b_trav) + up(b_st) + up(b_safe)));
  int* d_off = (int*)d;
  double* d_xy = (double*)(d + up(b_off));
  double* d_trav = (double*)(d + up(b_off) + up(b_xy));
  int* d_st = (int*)(d + up(b_off) + up(b_xy) + up(b_trav));
  unsigned char* d_safe = (unsigned char*)(d + up(b_off) + up(b_xy) + up(b_trav) + up(b_st));
  hipError_t e = hipMemcpyAsync(d_off, pose_offset, b_off, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess && n_poses > 0) e = hipMemcpyAsync(d_xy, pose_xy, (size_t)2 * n_poses * sizeof(double), hipMemcpyHostToDevice, c->stream);
  const size_t per = (size_t)c->geo.rows * c->geo.cols;
  if (e == hipSuccess)
    e = launch_check_circular_paths(c->geo, c->L.footprint + per * map, c->params.fp_default,
                                    c->check_inclination ? c->robot_slope + per * map : nullptr, n_paths, d_off, d_xy, d_safe,
                                    d_trav, d_st, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(is_safe, d_safe, b_safe, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(traversability, d_trav, b_trav, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(status, d_st, b_st, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(TE_ERR_HIP, "te_check_footprint_paths: %s", hipGetErrorString(e));
  return TE_OK;
}

int te_set_check_robot_inclination(te_ctx* c, int enabled) {
  if (!c) return fail(TE_ERR_INVALID_ARG, "te_set_check_robot_inclination: NULL");
  CtxLock lk(c);
  c->check_inclination = enabled != 0;
  return TE_OK;
}

namespace {
// batched checkInclination on the resident robot_slope layer of map `map`; c->mu held
int check_inclination_locked(te_ctx* c, int map, int n, const double* start_end_xy, unsigned char* ok, int* status,
                             const char* who) {
  if (!c->have_geo || !c->have_robot_slope) return fail(TE_ERR_NOT_READY, "%s: upload the layer robot_slope first", who);
  if (map < 0 || map >= c->geo.batch) return fail(TE_ERR_INVALID_ARG, "%s: map %d of %d", who, map, c->geo.batch);
  if (n == 0) return TE_OK;
  HIP_TRY(hipSetDevice(c->device));
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t b_seg = (size_t)4 * n * sizeof(double), b_ok = (size_t)n, b_st = (size_t)n * sizeof(int);
  char* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, up(b_seg) + up(b_st) + up(b_ok)));
  double* d_seg = (double*)d;
  int* d_st = (int*)(d + up(b_seg));
  unsigned char* d_ok = (unsigned char*)(d + up(b_seg) + up(b_st));
  const size_t per = (size_t)c->geo.rows * c->geo.cols;
  hipError_t e = hipMemcpyAsync(d_seg, start_end_xy, b_seg, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = launch_check_inclination(c->geo, c->robot_slope + per * map, n, d_seg, d_ok, d_st, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(ok, d_ok, b_ok, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(status, d_st, b_st, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(TE_ERR_HIP, "%s: %s", who, hipGetErrorString(e));
  return TE_OK;
}
}  // namespace

int te_check_inclination(te_ctx* c, int map, int n_segments, const double* start_end_xy, unsigned char* ok, int* status) {
  if (!c || n_segments < 0 || (n_segments > 0 && (!start_end_xy || !ok || !status)))
    return fail(TE_ERR_INVALID_ARG, "te_check_inclination: NULL argument");
  CtxLock lk(c);
  return check_inclination_locked(c, map, n_segments, start_end_xy, ok, status, "te_check_inclination");
}

int te_run_polygon_footprint(te_ctx* c, int n_points, const double* points_xy, double yaw) {
  if (!c || !points_xy) return fail(TE_ERR_INVALID_ARG, "te_run_polygon_footprint: NULL");
  if (n_points < 1 || n_points > TE_MAX_POLYGON_VERTICES)
    return fail(TE_ERR_INVALID_ARG, "te_run_polygon_footprint: %d footprint points (1..%d)", n_points, TE_MAX_POLYGON_VERTICES);
  if (!isfinite(yaw)) return fail(TE_ERR_INVALID_ARG, "te_run_polygon_footprint: yaw is not finite");
  for (int k = 0; k < 2 * n_points; ++k)
    if (!isfinite(points_xy[k])) return fail(TE_ERR_INVALID_ARG, "te_run_polygon_footprint: footprint point %d is not finite", k / 2);
  CtxLock lk(c);
  if (!c->have_geo || !c->footprint_done)
    return fail(TE_ERR_NOT_READY, "te_run_polygon_footprint: run the chain with the footprint pass first (it marks the untraversable cells)");
  if (c->geo.cols > 65535) return fail(TE_ERR_UNSUPPORTED, "te_run_polygon_footprint: more than 65535 columns");
  HIP_TRY(hipSetDevice(c->device));
  if (!c->poly_x) {
    const size_t lb = (c->layer_elems * sizeof(float) + 255) & ~(size_t)255;
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, 2 * lb);
    if (e != hipSuccess) return fail(TE_ERR_HIP, "te_run_polygon_footprint: hipMalloc(%zu bytes): %s", 2 * lb, hipGetErrorString(e));
    c->poly_x = (float*)p;
    c->poly_rot = (float*)((char*)p + lb);
  }
  PolygonArgs a;
  memset(&a, 0, sizeof(a));
  a.n = n_points;
  a.def = c->params.fp_default;
  rotate_footprint(n_points, points_xy, 0.0, a.off[0]);
  rotate_footprint(n_points, points_xy, yaw, a.off[1]);
  // offset tables (te_polygon.hip); polygons that do not fit the table format, or te_set_option(TE_OPT_POLYGON_PER_CELL) (a debugging
  // aid: both kernels give identical layers), take the kernel that evaluates every cell of every bounding box
  PolygonTables tabs;
  HIP_TRY(hipStreamSynchronize(c->stream));  // the previous call's table upload has been consumed
  c->poly_stream_host.clear();
  bool table = !c->opt_polygon_per_cell;
  for (int w = 0; w < 2 && table; ++w) table = build_polygon_table(c->geo, n_points, a.off[w], c->poly_stream_host, tabs.t[w]);
  if (!table) {
    HIP_TRY(launch_polygon_footprint(c->geo, a, c->L.trav, c->L.untrav, c->poly_x, c->poly_rot, c->stream));
    return TE_OK;
  }
  if (c->poly_stream_host.empty()) c->poly_stream_host.push_back(0);
  if (c->poly_stream_host.size() > c->poly_stream_cap) {
    if (c->poly_stream) (void)hipFree(c->poly_stream);
    c->poly_stream = nullptr;
    c->poly_stream_cap = 0;
    const size_t cap = c->poly_stream_host.size() + 1024;
    hipError_t e = hipMalloc((void**)&c->poly_stream, cap * sizeof(unsigned));
    if (e != hipSuccess) return fail(TE_ERR_HIP, "te_run_polygon_footprint: hipMalloc: %s", hipGetErrorString(e));
    c->poly_stream_cap = cap;
  }
  HIP_TRY(hipMemcpyAsync(c->poly_stream, c->poly_stream_host.data(), c->poly_stream_host.size() * sizeof(unsigned),
                         hipMemcpyHostToDevice, c->stream));
  HIP_TRY(launch_polygon_footprint_table(c->geo, a, tabs, c->poly_stream, c->L.trav, c->L.untrav, c->poly_x, c->poly_rot,
                                         c->stream));
  return TE_OK;
}

namespace {
// isTraversable(polygon) for a batch of validated polygons; context locked, mask present
int polygons_traversable_locked(te_ctx* c, int map, int n_polygons, const int* vertex_offset, const double* vertex_xy,
                                unsigned char* is_traversable, double* traversability, const char* who) {
  if (n_polygons == 0) return TE_OK;
  const int n_vert = vertex_offset[n_polygons];
  HIP_TRY(hipSetDevice(c->device));
  const size_t b_off = (size_t)(n_polygons + 1) * sizeof(int), b_xy = (size_t)2 * n_vert * sizeof(double);
  const size_t b_ok = (size_t)n_polygons, b_trav = (size_t)n_polygons * sizeof(double);
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  char* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, up(b_off) + up(b_xy) + up(b_trav) + up(b_ok)));
  int* d_off = (int*)d;
  double* d_xy = (double*)(d + up(b_off));
  double* d_trav = (double*)(d + up(b_off) + up(b_xy));
  unsigned char* d_ok = (unsigned char*)(d + up(b_off) + up(b_xy) + up(b_trav));
  const size_t per = (size_t)c->geo.rows * c->geo.cols;
  hipError_t e = hipMemcpyAsync(d_off, vertex_offset, b_off, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_xy, vertex_xy, b_xy, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess)
    e = launch_polygons_traversable(c->geo, c->params.fp_default, n_polygons, d_off, d_xy, c->L.trav + per * map,
                                    c->L.untrav + per * map, d_ok, d_trav, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(is_traversable, d_ok, b_ok, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(traversability, d_trav, b_trav, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(TE_ERR_HIP, "%s: %s", who, hipGetErrorString(e));
  return TE_OK;
}
}  // namespace

int te_polygons_traversable(te_ctx* c, int map, int n_polygons, const int* vertex_offset, const double* vertex_xy,
                            unsigned char* is_traversable, double* traversability) {
  if (!c || n_polygons < 0 || (n_polygons > 0 && (!vertex_offset || !vertex_xy || !is_traversable || !traversability)))
    return fail(TE_ERR_INVALID_ARG, "te_polygons_traversable: NULL argument");
  CtxLock lk(c);
  if (!c->have_geo || !c->footprint_done)
    return fail(TE_ERR_NOT_READY, "te_polygons_traversable: run the chain with the footprint pass first (it marks the untraversable cells)");
  if (map < 0 || map >= c->geo.batch) return fail(TE_ERR_INVALID_ARG, "te_polygons_traversable: map %d of %d", map, c->geo.batch);
  if (n_polygons == 0) return TE_OK;
  if (vertex_offset[0] != 0) return fail(TE_ERR_INVALID_ARG, "te_polygons_traversable: bad vertex offsets");
  for (int k = 0; k < n_polygons; ++k)
    if (vertex_offset[k + 1] <= vertex_offset[k])
      return fail(TE_ERR_INVALID_ARG, "te_polygons_traversable: polygon %d has no vertices (or the offsets decrease)", k);
  const int n_vert = vertex_offset[n_polygons];
  for (long k = 0; k < 2L * n_vert; ++k)
    if (!isfinite(vertex_xy[k])) return fail(TE_ERR_INVALID_ARG, "te_polygons_traversable: vertex %ld is not finite", k / 2);
  return polygons_traversable_locked(c, map, n_polygons, vertex_offset, vertex_xy, is_traversable, traversability,
                                     "te_polygons_traversable");
}

int te_polygon_untraversable_hull(te_ctx* c, int map, int n_vertices, const double* vertex_xy, unsigned char* is_traversable,
                                  double* traversability, int cap_vertices, int* n_hull, double* hull_xy) {
  if (!c || !vertex_xy || !is_traversable || !traversability || !n_hull || cap_vertices < 0 || (cap_vertices > 0 && !hull_xy))
    return fail(TE_ERR_INVALID_ARG, "te_polygon_untraversable_hull: NULL argument");
  if (n_vertices < 1) return fail(TE_ERR_INVALID_ARG, "te_polygon_untraversable_hull: a polygon needs at least one vertex");
  for (long k = 0; k < 2L * n_vertices; ++k)
    if (!isfinite(vertex_xy[k])) return fail(TE_ERR_INVALID_ARG, "te_polygon_untraversable_hull: vertex %ld is not finite", k / 2);
  CtxLock lk(c);
  if (!c->have_geo || !c->footprint_done)
    return fail(TE_ERR_NOT_READY, "te_polygon_untraversable_hull: run the chain with the footprint pass first (it marks the untraversable cells)");
  if (map < 0 || map >= c->geo.batch) return fail(TE_ERR_INVALID_ARG, "te_polygon_untraversable_hull: map %d of %d", map, c->geo.batch);
  *n_hull = 0;
  const int off[2] = {0, n_vertices};
  const int rc = polygons_traversable_locked(c, map, 1, off, vertex_xy, is_traversable, traversability, "te_polygon_untraversable_hull");
  if (rc != TE_OK || *is_traversable) return rc;  // :635-636 traversable: the empty polygon
  // untraversable: the rows of the bounding box that hold untraversable cells, then the hull on the host
  HIP_TRY(hipSetDevice(c->device));
  const size_t b_xy = (size_t)2 * n_vertices * sizeof(double), b_rows = (size_t)5 * c->geo.rows * sizeof(double);
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  char* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, up(b_xy) + b_rows));
  double* d_xy = (double*)d;
  double* d_rows = (double*)(d + up(b_xy));
  std::vector<double> rows5((size_t)5 * c->geo.rows);
  const size_t per = (size_t)c->geo.rows * c->geo.cols;
  hipError_t e = hipMemcpyAsync(d_xy, vertex_xy, b_xy, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = launch_polygon_untraversable_rows(c->geo, n_vertices, d_xy, c->L.untrav + per * map, d_rows, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(rows5.data(), d_rows, b_rows, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(TE_ERR_HIP, "te_polygon_untraversable_hull: %s", hipGetErrorString(e));
  std::vector<double> hull;
  untraversable_hull_from_rows(c->geo.rows, rows5.data(), hull);
  *n_hull = (int)(hull.size() / 2);
  if (*n_hull > cap_vertices)
    return fail(TE_ERR_INVALID_ARG, "te_polygon_untraversable_hull: the hull has %d vertices, room for %d", *n_hull, cap_vertices);
  if (!hull.empty()) memcpy(hull_xy, hull.data(), hull.size() * sizeof(double));
  return TE_OK;
}

int te_check_polygon_footprint_paths(te_ctx* c, int map, int n_paths, const int* pose_offset, const double* poses, int n_points,
                                     const double* points_xyz, const unsigned char* conservative, unsigned char* is_safe,
                                     double* traversability, double* area, int* status) {
  if (!c || n_paths < 0 || !points_xyz ||
      (n_paths > 0 && (!pose_offset || !poses || !is_safe || !traversability || !area || !status)))
    return fail(TE_ERR_INVALID_ARG, "te_check_polygon_footprint_paths: NULL argument");
  if (n_points < 1 || n_points > TE_MAX_POLYGON_VERTICES)
    return fail(TE_ERR_INVALID_ARG, "te_check_polygon_footprint_paths: %d footprint points (1..%d)", n_points, TE_MAX_POLYGON_VERTICES);
  for (int k = 0; k < 3 * n_points; ++k)
    if (!isfinite(points_xyz[k])) return fail(TE_ERR_INVALID_ARG, "te_check_polygon_footprint_paths: footprint point %d is not finite", k / 3);
  CtxLock lk(c);
  if (!c->have_geo || !c->footprint_done)
    return fail(TE_ERR_NOT_READY, "te_check_polygon_footprint_paths: run the chain with the footprint pass first (it marks the untraversable cells)");
  if (map < 0 || map >= c->geo.batch) return fail(TE_ERR_INVALID_ARG, "te_check_polygon_footprint_paths: map %d of %d", map, c->geo.batch);
  if (c->check_inclination && !c->have_robot_slope)
    return fail(TE_ERR_NOT_READY, "te_check_polygon_footprint_paths: check_robot_inclination is set but the layer robot_slope was never uploaded");
  if (n_paths == 0) return TE_OK;
  if (pose_offset[0] != 0) return fail(TE_ERR_INVALID_ARG, "te_check_polygon_footprint_paths: bad pose offsets");
  for (int k = 0; k < n_paths; ++k)
    if (pose_offset[k + 1] < pose_offset[k]) return fail(TE_ERR_INVALID_ARG, "te_check_polygon_footprint_paths: bad pose offsets");
  for (long k = 0; k < 7L * pose_offset[n_paths]; ++k)
    if (!isfinite(poses[k])) return fail(TE_ERR_INVALID_ARG, "te_check_polygon_footprint_paths: pose %ld is not finite", k / 7);
  // the polygons of all paths: built on the host (hulls, areas), in chunks on a few threads for large requests
  const int n_chunks = n_paths >= 4096 ? std::min<int>(16, std::max(1u, std::thread::hardware_concurrency())) : 1;
  std::vector<PathPolygons> chunk(n_chunks);
  auto first_of = [&](int q) { return (int)((long)n_paths * q / n_chunks); };
  {
    std::vector<std::thread> workers;
    for (int q = 1; q < n_chunks; ++q)
      workers.emplace_back([&, q]() {
        const int k0 = first_of(q);
        build_path_polygons(first_of(q + 1) - k0, pose_offset + k0, poses, n_points, points_xyz,
                            conservative ? conservative + k0 : nullptr, chunk[q]);
      });
    build_path_polygons(first_of(1), pose_offset, poses, n_points, points_xyz, conservative, chunk[0]);
    for (std::thread& w : workers) w.join();
  }
  // checkRobotInclination_ (:526-528, :553-557): one checkInclination per pose of a one-pose path / per segment
  // otherwise, all of them in one launch; incl_first[k] = index of path k's first test
  std::vector<unsigned char> incl_ok;
  std::vector<int> incl_st, incl_first;
  if (c->check_inclination) {
    incl_first.assign((size_t)n_paths + 1, 0);
    std::vector<double> seg;
    for (int k = 0; k < n_paths; ++k) {
      const int n = pose_offset[k + 1] - pose_offset[k];
      const double* q = poses + 7 * (size_t)pose_offset[k];
      if (n == 1) {
        seg.insert(seg.end(), {q[0], q[1], q[0], q[1]});
      } else {
        for (int i = 1; i < n; ++i) seg.insert(seg.end(), {q[7 * (i - 1)], q[7 * (i - 1) + 1], q[7 * i], q[7 * i + 1]});
      }
      incl_first[k + 1] = (int)(seg.size() / 4);
    }
    const int n_seg = incl_first[n_paths];
    incl_ok.assign(n_seg > 0 ? n_seg : 1, 0);
    incl_st.assign(n_seg > 0 ? n_seg : 1, 0);
    const int rc = check_inclination_locked(c, map, n_seg, seg.data(), incl_ok.data(), incl_st.data(),
                                            "te_check_polygon_footprint_paths");
    if (rc != TE_OK) return rc;
  }
  std::vector<unsigned char> ok;
  std::vector<double> val;
  for (int q = 0; q < n_chunks; ++q) {
    const PathPolygons& pp = chunk[q];
    const int k0 = first_of(q), nk = first_of(q + 1) - k0;
    const int n_poly = (int)pp.area.size();
    ok.assign(n_poly > 0 ? n_poly : 1, 0);
    val.assign(n_poly > 0 ? n_poly : 1, 0.0);
    const int rc = polygons_traversable_locked(c, map, n_poly, pp.vertex_offset.data(), pp.vertex_xy.data(), ok.data(),
                                               val.data(), "te_check_polygon_footprint_paths");
    if (rc != TE_OK) return rc;
    // the loop of :480-580 over the precomputed polygons; a path stops at its first untraversable polygon and keeps
    // the partial traversability / area, like `result` in the reference
    for (int kk = 0; kk < nk; ++kk) {
      const int k = k0 + kk;
      is_safe[k] = 0;
      traversability[k] = 0.0;
      area[k] = 0.0;
      status[k] = pp.status[kk];
      if (pp.status[kk] == 2) continue;
      const int n = pose_offset[k + 1] - pose_offset[k];
      bool good = true;
      for (int s = 0; s < pp.count[kk] && good; ++s) {
        const int g = pp.first[kk] + s;
        if (c->check_inclination && !incl_ok[incl_first[k] + s]) {  // before isTraversable (:553-557); the partial result stays
          status[k] = incl_st[incl_first[k] + s];
          good = false;
          break;
        }
        if (!ok[g]) {
          good = false;
          break;
        }
        if (n == 1 || s == 0) {  // :543-544, :576-577
          area[k] = pp.area[g];
          traversability[k] = val[g];
        } else {  // :570-575
          const double area_previous = area[k];
          const double area_polygon = pp.area[g] - pp.area_previous[g];
          area[k] += area_polygon;
          traversability[k] = (area_polygon * val[g] + area_previous * traversability[k]) / area[k];
        }
      }
      if (good && pp.status[kk] == 0) is_safe[k] = 1;
    }
  }
  return TE_OK;
}

int te_path_polygons(int n_paths, const int* pose_offset, const double* poses, int n_points, const double* points_xyz,
                     const unsigned char* conservative, int cap_polygons, int cap_vertices, int* n_polygons, int* n_vertices,
                     int* polygon_first, int* vertex_offset, double* vertex_xy, double* area) {
  if (n_paths < 0 || !n_polygons || !n_vertices || !points_xyz || (n_paths > 0 && (!pose_offset || !poses)))
    return fail(TE_ERR_INVALID_ARG, "te_path_polygons: NULL argument");
  if (n_points < 1 || n_points > TE_MAX_POLYGON_VERTICES)
    return fail(TE_ERR_INVALID_ARG, "te_path_polygons: %d footprint points (1..%d)", n_points, TE_MAX_POLYGON_VERTICES);
  for (int k = 0; k < n_paths; ++k)
    if (pose_offset[0] != 0 || pose_offset[k + 1] < pose_offset[k]) return fail(TE_ERR_INVALID_ARG, "te_path_polygons: bad pose offsets");
  // the same input checks as te_check_polygon_footprint_paths: a NaN pose would otherwise come back as a degenerate hull
  for (int k = 0; k < 3 * n_points; ++k)
    if (!isfinite(points_xyz[k])) return fail(TE_ERR_INVALID_ARG, "te_path_polygons: footprint point %d is not finite", k / 3);
  for (long k = 0; n_paths > 0 && k < 7L * pose_offset[n_paths]; ++k)
    if (!isfinite(poses[k])) return fail(TE_ERR_INVALID_ARG, "te_path_polygons: pose %ld is not finite", k / 7);
  PathPolygons pp;
  build_path_polygons(n_paths, pose_offset, poses, n_points, points_xyz, conservative, pp);
  *n_polygons = (int)pp.area.size();
  *n_vertices = pp.vertex_offset.back();
  if (*n_polygons > cap_polygons || *n_vertices > cap_vertices)
    return fail(TE_ERR_INVALID_ARG, "te_path_polygons: %d polygons / %d vertices do not fit the buffers (%d / %d)", *n_polygons,
                *n_vertices, cap_polygons, cap_vertices);
  // (nothing to write: a sizing call, or paths without poses -- the buffers may be NULL then)
  if (!polygon_first || !vertex_offset || (*n_vertices && !vertex_xy) || (*n_polygons && !area))
    return fail(TE_ERR_INVALID_ARG, "te_path_polygons: NULL output");
  for (int k = 0; k < n_paths; ++k) polygon_first[k] = pp.first[k];
  polygon_first[n_paths] = *n_polygons;
  memcpy(vertex_offset, pp.vertex_offset.data(), pp.vertex_offset.size() * sizeof(int));
  if (*n_vertices) memcpy(vertex_xy, pp.vertex_xy.data(), pp.vertex_xy.size() * sizeof(double));
  if (*n_polygons) memcpy(area, pp.area.data(), pp.area.size() * sizeof(double));
  return TE_OK;
}

}  // extern "C"
