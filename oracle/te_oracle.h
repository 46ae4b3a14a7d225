/*
 * te_oracle.h -- CPU ORACLE for the traversability filter chain.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's CPU algorithm for the hot path
 * (surface normals -> slope -> step -> roughness -> weighted combine -> circular footprint).
 * It exists to CHECK the HIP path and to be TIMED as the CPU baseline; it is never the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product (libtravgpu.so) neither links nor calls anything in this directory.
 *
 * Reference files restated (paths relative to /root/reference):
 *   traversability_estimation_filters/src/SlopeFilter.cpp:59-88
 *   traversability_estimation_filters/src/StepFilter.cpp:102-182
 *   traversability_estimation_filters/src/RoughnessFilter.cpp:73-132
 *   traversability_estimation/src/TraversabilityMap.cpp:307-318,654-746,774-921
 *   traversability_estimation/config/robot_filter_parameter.yaml:1-37 (chain order and defaults)
 * Un-vendored dependencies restated from their published algorithm (no version pinned by the
 * reference; de-facto ros-noetic-grid-map 1.6.x / Eigen 3.3.7 -- SURVEY.md 8c):
 *   grid_map_core : GridMap geometry (getPositionFromIndex / getIndexFromPosition / isInside /
 *                   getSubmap), GridMapIterator, CircleIterator, SpiralIterator, LineIterator
 *   grid_map_filters : NormalVectorsFilter (area method), MathExpressionFilter (fixed weighted
 *                   sum form), DeletionFilter
 *
 * Parity pin: the reference's own fixture traversability_estimation/maps/elevation_map.bag holds
 * golden outputs of the default chain; tests/test_oracle_kat.py checks this oracle against it
 * (bit-exact on step and combine, bit-exact on slope/roughness except the two exactly-planar
 * border cells documented in SURVEY.md F6).  The circular-footprint pass has NO golden vector in
 * the reference ("parity unpinned" for te_oracle_footprint).
 * One output is undefined in the reference itself: the SIGN of a horizontal normal (nz == 0 up to the eigen-solver's
 * rounding: collinear discs, the usual case at a one-cell tie radius).  NormalVectorsFilter flips only nz < 0, so rounding
 * noise picks n or -n -- here the noise of Jacobi on absolute coordinates, in the reference that of Eigen's solver; slope,
 * roughness and every later layer do not depend on it (tests/helpers.py: orient_horizontal_normals).
 *
 * Data contract: layers are float32, COLUMN-major like grid_map::Matrix (Eigen::MatrixXf):
 * element (row i, col j) at data[j * rows + i].  Invalid cell == non-finite value.
 * All per-cell arithmetic is double, as in the reference; only layer storage is float.
 */
#ifndef TE_ORACLE_H
#define TE_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct teo_geom {
  int rows, cols;      /* grid_map size(0), size(1) */
  double res;          /* resolution [m] */
  double len_x, len_y; /* rows*res, cols*res (GridMap::setGeometry) */
  double pos_x, pos_y; /* map centre position */
} teo_geom;

typedef struct teo_params {
  /* NormalVectorsFilter (robot_filter_parameter.yaml:3-9) */
  double normals_radius;
  int normals_axis; /* 0:x 1:y 2:z  (normal_vector_positive_axis) */
  /* SlopeFilter (:10-14) */
  double slope_critical;
  /* StepFilter (:15-22) */
  double step_critical, step_radius1, step_radius2;
  int step_ncrit;
  /* RoughnessFilter (:23-28) */
  double rough_critical, rough_radius;
  /* MathExpressionFilter fixed form (:29-33): out = w_scale * ((w_slope*s + w_step*t) + w_rough*r) in float32 */
  float w_scale, w_slope, w_step, w_rough;
  /* circular footprint (TraversabilityMap.cpp:307-318, robot_footprint_parameter.yaml, robot.yaml) */
  double fp_radius, fp_offset;   /* radiusMin = fp_radius, radiusMax = fp_radius + fp_offset */
  double fp_default;             /* traversability_default for NaN traversability */
  double fp_max_gap;             /* max_gap_width */
  double fp_critical_step;       /* criticalStepHeight_ (= stepFilter.critical_value, TraversabilityMap.cpp:117-126) */
  int fp_check_roughness;        /* footprint/verify_roughness_footprint */
} teo_params;

void teo_geom_init(teo_geom* g, int rows, int cols, double res, double pos_x, double pos_y);
void teo_params_default(teo_params* p); /* shipped YAML defaults */

/* number of OpenMP threads used by the loops below (1 = the reference's single thread) */
void teo_set_threads(int n);
/* 1: NormalVectorsFilter's degenerate-plane rule of the filter that wrote the reference's bag (rank-deficient scatter matrix ->
 * UnitZ; te_oracle.c).  0 (default): the current area method. */
void teo_set_normals_rank_rule(int on);
int teo_get_max_threads(void);

/* a1: NormalVectorsFilter, area method. Outputs NaN where the centre elevation is invalid. */
int teo_normals(const teo_geom* g, const float* elev, double radius, int axis, float* nx, float* ny, float* nz);
/* a2: SlopeFilter::update */
int teo_slope(const teo_geom* g, const float* nz, double crit, float* out);
/* a4+a5: StepFilter::update. step_height_out may be NULL (temp layer is erased by the filter). */
int teo_step(const teo_geom* g, const float* elev, double crit, double r1, double r2, int ncrit, float* out,
             float* step_height_out);
/* a7: RoughnessFilter::update */
int teo_roughness(const teo_geom* g, const float* elev, const float* nx, const float* ny, const float* nz,
                  double crit, double radius, float* out);
/* a9: MathExpressionFilter, fixed weighted-sum form, float32 arithmetic */
int teo_combine(long n, const float* slope, const float* step, const float* rough, float w_scale, float w_slope,
                float w_step, float w_rough, float* out);
/* a1..a10 in YAML order; normals may be NULL (DeletionFilter drops them) */
int teo_chain(const teo_geom* g, const teo_params* p, const float* elev, float* slope, float* step, float* rough,
              float* trav, float* nx, float* ny, float* nz);
/* a13+a14: traversabilityFootprint(radius, offset) over the whole map, starting from all-NaN caches.
 * slope_fp/step_fp/rough_fp (memo layers) may be NULL. */
int teo_footprint(const teo_geom* g, const teo_params* p, const float* elev, const float* slope, const float* step,
                  const float* rough, const float* trav, float* footprint, float* slope_fp, float* step_fp,
                  float* rough_fp);

/* helpers exposed for tests */
int teo_circle_count(const teo_geom* g, int i, int j, double radius);
int teo_spiral_offsets(const teo_geom* g, int ci, int cj, double radius, int* di, int* dj, int* ring, int cap);

/* N2: checkCircularFootprintPath (TraversabilityMap.cpp:344-462) for n_paths paths on a complete traversability_footprint
 * layer; path k has the poses pose_xy[2*pose_offset[k] .. 2*pose_offset[k+1]) (x, y pairs).  status: 0 ok, 1 a pose of a
 * multi-pose path lies outside the map (undefined in the reference), 2 no poses. */
int teo_check_circular_paths(const teo_geom* g, const float* footprint, double fp_default, int n_paths,
                             const int* pose_offset, const double* pose_xy, unsigned char* is_safe,
                             double* traversability, int* status);
/* the same with footprint/check_robot_inclination == true (:114, :366-370, :390-394): robot_slope is the layer
 * "robot_slope" (NULL: option off); a path whose checkInclination (:748-762) fails is unsafe with the default result.
 * status 1 also when a position handed to checkInclination lies outside the map (atPosition throws / getIndex ignored). */
int teo_check_circular_paths_incl(const teo_geom* g, const float* footprint, double fp_default, const float* robot_slope,
                                  int n_paths, const int* pose_offset, const double* pose_xy, unsigned char* is_safe,
                                  double* traversability, int* status);
/* batched TraversabilityMap::checkInclination(start, end) (:748-762): segment k = start_end_xy[4k .. 4k+4) = sx sy ex ey */
int teo_check_inclination(const teo_geom* g, const float* robot_slope, int n, const double* start_end_xy, unsigned char* ok,
                          int* status);

/* N3: batched TraversabilityMap::isTraversable(polygon, traversability) (:586-645); polygon k has the vertices
 * vertex_xy[2*vertex_offset[k] .. 2*vertex_offset[k+1]).  Returns -1 for a polygon without vertices. */
int teo_polygons_traversable(const teo_geom* g, const teo_params* p, const float* elev, const float* slope, const float* step,
                             const float* rough, const float* trav, int n_polygons, const int* vertex_offset,
                             const double* vertex_xy, unsigned char* is_traversable, double* traversability);
/* isTraversable(polygon, computeUntraversablePolygon = true, ..) (:592-645): additionally the untraversable polygon = convex
 * hull (monotoneChainConvexHullOfPoints) of the positions of the polygon's untraversable cells; n_hull = 0 when traversable.
 * Returns -3 when the hull has more than cap vertices. */
int teo_polygon_untraversable_hull(const teo_geom* g, const teo_params* p, const float* elev, const float* slope,
                                   const float* step, const float* rough, const float* trav, int n_vertices,
                                   const double* vertex_xy, unsigned char* is_traversable, double* traversability, int cap,
                                   int* n_hull, double* hull_xy);
/* the footprint points turned by yaw about z exactly as :250-283 does it (kindr angle-axis -> Eigen quaternion -> matrix) */
void teo_rotate_footprint(int n_points, const double* points_xy, double yaw, double* out_xy);
/* traversabilityFootprint(footprintYaw) (:239-305): layers traversability_x / traversability_rot */
int teo_polygon_footprint(const teo_geom* g, const teo_params* p, const float* elev, const float* slope, const float* step,
                          const float* rough, const float* trav, int n_points, const double* points_xy, double yaw,
                          float* trav_x, float* trav_rot);

/* N2 (polygon footprints): checkPolygonalFootprintPath (:464-584) for n_paths paths.  poses: 7 doubles per pose (position
 * x y z, orientation x y z w), footprint points_xyz: 3 doubles per point (<= 32 points), conservative: per path flag or
 * NULL.  Outputs per path like TraversabilityResult: is_safe, traversability, area (partial values stay when a segment
 * fails, as in the reference).  status: 0 ok, 2 no poses, 3 conservative vertex lists outgrew the buffer. */
int teo_check_polygon_paths(const teo_geom* g, const teo_params* p, const float* elev, const float* slope, const float* step,
                            const float* rough, const float* trav, int n_paths, const int* pose_offset, const double* poses,
                            int n_points, const double* points_xyz, const unsigned char* conservative, unsigned char* is_safe,
                            double* traversability, double* area, int* status);
/* with footprint/check_robot_inclination == true (:526-528, :553-557); robot_slope NULL: option off; status 1: a
 * position handed to checkInclination lies outside the map */
int teo_check_polygon_paths_incl(const teo_geom* g, const teo_params* p, const float* elev, const float* slope,
                                 const float* step, const float* rough, const float* trav, const float* robot_slope,
                                 int n_paths, const int* pose_offset, const double* poses, int n_points,
                                 const double* points_xyz, const unsigned char* conservative, unsigned char* is_safe,
                                 double* traversability, double* area, int* status);

#ifdef __cplusplus
}
#endif
#endif
