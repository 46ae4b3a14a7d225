/*
 * travgpu.h -- C-ABI of libtravgpu.so: the MI355X (gfx950) traversability filter chain.
 *
 * This is the drop-in boundary for ONE hot path of leggedrobotics/traversability_estimation: the
 * filter chain that `filter_chain_.update(elevationMapCopy, traversabilityMapCopy)` runs
 * (traversability_estimation/src/TraversabilityMap.cpp:214) plus the circular footprint pass
 * (TraversabilityMap.cpp:307-318).  The shim owns the device-resident elevation and output layers
 * and launches the HIP chain; plain pointers and sizes only, no C++/torch types.
 *
 * Which reference interface each entry point replaces (paths relative to the reference root):
 *
 *   te_set_params        <- FilterBase<T>::getParam() reads in
 *                           traversability_estimation_filters/src/SlopeFilter.cpp:34-56,
 *                           StepFilter.cpp:38-99, RoughnessFilter.cpp:36-70, the
 *                           NormalVectorsFilter / MathExpressionFilter entries of
 *                           traversability_estimation/config/robot_filter_parameter.yaml:3-9,29-33
 *                           and the footprint parameters read in TraversabilityMap.cpp:108-126
 *   te_set_geometry      <- grid_map::GridMap::setGeometry (length/resolution/position) as used by
 *                           TraversabilityMap::setElevationMap, TraversabilityMap.cpp:135-154
 *   te_upload_elevation  <- the "elevation" layer of mapIn handed to every plugin's
 *                           update(const T& mapIn, T& mapOut) (SlopeFilter.cpp:59, StepFilter.cpp:102,
 *                           RoughnessFilter.cpp:73)
 *   te_run_filter        <- one plugin's update(): SlopeFilter.cpp:59-88 / StepFilter.cpp:102-182 /
 *                           RoughnessFilter.cpp:73-132, reading the layers that plugin reads from mapIn
 *   te_run_chain         <- filters::FilterChain<GridMap>::update, TraversabilityMap.cpp:214
 *                           (NormalVectorsFilter -> SlopeFilter::update -> StepFilter::update ->
 *                           RoughnessFilter::update -> MathExpressionFilter -> DeletionFilter)
 *   te_run_footprint     <- TraversabilityMap::traversabilityFootprint(radius, offset),
 *                           TraversabilityMap.cpp:307-318 (isTraversable :654-746,
 *                           isTraversableForFilters :774-792, checkFor{Slope,Step,Roughness} :794-921)
 *   te_download_layer    <- mapOut.add(type_) / mapOut.at(type_, index) results of each plugin
 *                           (SlopeFilter.cpp:63,77-80 etc.)
 *   te_check_footprint_paths, te_check_polygon_footprint_paths
 *                        <- TraversabilityMap::checkFootprintPath, TraversabilityMap.cpp:320-342
 *                           (checkCircularFootprintPath :344-462, checkPolygonalFootprintPath :464-584)
 *   te_check_inclination, te_set_check_robot_inclination
 *                        <- TraversabilityMap::checkInclination :748-762 and footprint/check_robot_inclination :114
 *   te_polygons_traversable <- TraversabilityMap::isTraversable(polygon, traversability), :586-645
 *   te_polygon_untraversable_hull <- isTraversable(polygon, computeUntraversablePolygon, ..), :592-645
 *   te_run_polygon_footprint <- TraversabilityMap::traversabilityFootprint(footprintYaw), :239-305
 *   te_upload_msg / te_download_msg / te_msg_* / te_bag_*
 *                        <- GridMapRosConverter::fromMessage / toMessage / loadFromBag / saveToBag as called in
 *                           TraversabilityMap.cpp:135-154 and TraversabilityEstimation.cpp:125-152, 248-270, 318-329
 *
 * Data contract (identical to grid_map::Matrix = Eigen::MatrixXf): float32, COLUMN-major,
 * element (row i, col j) of map m at ptr[m*rows*cols + j*rows + i]; invalid cell = non-finite.
 * The device layers are in logical order; a GridMap that has been move()d (circular-buffer start index != (0,0))
 * goes through te_upload_layer_circular / te_download_layer_circular / te_upload_msg, which rotate inside the copy.
 *
 * All functions return TE_OK (0) or a negative te_status; te_last_error() gives the message of the
 * calling thread's last failure.  A context is internally serialised (one mutex, one HIP stream);
 * different contexts may be used concurrently from different threads.
 */
#ifndef TRAVGPU_H
#define TRAVGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TE_ABI_VERSION 1

typedef enum te_status {
  TE_OK = 0,
  TE_ERR_INVALID_ARG = -1, /* NULL pointer, bad enum, bad size */
  TE_ERR_BAD_PARAM = -2,   /* a filter parameter outside the range the reference's configure() accepts */
  TE_ERR_NOT_READY = -3,   /* geometry/params/elevation missing, or chain not run before footprint */
  TE_ERR_HIP = -4,         /* HIP runtime error (message has hipGetErrorString) */
  TE_ERR_NO_DEVICE = -5,   /* no usable gfx950 device: the library never falls back to the CPU */
  TE_ERR_UNSUPPORTED = -6  /* e.g. radius too large for the on-chip tile */
} te_status;

/* Layers owned by a context (device-resident, [batch][cols][rows] float32). */
typedef enum te_layer {
  TE_LAYER_ELEVATION = 0,
  TE_LAYER_SLOPE = 1,       /* traversability_slope      (SlopeFilter map_type)     */
  TE_LAYER_STEP = 2,        /* traversability_step       (StepFilter map_type)      */
  TE_LAYER_ROUGHNESS = 3,   /* traversability_roughness  (RoughnessFilter map_type) */
  TE_LAYER_TRAVERSABILITY = 4,
  TE_LAYER_FOOTPRINT = 5,   /* traversability_footprint  */
  TE_LAYER_NORMAL_X = 6,    /* surface_normal_{x,y,z}: only kept with TE_RUN_KEEP_NORMALS */
  TE_LAYER_NORMAL_Y = 7,
  TE_LAYER_NORMAL_Z = 8,
  TE_LAYER_SLOPE_FOOTPRINT = 9,   /* memo layers of checkForSlope/Step/Roughness (0/1/NaN) */
  TE_LAYER_STEP_FOOTPRINT = 10,
  TE_LAYER_ROUGHNESS_FOOTPRINT = 11,
  TE_LAYER_TRAVERSABILITY_X = 12,   /* traversability_x / traversability_rot: exist after te_run_polygon_footprint */
  TE_LAYER_TRAVERSABILITY_ROT = 13,
  TE_LAYER_ROBOT_SLOPE = 14,   /* input of checkInclination (robotSlopeType_, TraversabilityMap.cpp:47): optional, exists
                                  from its first upload on (te_upload_layer / _circular / te_upload_msg / te_device_ptr) */
  TE_LAYER_COUNT = 15
} te_layer;

/* te_run_chain flags */
#define TE_RUN_KEEP_NORMALS 0x1u /* also write surface_normal_{x,y,z} (i.e. no DeletionFilter) */
#define TE_RUN_FOOTPRINT    0x2u /* run the circular footprint pass right after the chain */
#define TE_RUN_FOOTPRINT_MEMO 0x8u /* with the footprint pass: also write slope_/step_/roughness_footprint (0/1/NaN) */
#define TE_RUN_SEQUENTIAL 0x10u /* one HIP stream only (default: step filter and normals kernel overlap on two streams) */
#define TE_RUN_NORMALS_ONLY 0x20u /* measurement aid: only the normals/slope/roughness kernel (+ its fix-up pass) */
#define TE_RUN_GENERIC_KERNELS 0x4u /* use only the shape-generic kernels (also the path for tie radii); for A/B tests */

/* The reference's plugins one at a time (te_run_filter): what the drop-in adapters of
 * traversabilityFilters/{Slope,Step,Roughness}Filter call, each reading exactly the layers the
 * reference plugin reads from mapIn. */
typedef enum te_filter {
  TE_FILTER_SLOPE = 1,     /* in: surface_normal_z                      out: traversability_slope     (SlopeFilter.cpp:59-88) */
  TE_FILTER_STEP = 2,      /* in: elevation                             out: traversability_step      (StepFilter.cpp:102-182) */
  TE_FILTER_ROUGHNESS = 3, /* in: elevation, surface_normal_{x,y,z}     out: traversability_roughness (RoughnessFilter.cpp:73-132) */
  TE_FILTER_COMBINE = 4,   /* in: the three scores                      out: traversability           (MathExpressionFilter) */
  TE_FILTER_NORMALS = 5    /* in: elevation                             out: surface_normal_{x,y,z}   (NormalVectorsFilter, area method;
                              robot_filter_parameter.yaml:3-9; README.md:173 "Surface Normals Filter") */
} te_filter;

/* Filter parameters: same keys, defaults and validity ranges as the reference's configure()s.
 * POD; `size` must be sizeof(te_params) (ABI check).  This is the blob that is broadcast to the
 * other ranks (RCCL) when the batch is sharded over several GPUs. */
typedef struct te_params {
  uint32_t size;
  uint32_t abi_version;
  /* gridMapFilters/NormalVectorsFilter: radius, normal_vector_positive_axis */
  double normals_radius;
  int32_t normals_axis; /* 0:x 1:y 2:z */
  int32_t _pad0;
  /* traversabilityFilters/SlopeFilter: critical_value in [0, pi/2] */
  double slope_critical;
  /* traversabilityFilters/StepFilter: critical_value>=0, first/second_window_radius>=0, critical_cell_number>0 */
  double step_critical;
  double step_radius1;
  double step_radius2;
  int32_t step_ncrit;
  int32_t _pad1;
  /* traversabilityFilters/RoughnessFilter: critical_value>=0, estimation_radius>=0 */
  double rough_critical;
  double rough_radius;
  /* gridMapFilters/MathExpressionFilter, fixed form, float32:
   *   traversability = w_scale * ((w_slope*slope + w_step*step) + w_rough*roughness)
   * default (1.0f/3.0f) and 1,1,1 == the shipped expression bit for bit */
  float w_scale, w_slope, w_step, w_rough;
  /* circular footprint: radiusMin = fp_radius, radiusMax = fp_radius + fp_offset */
  double fp_radius;
  double fp_offset;
  double fp_default;       /* footprint/traversability_default */
  double fp_max_gap;       /* max_gap_width */
  double fp_critical_step; /* criticalStepHeight_ = stepFilter.critical_value */
  int32_t fp_check_roughness;
  int32_t _pad2;
} te_params;

typedef struct te_ctx te_ctx;

/* Fill `p` with the shipped defaults (robot_filter_parameter.yaml, robot_footprint_parameter.yaml, robot.yaml). */
int te_params_default(te_params* p);
/* Range checks of the reference's configure()s; TE_ERR_BAD_PARAM + message on violation. */
int te_params_validate(const te_params* p);

int te_device_count(int* count);
int te_create(int device, te_ctx** out);
int te_destroy(te_ctx* ctx);

int te_set_params(te_ctx* ctx, const te_params* p);
int te_get_params(te_ctx* ctx, te_params* p);
/* Choices between kernels that produce IDENTICAL layers (no reference counterpart: the reference has one code path).
 * The defaults pick by input size; the options exist so that tests and measurements can reach every path through the
 * ABI -- the library never reads the environment. */
#define TE_OPT_FP_BLOCKED_WALK 1          /* discs with an untraversable cell: 0 by list length, 1 one disc per wavefront, 2 one disc per lane */
#define TE_OPT_FP_BLOCKED_BLOCKS_PER_CU 2 /* grid of that kernel: 0 default (24), 1 .. 32 */
#define TE_OPT_GRAPH_REPLAY 4             /* whole-map launches as a captured hipGraph: 0 by size (the default: from 2^22 cells), 1 always, 2 never */
#define TE_OPT_POLYGON_PER_CELL 3         /* polygon footprint layers: 1 evaluates every cell of every bounding box instead of the offset table */
#define TE_OPT_BCAST_RCCL 5               /* te_bcast_params, set on the ROOT context: 0 RCCL only between different devices (the default), 1 also when all contexts share one device (a communicator of one rank) */
#define TE_OPT_NORMALS_RANK_RULE 6        /* 1: NormalVectorsFilter as grid_map <= 1.6 had it (the filter that wrote TE/maps/elevation_map.bag): a disc whose scatter matrix is rank-deficient -- exactly planar -- gets UnitZ; runs the shape-generic kernels.  0 (default): the current area method */
int te_set_option(te_ctx* ctx, int option, int value);
/* rows = size(0), cols = size(1) of every map of the batch; (pos_x,pos_y) = map centre. */
int te_set_geometry(te_ctx* ctx, int rows, int cols, int batch, double resolution, double pos_x, double pos_y);

/* Host -> device copy of `nmaps` maps starting at batch slot `map0` (column-major float32). */
int te_upload_elevation(te_ctx* ctx, const float* host, int map0, int nmaps);
/* Overwrite the h x w sub-rectangle with top-left cell (row0, col0) of map `map` from a packed
 * column-major h x w host tile (dirty-region update). */
int te_upload_tile(te_ctx* ctx, const float* host_tile, int map, int row0, int col0, int h, int w);
/* Device pointer of a layer ([batch][cols][rows] float32) for zero-copy producers/consumers
 * (e.g. a torch tensor filled on the same device); valid until te_set_geometry/te_destroy.
 * Writers order their work against the context themselves (te_sync, or the same device stream order).  Handing out
 * TE_LAYER_TRAVERSABILITY marks that layer as caller-writable until the next te_set_geometry: te_run_footprint and
 * region runs then no longer assume its values are bounded by the chain's weights (they use the double-precision
 * footprint kernel); te_run_chain with TE_RUN_FOOTPRINT rewrites every cell first and is unaffected. */
int te_device_ptr(te_ctx* ctx, int layer, void** dptr, size_t* bytes);
/* The optional input layer robot_slope (checkInclination, TraversabilityMap.cpp:748-762; it travels with the elevation
 * map when the caller has one) is present after an upload.  present = 0 declares it absent again -- an elevation map that
 * comes without the layer must not be checked against the previous map's -- and present = 1 declares a buffer filled
 * through te_device_ptr ready.  With check_robot_inclination set the path checks fail (TE_ERR_NOT_READY) while the
 * layer is absent, like the reference's atPosition() throws for a missing layer. */
int te_set_layer_present(te_ctx* ctx, int layer, int present);

/* Host -> device copy of any layer (e.g. surface_normal_z for TE_FILTER_SLOPE); elevation marks the context ready. */
int te_upload_layer(te_ctx* ctx, int layer, const float* host, int map0, int nmaps);
/* The same for a layer in GridMap's circular-buffer order: logical cell (i, j) is stored at ((i + start_row) % rows,
 * (j + start_col) % cols) (grid_map_core getBufferIndexFromIndex; start index = GridMap::getStartIndex(), non-zero after
 * GridMap::move).  The reference's filters see maps in this form: their iterators (StepFilter.cpp:112,124) hide the
 * start index.  The device layers are always in logical order. */
int te_upload_layer_circular(te_ctx* ctx, int layer, const float* host, int map, int start_row, int start_col);
int te_download_layer_circular(te_ctx* ctx, int layer, float* host, int map, int start_row, int start_col);
/* Whole-layer uploads (all maps of the batch) that run BESIDE the calls that follow: the reference's FilterChain hands every
 * plugin the whole map (SlopeFilter.cpp:62-63, StepFilter.cpp:105-107, RoughnessFilter.cpp:76-77), so a plugin knows which
 * layers its successors will read (StepFilter: elevation; RoughnessFilter: elevation + surface_normal_{x,y,z}) and can send
 * them host -> device while its own filter runs and its output crosses device -> host -- PCIe is full duplex.  Returns at
 * once; a thread of the library stages the buffers through a second ring of page-locked slots.  Until te_wait_prefetch
 * returns the host buffers must stay valid.  te_run_filter, te_download_layer(_circular), te_get_params / te_set_params
 * run BESIDE a prefetch in flight as long as they neither read nor write one of its layers; a call that does (e.g.
 * TE_FILTER_STEP beside an elevation prefetch, a download of a prefetched layer) and every other entry point finish the
 * prefetch first -- no call ever sees a half-written layer.  A prefetch that failed leaves its layers undefined:
 * te_wait_prefetch reports it, and an elevation layer among them must be uploaded again before the next chain
 * (TE_ERR_NOT_READY until then).  A prefetched elevation layer is scanned for invalid cells when the prefetch is joined (one
 * short kernel on the context's stream and a wait for its two counters, as te_upload_elevation does): the count and the
 * run count pick the normals kernel's march and strip height (hole-free, scattered cells, unobserved regions).  n <= 8. */
int te_prefetch_layers(te_ctx* ctx, int n, const int* layers, const float* const* hosts);
int te_wait_prefetch(te_ctx* ctx);
/* Run ONE of the reference's plugins on the resident layers (see te_filter). */
int te_run_filter(te_ctx* ctx, int filter, unsigned flags);
int te_run_chain(te_ctx* ctx, unsigned flags);
/* Re-filter only the cells whose outputs can change when the h x w rectangle at (row0,col0) of map
 * `map` changed (the rectangle dilated by the chain's reach). */
/* With TE_RUN_FOOTPRINT (| TE_RUN_FOOTPRINT_MEMO) the circular footprint pass follows on the cells that can see the
 * change (the re-filtered cells grown by 3 cells for isTraversableForFilters and by the footprint's reach); the
 * traversability_footprint layer must have been complete before (te_run_chain with the flag, or te_run_footprint),
 * TE_ERR_NOT_READY otherwise.  The caller of the reference's node re-filters the whole map on every update
 * (TraversabilityMap.cpp:202-237); this is the incremental form of the same call.
 * A footprint shape none of the shape-specialised sum kernels takes (a tie radius that is not a whole number of cells,
 * a reach of 17..20 cells, a map narrower than 64 cells) is served by the general kernel, which recomputes the footprint
 * layer of the WHOLE map `map` (no other map of the batch): cost O(map), not O(region); cells outside the region are
 * recomputed from unchanged inputs and keep their values to within the fixed-point kernels' rounding (< 1e-6). */
int te_run_chain_region(te_ctx* ctx, unsigned flags, int map, int row0, int col0, int h, int w);
/* The h x w rectangle at (row0, col0) of a layer of map `map` into a packed column-major h x w host tile (the layout
 * te_upload_tile reads); returns when the tile is in host memory. */
int te_download_tile(te_ctx* ctx, int layer, int map, int row0, int col0, int h, int w, float* host_tile);
/* Streaming updates (BASELINE configs[4]: a resident map, one dirty tile per tick).  te_upload_tile_async returns at
 * once: the tile crosses PCIe on the context's copy-in stream into a device staging slot -- concurrently with the kernels
 * of the previous tick -- and is placed into the elevation layer on the compute stream, in order with the launches
 * that follow (te_run_chain_region).  te_download_tile_async is its mirror: the rectangle is copied to a staging slot in
 * order with the launches before it and crosses PCIe on the copy-out stream while the next tick computes.  Two slots
 * each way.  host_tile must stay valid until te_sync (which waits for all three streams) -- page-lock it (te_pin_host),
 * otherwise the runtime stages the copy itself and the call blocks for its duration. */
int te_upload_tile_async(te_ctx* ctx, const float* host_tile, int map, int row0, int col0, int h, int w);
int te_download_tile_async(te_ctx* ctx, int layer, int map, int row0, int col0, int h, int w, float* host_tile);
int te_run_footprint(te_ctx* ctx);
/* Batched TraversabilityMap::checkFootprintPath for circular footprints (TraversabilityMap.cpp:320-342 ->
 * checkCircularFootprintPath :344-462) on the resident traversability_footprint layer of map `map`, which must be
 * complete (te_run_chain with TE_RUN_FOOTPRINT or te_run_footprint, with fp_radius = the paths' radius and
 * fp_offset = 0.15 like :348): isTraversable() then takes its memo branch (:672-677) for every centre.
 * Path k has the poses pose_xy[2*pose_offset[k] .. 2*pose_offset[k+1]) (x, y in the map frame; pose_offset[0] == 0).
 * Outputs per path: is_safe and traversability (TraversabilityResult, :352-355), status 0 ok / 1 a pose of a
 * multi-pose path lies outside the map (the reference ignores getIndex()'s failure there: undefined) / 2 no poses
 * (:330-334).  With te_set_check_robot_inclination(ctx, 1) every pose of a one-pose path / every segment first passes
 * checkInclination (:366-370, :390-394) on the layer robot_slope; a failure leaves the default result (unsafe, 0), and
 * status 1 also marks a position checkInclination was handed outside the map (atPosition throws there).  Reference
 * options not covered: publishPolygons (ROS markers) and the untraversable polygon of compute_untraversable_polygon, which
 * on a complete footprint layer is Polygon::fromCircle of the unsafe centres (:676-678) -- no map data; the result is
 * the same with and without it.  Host buffers; synchronous. */
int te_check_footprint_paths(te_ctx* ctx, int map, int n_paths, const int* pose_offset, const double* pose_xy,
                             unsigned char* is_safe, double* traversability, int* status);
/* footprint/check_robot_inclination (TraversabilityMap.cpp:114, default false): when set, te_check_footprint_paths and
 * te_check_polygon_footprint_paths run checkInclination before every isTraversable, reading TE_LAYER_ROBOT_SLOPE (the
 * reference's layer "robot_slope", written by whoever estimates the robot's inclination); TE_ERR_NOT_READY from the
 * path checks if that layer was never uploaded. */
int te_set_check_robot_inclination(te_ctx* ctx, int enabled);
/* Batched TraversabilityMap::checkInclination(start, end) (:748-762) on the layer robot_slope of map `map`: segment k =
 * start_end_xy[4k .. 4k+4) = start x y, end x y.  start == end: ok = the cell's value != 0; otherwise a LineIterator from
 * the start index to the end index, cells that are not valid skipped, ok = no cell is 0.  status 0 / 1 a position lies
 * outside the map (ok = 0 then).  Host buffers; synchronous. */
int te_check_inclination(te_ctx* ctx, int map, int n_segments, const double* start_end_xy, unsigned char* ok, int* status);
/* TraversabilityMap::traversabilityFootprint(footprintYaw) (TraversabilityMap.cpp:239-305): for every cell of every map the
 * footprint polygon (n_points vertices points_xy = x0 y0 x1 y1 .. in the footprint frame, footprint/footprint_polygon
 * :91-103) centred on the cell, as given -> layer traversability_x, and turned by `yaw` about z -> traversability_rot;
 * each cell gets isTraversable(polygon)'s mean (:586-645) or 0 when the polygon touches an untraversable cell.  Needs the
 * untraversable-cell mask the circular footprint pass leaves behind (te_run_chain with TE_RUN_FOOTPRINT or
 * te_run_footprint first).  At most TE_MAX_POLYGON_VERTICES points.  Asynchronous like te_run_chain; read the layers
 * with te_download_layer(TE_LAYER_TRAVERSABILITY_X / _ROT). */
#define TE_MAX_POLYGON_VERTICES 32
int te_run_polygon_footprint(te_ctx* ctx, int n_points, const double* points_xy, double yaw);
/* Batched TraversabilityMap::isTraversable(polygon, traversability) (:586-645) on map `map`: polygon k has the vertices
 * vertex_xy[2*vertex_offset[k] .. 2*vertex_offset[k+1]) in the map frame (at least one each; vertex_offset[0] == 0).
 * traversability[k] = mean over the polygon's cells, traversabilityDefault_ when it covers no cell centre, 0 when
 * is_traversable[k] == 0.  The per-segment polygons of checkPolygonalFootprintPath (:464-584) go through this.  Same
 * precondition as above.  Host buffers; synchronous. */
int te_polygons_traversable(te_ctx* ctx, int map, int n_polygons, const int* vertex_offset, const double* vertex_xy,
                            unsigned char* is_traversable, double* traversability);
/* TraversabilityMap::isTraversable(polygon, computeUntraversablePolygon = true, traversability, untraversablePolygon)
 * (:592-645; FootprintPath.compute_untraversable_polygon) for ONE polygon on map `map`: is_traversable / traversability as
 * te_polygons_traversable, plus the untraversable polygon = grid_map::Polygon::monotoneChainConvexHullOfPoints of the
 * positions of every untraversable cell inside the polygon (*n_hull = 0 when traversable; the points as collected when
 * there are at most three).  hull_xy holds cap_vertices vertices (x y); TE_ERR_INVALID_ARG with *n_hull set if the hull
 * has more.  The device reduces every row of the bounding box to its outermost untraversable cells, the chain runs on
 * the host.  For circular footprints on a complete footprint layer the reference's untraversable polygon is just
 * Polygon::fromCircle(center, radius + offset) of an unsafe centre (:676-678): no map data, left to the caller.
 * Same precondition as te_polygons_traversable.  Host buffers; synchronous. */
int te_polygon_untraversable_hull(te_ctx* ctx, int map, int n_vertices, const double* vertex_xy, unsigned char* is_traversable,
                                  double* traversability, int cap_vertices, int* n_hull, double* hull_xy);
/* Batched TraversabilityMap::checkFootprintPath for polygonal footprints (checkPolygonalFootprintPath, :464-584).
 * Path k has the poses poses[7*pose_offset[k] .. 7*pose_offset[k+1]) -- position x y z, orientation x y z w, as in
 * geometry_msgs/Pose; the footprint is n_points points x y z (path.footprint.polygon.points) in the footprint frame;
 * conservative[k] = path.conservative (NULL: all false).  The pose polygons (toPosition * orientation * point), their
 * conservative extensions, the convex hull of consecutive ones (grid_map::Polygon::convexHull) and the areas are computed
 * on the host, every polygon's isTraversable on the device in one launch.  Outputs per path = TraversabilityResult:
 * is_safe, traversability, area (a path that fails keeps the values of the segments before, as the reference's result
 * does).  status: 0 ok, 1 check_robot_inclination is set and a position handed to checkInclination lies outside the map,
 * 2 no poses (:330-334), 3 the conservative vertex lists outgrew 1024 vertices.  With te_set_check_robot_inclination
 * checkInclination runs before every polygon (:526-528, :553-557).  Not covered: publishPolygons (ROS markers); the
 * untraversable polygon of compute_untraversable_polygon is te_polygon_untraversable_hull on the polygons
 * te_path_polygons returns (it is only ever published, :531-533, :559-561; the result is the same).  Host buffers; synchronous. */
int te_check_polygon_footprint_paths(te_ctx* ctx, int map, int n_paths, const int* pose_offset, const double* poses, int n_points,
                                     const double* points_xyz, const unsigned char* conservative, unsigned char* is_safe,
                                     double* traversability, double* area, int* status);
/* Host part of the above on its own (no device, no context): the polygons checkPolygonalFootprintPath hands to
 * isTraversable -- the pose polygon of a one-pose path, the convex hull of consecutive (conservatively extended) pose
 * polygons otherwise -- e.g. to publish them like publishFootprintPolygon (:527, :558).  Path k owns the polygons
 * polygon_first[k] .. polygon_first[k+1]); polygon p has the vertices vertex_xy[2*vertex_offset[p] .. 2*vertex_offset[p+1])
 * and the area area[p] (Polygon::getArea).  *n_polygons / *n_vertices receive the totals; nothing is written beyond
 * cap_polygons polygons / cap_vertices vertices (TE_ERR_INVALID_ARG then: call once with zero capacities to size).
 * Poses and footprint points must be finite (TE_ERR_INVALID_ARG).  The per-path status of the full check is not reported
 * here: a path without poses owns no polygon (polygon_first[k] == polygon_first[k+1]); a path whose conservative vertex
 * lists outgrow 1024 vertices ends with the last polygon that fits (te_check_polygon_footprint_paths reports status 3). */
int te_path_polygons(int n_paths, const int* pose_offset, const double* poses, int n_points, const double* points_xyz,
                     const unsigned char* conservative, int cap_polygons, int cap_vertices, int* n_polygons, int* n_vertices,
                     int* polygon_first, int* vertex_offset, double* vertex_xy, double* area);
int te_sync(te_ctx* ctx);

int te_download_layer(te_ctx* ctx, int layer, float* host, int map0, int nmaps);
/* Page-lock a host buffer the caller keeps across frames (a GridMap layer that lives as long as the node): uploads from
 * and downloads into it then run as direct DMA instead of through the runtime's staging copies.  te_unpin_host before
 * the buffer is freed.  Purely an optimisation: every transfer entry point accepts pageable memory too. */
int te_pin_host(void* host, size_t bytes);
int te_unpin_host(void* host);

/* ---- several GPUs from one process -------------------------------------------------------------
 * The path shards on the batch axis only: maps are independent, there is no data-path collective.  A host that owns
 * all GPUs of a node (the reference's node is one process) creates one context per device, gives every context its
 * block of the batch, and drives them from one thread -- every entry point above is asynchronous on its context's own
 * stream, so the devices run side by side.
 *   te_shard_range      contiguous block [first, first + count) of `batch` maps owned by shard k of n (sizes differ by
 *                       at most one map; the same split as traversability_estimation_amd/dist.py)
 *   te_bcast_params     every context receives the parameters of ctxs[root].  Contexts on different devices get the
 *                       te_params block through an RCCL broadcast over xGMI (librccl is loaded when first needed;
 *                       communicators are created per call: this is configure-time, TraversabilityMap.cpp:764-772);
 *                       contexts sharing the root's device are copied on the host.  TE_ERR_UNSUPPORTED if several
 *                       devices are involved and librccl cannot be loaded.
 *   te_run_chain_multi  te_run_chain(flags) on every context, in order, without waiting
 *   te_sync_multi       te_sync on every context */
int te_shard_range(int batch, int n_shards, int k, int* first, int* count);
int te_bcast_params(te_ctx** ctxs, int n, int root);
int te_run_chain_multi(te_ctx** ctxs, int n, unsigned flags);
int te_sync_multi(te_ctx** ctxs, int n);

/* Time `iters` back-to-back te_run_chain(flags) launches with HIP events on the context's stream
 * (after `warmup` untimed ones); inputs and outputs stay resident in HBM. */
int te_time_chain(te_ctx* ctx, unsigned flags, int warmup, int iters, float* ms_per_iter);
/* The same, one event pair per launch: ms[k] = device time of the k-th launch (for a median; the launches are not
 * back to back, every one is waited for). */
int te_time_chain_samples(te_ctx* ctx, unsigned flags, int warmup, int iters, float* ms);

/* ---- wire formats either side of the chain: grid_map_msgs/GridMap (ROS1 serialisation) and rosbag V2.0 ----
 * The reference node gets its elevation map as such a message (TraversabilityEstimation.cpp:248-270 requestElevationMap ->
 * GridMapRosConverter::fromMessage), loads / saves maps from / to bags (:125-152 loadFromBag, :318-329 saveToBag) and
 * publishes its result with toMessage.  A layer's float32 payload in the message IS the layer's Eigen matrix
 * (column-major, circular start index = outer/inner_start_index), so it moves between the message buffer and the device
 * with te_upload_layer_circular's rectangle copies: no host-side GridMap, no reshuffling. */
#define TE_MSG_MAX_NAME 64
typedef struct te_msg_info {
  uint32_t seq, stamp_sec, stamp_nsec; /* info.header */
  char frame_id[TE_MSG_MAX_NAME];      /* NUL-terminated */
  double resolution, length_x, length_y;
  double pose[7];                /* position x y z, orientation x y z w; GridMap uses position x, y only */
  int32_t rows, cols;            /* size of every layer: dim[1].size, dim[0].size */
  int32_t start_row, start_col;  /* outer_start_index, inner_start_index = GridMap::getStartIndex()(0), (1) */
  int32_t n_layers, n_basic_layers;
} te_msg_info;

/* Validate a serialised message and describe it.  Rejected (TE_ERR_INVALID_ARG + message): truncation, layers/data count
 * mismatch (fromMessage's own check), storage order other than (column_index, row_index), layer sizes that disagree with
 * each other or with round(length / resolution), start index outside the map. */
int te_msg_parse(const void* msg, size_t len, te_msg_info* info);
/* Name of layer k (NUL-terminated into name[TE_MSG_MAX_NAME]) and byte offset of its rows*cols float32 payload. */
int te_msg_layer(const void* msg, size_t len, int k, char* name, size_t* data_offset);
/* toMessage for host layers: layer_data[k] = rows*cols float32 in the GridMap's own (column-major, circular) storage order;
 * every field of `info` is used.  *written = bytes needed even when `cap` is too small (out = NULL, cap = 0 sizes). */
int te_msg_write(const te_msg_info* info, int n_layers, const char* const* names, const float* const* layer_data, int n_basic,
                 const char* const* basic_names, void* out, size_t cap, size_t* written);
/* fromMessage + upload in one step: (re)sets the context's geometry from the message (batch 1: rows, cols, resolution,
 * position) when it differs, then copies layer `layer_name` into device layer `layer` (te_layer), undoing the circular
 * start index.  `info` (may be NULL) receives the parsed description. */
int te_upload_msg(te_ctx* ctx, const void* msg, size_t len, const char* layer_name, int layer, te_msg_info* info);
/* toMessage: serialise n_layers device layers (te_layer ids `layers`, message names `names`) of map 0 into `out`.
 * Geometry comes from the context; seq, stamp, frame_id, pose z / orientation and the start index from `info` (its
 * rows / cols / lengths / resolution are ignored).  *written = bytes needed even when TE_ERR_INVALID_ARG reports that
 * `cap` is too small (call with out = NULL, cap = 0 to size the buffer). */
int te_download_msg(te_ctx* ctx, const te_msg_info* info, int n_layers, const int* layers, const char* const* names, int n_basic,
                    const char* const* basic_names, void* out, size_t cap, size_t* written);
/* loadFromBag: the last grid_map_msgs/GridMap message stored under `topic` in an (uncompressed-chunk) rosbag V2.0 image. */
int te_bag_find_message(const void* bag, size_t len, const char* topic, size_t* msg_offset, size_t* msg_len);
/* saveToBag: a one-message bag (stamp 0.0 is written as ros::TIME_MIN like the reference).  Sizing call as above. */
int te_bag_write(const void* msg, size_t msg_len, const char* topic, uint32_t stamp_sec, uint32_t stamp_nsec, void* out,
                 size_t cap, size_t* written);

const char* te_last_error(void);
const char* te_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TRAVGPU_H */
