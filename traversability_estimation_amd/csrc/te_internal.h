// te_internal.h -- shared between the C-ABI shim (te_shim.hip) and the gfx950 kernels (te_kernels.hip).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <vector>

#include "travgpu.h"

namespace te {

// Measurement switches: environment variables read once per process, in a LAB build only (-DTE_LAB: build.py --lab ->
// libtravgpu_lab.so, which tools/ load through TRAVGPU_LIB).  In the shipped library they are compile-time constants:
// nothing the product does depends on the environment.  (Choices between kernels with identical results that the tests
// need to reach are part of the C-ABI instead: te_set_option.)
#ifdef TE_LAB
inline int lab_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
inline bool lab_flag(const char* name) { return getenv(name) != nullptr; }
#else
constexpr int lab_int(const char*, int dflt) { return dflt; }
constexpr bool lab_flag(const char*) { return false; }
#endif

// A roctx range over the enclosing scope (te_trace.hip): the phases of a launch in `rocprofv3 --marker-trace`.
class TraceRange {
 public:
  explicit TraceRange(const char* name);
  ~TraceRange();
  TraceRange(const TraceRange&) = delete;
  TraceRange& operator=(const TraceRange&) = delete;

 private:
  bool live_;
};
bool trace_available();

constexpr int kMaxRadiusCells = 32;  // largest stencil radius (in cells) a launch supports
// The marching kernels of te_march5.h load the rows above / below a map unconditionally (and stage them as "nothing
// there"): up to the stencil radius above the first row, and the radius plus the prefetch distance below the last.  The
// context's slab therefore starts and ends with this many rows of slack (te_set_geometry).
constexpr int kSlabGuardRows = kMaxRadiusCells + 16;
constexpr int kMaxTies = 32;         // offsets lying exactly on the circle (tie radii)
constexpr int kMaxSpiral = 4096;     // ordered offsets of the footprint spiral

// Map geometry as the kernels need it.  Device layout of every layer: [batch][cols][rows] float32,
// i.e. grid_map's column-major matrix; the FAST axis is the grid_map row index i ("x" below).
struct Geo {
  int rows, cols, batch;
  double res;
  double ax, ay;  // position of cell 0: pos + (0.5*len - 0.5*res); x(i) = ax + res*(-i)
  double len_x, len_y, pos_x, pos_y;
};

// A disc {(di,dj): di^2+dj^2 <= (radius/res)^2} as row runs: for column offset dj (|dj|<=R) the cells
// di in [-hw[|dj|], hw[|dj|]].  Offsets whose squared norm equals (radius/res)^2 up to rounding are
// NOT in the runs: whether CircleIterator keeps them depends on the double rounding of the cell
// positions, so they are listed in `tie_*` and tested per cell with the reference's own formula.
struct Disc {
  int R;                         // largest |offset| in the runs (-1: empty disc)
  int hw[kMaxRadiusCells + 1];   // -1 where the run is empty
  int n_ties;
  int8_t tie_di[kMaxTies], tie_dj[kMaxTies];
  double r2;                     // radius*radius (double, as CircleIterator computes it)
  int reach;                     // max(R, max |tie offset|)
  int npoints;                   // number of cells in the runs (full disc, away from borders)
  int Q;                         // tie-free discs: largest a^2+b^2 <= (radius/res)^2 (names the shape); -1 with ties
};

struct ChainParams {
  Disc normals, rough, step1, step2;
  int same_rough_disc;  // roughness radius selects the same cells as the normals radius
  int axis;             // normal_vector_positive_axis
  int rank_rule;        // TE_OPT_NORMALS_RANK_RULE: a rank-deficient scatter matrix gives UnitZ (generic normals kernel)
  double slope_crit, step_crit, rough_crit;
  int step_ncrit;
  float w_scale, w_slope, w_step, w_rough;
};

// THE CLIP OF A SCORE.  SlopeFilter / RoughnessFilter clip their scores at 0 (SlopeFilter.cpp:77-81, RoughnessFilter.cpp:119-124)
// and the footprint checks treat "score == 0" as a category of its own: checkForSlope / checkForRoughness run -- and
// memoise -- only where the score IS 0 (TraversabilityMap.cpp:869, :897).  A fast tail whose score lands within its own
// error of the clip therefore does not decide the cell: it writes the kExactNaN payload into the slope layer and flags the
// tile, and the fix-up pass settles the cell with the generic arithmetic (gather in the generic order, cyclic Jacobi,
// double acos: bit-identical to the generic kernels and the oracle).  The bands are the fast tails' error bounds in score
// units with a margin of 3, from the parameters (host side):
//   slope:  (float32 acos polynomial + float rounding: 4e-7 rad  +  one float32 ulp of nz: 1.2e-7 / sin(crit) rad) / crit
//   roughness: float32 square root, product and 1/crit: 3 * 2^-23, relative to a score distance of 1 at the clip
constexpr unsigned kExactNaNBits = 0x7fc00001u;  // slope (given normals: roughness) of a cell the fix-up pass must settle exactly
inline float clip_band_slope(double crit) {
  if (!(crit > 0.0)) return 0.0f;
  double s = sin(crit);
  s = s > 1e-3 ? s : 1e-3;
  const double e = 3.0 * (4e-7 + 1.2e-7 / s) / crit;
  return (float)(e < 2e-6 ? 2e-6 : (e > 1e-2 ? 1e-2 : e));
}
inline float clip_band_rough(double crit) { return crit > 0.0 ? 2e-6f : 0.0f; }

constexpr unsigned kDeferCombine = 0x8000u;  // internal launch_chain flag: the footprint mask kernel will write `traversability`

struct Region {  // half-open cell rectangle of one map (or all maps when map < 0)
  int map, i0, j0, i1, j1;
};

struct FootprintParams {
  Disc slope_disc;   // circle(3*res) of checkForSlope / checkForRoughness
  Disc step_disc;    // circle(2.5*res) of checkForStep
  Disc fp_disc;      // circle(radiusMax) of the spiral
  int reach;         // largest |offset| the spiral visits
  int ncrit_slope, ncrit_rough;
  double rmin, rmax, def, max_gap, crit_step;
  int check_rough;
  int n_spiral;      // entries of the ordered spiral table
};

struct Layers {
  float* elev;
  float* slope;
  float* step;
  float* rough;
  float* trav;
  float* footprint;
  float* nx;
  float* ny;
  float* nz;
  float* slope_fp;
  float* step_fp;
  float* rough_fp;
  float* step_height;  // temp layer of StepFilter (never leaves the device)
  uint8_t* untrav;     // !isTraversableForFilters per cell
  float* tie_scratch;          // one float per cell: the step filter at a tie radius (te_fast_step.hip); nullptr: not allocated
  unsigned* fp_blocked;        // k_fp_slide4's list of cells whose disc holds an untraversable cell (one entry per cell at most) ...
  unsigned* fp_blocked_count;  // ... [0] entries of the list (some hold kF4NoCell), [1] cells listed, [3] first spiral entry that can be untraversable
                               // in a listed disc, [4] entries of the scratch reserved (k_fp_mask resets [0], [1], [4])
  unsigned* fp_scratch;        // k_fp_slide5 collects a block's cells here (one reservation per block, sized for its whole strip) and copies
                               // them to the list when the strip is done: the list stays dense whatever the reservations
  size_t fp_blocked_cap;       // entries the list holds (cells + fast::f4_list_slack)
  int* block_flags;    // one flag per block of the shape-specialised normals kernel ("needs the fix-up pass")
  uint8_t* untrav_flags;  // one byte per 64 x 4 cells: "holds an untraversable cell" as of the mask kernel's last pass over them (1 until then);
                          // k_fp_slide5 does not fetch the mask bytes of a strip whose flags are all clear
  int* clip_table;     // x/y moments of the normals disc clipped by the map border (build_clip_table)
  // second stream + fork/join events: step filter || normals kernel on whole-map runs (nullptr: sequential)
  hipStream_t aux_stream;
  hipEvent_t ev_fork, ev_join;
  int fb_walk, fb_blocks_per_cu;  // te_set_option: k_fp_blocked's walk (0: by the length of the list, 1: per wavefront, 2: per lane) and grid (0: default)
  int sparse_holes;  // 1: at most a few per mille of the elevation cells are invalid (counted at upload): k_normals3 takes its sparse march
  int no_holes;      // 1: none of them is (counted at upload): the clean march alone, on its slim ring (k_normals3s)
  int skip_clean;    // 1: scattered invalid cells, several per strip on average (sparse march): the clean march is not attempted first
  int short_strips;  // 1: invalid cells counted, too many for the sparse march -- unobserved regions as a rule: k_normals3 cuts the map
                     // into strips of 32 rows, more blocks than resident slots (a strip along the edge of a region takes 2.6x
                     // the time of a clean one and the pass lasted as long as its slowest strip: te_normals3.hip, launch3)
  char* hole_queue;  // its scratch: normals_hole_queue_bytes() (te_normals3.hip), nullptr: the dense march serves
};

// polygon footprints (te_polygon.hip)
struct PolygonArgs {
  int n;
  double def;                                       // traversabilityDefault_
  double off[2][2 * TE_MAX_POLYGON_VERTICES];       // vertex offsets from the centre cell: [0] as given, [1] turned by yaw
};
// offset table of one footprint polygon (build_polygon_table): rows of the bounding box that hold cells, each a header
// word (di index | items << 8) followed by its items (dj index | run length << 8 | "decided per cell" << 31)
struct PolygonTable {
  int first;              // index of the first word in the stream
  int n_rows;
  int n_uncertain;
  int di_min, dj_min;     // offset of table index 0
  int di_span, dj_span;   // extent of the staged tile in offsets
};
struct PolygonTables {
  PolygonTable t[2];
};
bool build_polygon_table(const Geo& g, int n, const double* off, std::vector<unsigned>& stream, PolygonTable& tb);
hipError_t launch_polygon_footprint_table(const Geo& g, const PolygonArgs& a, const PolygonTables& tabs, const unsigned* d_stream,
                                          const float* trav, const uint8_t* untrav, float* out_x, float* out_rot,
                                          hipStream_t stream);
void rotate_footprint(int n_points, const double* points_xy, double yaw, double* out_xy);
hipError_t launch_polygon_footprint(const Geo& g, const PolygonArgs& a, const float* trav, const uint8_t* untrav, float* out_x,
                                    float* out_rot, hipStream_t stream);
hipError_t launch_polygons_traversable(const Geo& g, double def, int n_polygons, const int* vertex_offset, const double* vertex_xy,
                                       const float* trav, const uint8_t* untrav, unsigned char* is_traversable,
                                       double* traversability, hipStream_t stream);

// isTraversable(polygon, computeUntraversablePolygon = true): per row index of the map (x, y first, y last, y between,
// count) of the polygon's untraversable cells (rows5: 5 doubles x g.rows), and the hull the host builds from them
hipError_t launch_polygon_untraversable_rows(const Geo& g, int n, const double* vertex_xy, const uint8_t* untrav, double* rows5,
                                             hipStream_t stream);
void untraversable_hull_from_rows(int rows, const double* rows5, std::vector<double>& hull_xy);

// the polygons checkPolygonalFootprintPath evaluates for a batch of paths (host side), in path order
constexpr int kMaxPathPolygonVertices = 1024;
struct PathPolygons {
  std::vector<int> vertex_offset;     // polygon p: vertices [vertex_offset[p], vertex_offset[p+1])
  std::vector<double> vertex_xy;
  std::vector<double> area;           // polygon.getArea()
  std::vector<double> area_previous;  // polygon1.getArea() of the same iteration (:574)
  std::vector<int> first, count, status;  // per path
};
void build_path_polygons(int n_paths, const int* pose_offset, const double* poses, int n_points, const double* points_xyz,
                         const unsigned char* conservative, PathPolygons& out);

// k_normals_fixup: every workgroup owns kFixTiles tiles that are fix_groups() apart (flagged tiles come in runs
// and must spread over many workgroups) and whose flags are adjacent in memory (one coalesced load).
// Up to 2048 tiles (a 1024 x 512 map) every tile has a workgroup of its own: a handful of flagged tiles queueing up in
// three workgroups was half of the chain's latency on the reference's own 100 x 133 map.
constexpr int kFixTiles = 8;
inline int fix_groups(int ntiles) { return ntiles <= 2048 ? ntiles : (ntiles + kFixTiles - 1) / kFixTiles; }

struct FastGrid {  // fix-up flag grid of the last sliding-disc normals launch: 64x16 tiles of the region
  int ntx, nty, nbz;
  int frame;  // > 0: the fix-up pass also owns every cell within `frame` cells of the map border (k_normals3 computes
              // only discs that lie inside the map), whatever the flags and the slope layer say;
              // < 0: the kernel settled every cell itself (k_normals_small): no fix-up pass at all
};

// te_stage.hip: whole-layer transfers through pageable caller buffers -- a ring of page-locked slots per context, chunked,
// the host-side copies on a process-wide pool of threads beside the DMA of the neighbouring chunk
struct HostStager {
  static constexpr size_t kChunk = (size_t)8 << 20, kMinBytes = (size_t)4 << 20;
  static constexpr int kSlots = 4;
  char* slot[kSlots] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev[kSlots] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_order = nullptr;
  hipStream_t stream = nullptr;
  int pool = 0;  // which of the process' copy-thread pools serves this ring (te_stage.hip)
  ~HostStager();
  void release();      // (with the context's device current)
  hipError_t ensure();
  // host -> device behind everything queued on `compute`; `compute` is made to wait for the data; returns once `host` may be reused
  hipError_t upload(void* dev, const void* host, size_t bytes, hipStream_t compute);
  // device -> host behind everything queued on `compute`; returns once `host` holds the data
  hipError_t download(void* host, const void* dev, size_t bytes, hipStream_t compute);
};

// The marching kernels of te_march5.h read kSlabGuardRows rows above and below the layers they are given (unconditional
// raw buffer loads, no bounds check: DESIGN.md).  Layers of a context's slab have that slack by construction; any other
// pointer -- a caller's own allocation handed in through a future entry point -- is checked here against the allocation
// that holds it, and the launchers fall back to the bounds-checked generic kernels when the slack is not there.
// (One driver query per distinct (pointer, shape); the verdicts are dropped whenever a context frees its layers.)
inline std::atomic<unsigned>& guard_cache_generation() {
  static std::atomic<unsigned> g{0};
  return g;
}
inline bool layer_has_guard_rows(const void* p, const Geo& g, size_t elem_bytes) {
  struct Entry {
    const void* p;
    int rows, cols, batch;
    unsigned gen;
    bool ok;
  };
  static thread_local Entry cache[8] = {};
  static thread_local int next = 0;
  const unsigned gen = guard_cache_generation().load(std::memory_order_acquire);
  for (const Entry& e : cache)
    if (e.p == p && e.rows == g.rows && e.cols == g.cols && e.batch == g.batch && e.gen == gen && p) return e.ok;
  bool ok = false;
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  if (p && hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) == hipSuccess) {
    const size_t guard = (size_t)kSlabGuardRows * (size_t)g.rows * elem_bytes;
    const size_t layer = (size_t)g.rows * (size_t)g.cols * (size_t)(g.batch > 0 ? g.batch : 1) * elem_bytes;
    const char* lo = (const char*)base;
    const char* hi = lo + size;
    ok = (size_t)((const char*)p - lo) >= guard && (const char*)p + layer <= hi && (size_t)(hi - ((const char*)p + layer)) >= guard;
  } else {
    (void)hipGetLastError();
  }
  cache[next] = Entry{p, g.rows, g.cols, g.batch, gen, ok};
  next = (next + 1) & 7;
  return ok;
}

// Compute units of the CURRENT device (hipGetDevice), cached per device; the launchers size their grids from it.
// (A function-local static of the first device's count would be a data race with one context per thread and the
// wrong capacity on a node with different devices.)
inline int device_cus() {
  static std::atomic<int> cache[64];
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess) return cus;
  if (dev >= 0 && dev < 64) {
    const int v = cache[dev].load(std::memory_order_relaxed);
    if (v > 0) return v;
  }
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  if (dev >= 0 && dev < 64) cache[dev].store(cus, std::memory_order_relaxed);
  return cus;
}

// launch wrappers (te_kernels.hip); all asynchronous on `stream`
hipError_t launch_filter(const Geo& g, const ChainParams& p, const Layers& L, int filter, unsigned flags,
                         hipStream_t stream);
hipError_t launch_chain(const Geo& g, const ChainParams& p, const Layers& L, const Region& r, unsigned flags,
                        hipStream_t stream);
// spiral_table: [n_spiral][4] int16 {di, dj, ring, tie}; clip_table: build_clip_table(fp_disc, reach)
// trav_cap: upper bound of the finite traversability values if the layer was written by the chain (see footprint_slide4), else < 0
// region (nullptr: all maps, all cells): the cells whose scores changed; the mask is recomputed within 3 cells of them
// and the footprint within the footprint's reach of those (a footprint shape only the general kernel serves: on every cell)
hipError_t launch_footprint(const Geo& g, const FootprintParams& p, const Layers& L, const int16_t* spiral_table,
                            const int* clip_table, bool write_memo, const ChainParams* combine, double trav_cap, hipStream_t stream,
                            const Region* region = nullptr);
int chain_max_reach(const ChainParams& p);
// the flag grid of Layers::untrav_flags: one byte per 64 x 4 cells of every map
inline int untrav_flag_ntx(int rows) { return (rows + 63) / 64; }
inline int untrav_flag_nfy(int cols) { return (cols + 3) / 4; }
inline size_t untrav_flag_bytes(int rows, int cols, int batch) { return (size_t)untrav_flag_ntx(rows) * (size_t)untrav_flag_nfy(cols) * (size_t)batch; }
// te_paths.hip: checkCircularFootprintPath for a batch of paths on the (complete) footprint layer of one map;
// robot_slope: the layer checkInclination reads (nullptr: footprint/check_robot_inclination off)
hipError_t launch_check_circular_paths(const Geo& g, const float* footprint, double fp_default, const float* robot_slope,
                                       int n_paths, const int* pose_offset, const double* pose_xy, unsigned char* is_safe,
                                       double* traversability, int* status, hipStream_t stream);
// batched checkInclination(start, end): segment k = start_end_xy[4k .. 4k+4)
hipError_t launch_check_inclination(const Geo& g, const float* robot_slope, int n, const double* start_end_xy,
                                    unsigned char* ok, int* status, hipStream_t stream);

// shape-specialised kernels (te_fast_*.hip); return false when the shape Q is not instantiated
namespace fast {
bool step_height_fast(int Q, const Geo& g, const float* elev, float* sh, const Region& r, hipStream_t s);
bool step_score_fast(int Q, const Geo& g, double crit, int ncrit, const float* sh, float* out, const Region& r,
                     hipStream_t s);
// te_step5.hip: the marching kernels behind the two functions above (and, RAW -- second output pointer set -- behind the two below)
bool step_height5(int Q, const Geo& g, const float* elev, float* sh, float* sh_min, const Region& r, hipStream_t s);
bool step_score5(int Q, const Geo& g, double crit, int ncrit, const float* sh, float* out, float* out_count, const Region& r, hipStream_t s);
// ... at a tie radius (whole-cell radii of 2 .. 10 cells): the marching kernels on the shape without its circle, then the
// accepted circle cells folded in per cell; scratch: one float per cell of the layer (Layers::tie_scratch)
bool step_height_ties(const Disc& d, const Geo& g, const float* elev, float* sh, float* scratch, const Region& r, hipStream_t s);
bool step_score_ties(const Disc& d, const Geo& g, double crit, int ncrit, const float* sh, float* out, float* scratch, const Region& r,
                     hipStream_t s);
// normals + slope + roughness (same disc for normals and roughness, positive axis z); with `combine`
// the traversability layer is written too (the step layer must be complete).
// *combined tells whether it did (the k_normals3 path leaves the combine to the caller).
bool normals_fast(const Geo& g, const ChainParams& p, const Layers& L, bool keep_normals, bool combine,
                  const Region& r, int* block_flags, const int* clip_table, FastGrid* fg, hipStream_t s, bool* combined);
// te_slide_normals.hip: RoughnessFilter alone with the normals of the layers (false: shape / map not taken)
bool roughness_given_fast(const Geo& g, const ChainParams& p, const Layers& L, const Region& r, int* block_flags, FastGrid* fg, hipStream_t s);
// te_normals3.hip: the cells whose disc lies inside the map (false: shape / region not taken)
int footprint_inner_q(double res, double rmin, double rmax);  // te_footprint3.hip
bool normals_fast3(const Geo& g, const ChainParams& p, const Layers& L, bool keep_normals, const Region& r, int* block_flags,
                   FastGrid* fg, hipStream_t s);
// te_normals_small.hip: discs that reach at most two cells (the one-cell tie radius of the default parameters on a 0.05 m
// map; small launches of the tie-free 5- to 13-point discs), one cell per thread; false: not taken
// write_step: both step windows hold one cell -- the step score is 1 (valid) / NaN and this kernel writes the layer; combine:
// ... and the weighted sum (the step layer must be complete, or written here)
bool normals_small(const Geo& g, const ChainParams& p, const Layers& L, bool keep_normals, const Region& r, int* block_flags, FastGrid* fg,
                   hipStream_t s, bool write_step = false, bool combine = false);
// ... and the whole chain of a small whole-map launch (at most 2^18 cells, a disc of at most 13 cells, step windows of at most 3 x 3
// cells) in one kernel: normals, slope, roughness, both step passes, the weighted sum; false: not taken
bool chain_window(const Geo& g, const ChainParams& p, const Layers& L, bool keep_normals, const Region& r, bool combine, hipStream_t s);
int normals_fast_max_blocks(const Geo& g);
size_t normals_hole_queue_bytes();  // te_normals3.hip: scratch of the sparse-hole march for one device (any map, any batch)
// te_footprint3.hip: the sliding-sum kernel of the circular footprint pass (false: shape / map not taken)
// region: the output cells to compute (whole block columns and the rows [j0, j1) of map `map`); nullptr: every map, every cell
bool footprint_slide3(const Geo& g, const FootprintParams& p, const Layers& L, const int16_t* spiral_table, const int* clip_table,
                      hipStream_t s, const Region* region = nullptr);
// te_footprint4.hip: the same on 32-bit fixed point, when the values of the traversability layer are bounded by tcap
// (tcap < 0: no bound known)
bool footprint_slide4(const Geo& g, const FootprintParams& p, const Layers& L, const int16_t* spiral_table, const int* clip_table,
                      double tcap, hipStream_t s, const Region* region = nullptr, bool finish = true);
// te_footprint5.hip: the same sum in scatter form (tie-free discs up to 15 cells); *needs_blocked: the caller owes
// footprint_blocked4 for the listed cells (after every launch of the pass)
bool footprint_slide5(const Geo& g, const FootprintParams& p, const Layers& L, const int* clip_table, double tcap, hipStream_t s,
                      const Region* region, bool* needs_blocked);
constexpr int kClipInts = 6 * (2 * kMaxRadiusCells + 1) * (2 * kMaxRadiusCells + 1);  // one clip table of the normals disc; for a tie radius the
                                                                                      // table of the disc with its circle follows, then the packed offsets
constexpr int kFpClipInts = 6 * 41 * 41;  // one clip table of the footprint disc (reach <= 20); a second one follows it for a tie
                                           // radius: the disc with the cells on its circle (k_fp_slide4<Q, true>)
constexpr int kF4Chunk = 256;              // entries of the list a block reserves at a time
constexpr unsigned kF4NoCell = 0xffffffffu;  // an unused entry
size_t f4_list_slack(int rows, int cols, int batch);
// ... its second half: the cells whose disc holds an untraversable cell (finish = false above: the caller runs it, after
// every k_fp_slide4 launch of the pass has completed)
void footprint_blocked4(const Geo& g, const FootprintParams& p, const Layers& L, const int16_t* spiral_table, hipStream_t s);
void build_clip_table(const Disc& d, int Rk, int* out);  // (2*Rk+1)^2 * 6 ints, clip codes relative to radius Rk
}  // namespace fast

}  // namespace te
