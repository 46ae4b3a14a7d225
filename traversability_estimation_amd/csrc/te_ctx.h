// te_ctx.h -- the context behind the C-ABI (struct te_ctx) and what the translation units of the shim share:
//   te_shim.hip      context, parameters, geometry, tables, the launch entry points (te_run_*), te_sync, timing
//   te_transfer.hip  uploads / downloads: whole layers, tiles, circular-buffer order, GridMap messages, prefetch, pinning
//   te_paths_api.hip path checks, inclination, polygon footprint layers (N2, N3)
//   te_multi.hip     the batch axis over several contexts / devices (te_shard_range, te_bcast_params over RCCL, *_multi)
// No CPU fallback of any kind: without a gfx950 device te_create() fails with TE_ERR_NO_DEVICE.
#pragma once
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <dlfcn.h>
#include <functional>
#include <chrono>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include <cstdlib>
#include "te_internal.h"
#include "te_msg.h"

struct te_ctx;

namespace te {
namespace shim {
// records the message te_last_error() returns (thread-local) and hands `code` back
int fail(int code, const char* fmt, ...);
}  // namespace shim
}  // namespace te

#define HIP_TRY(expr)                                                                            \
  do {                                                                                           \
    hipError_t e__ = (expr);                                                                     \
    if (e__ != hipSuccess) {                                                                     \
      (void)hipGetLastError(); /* the runtime's last-error slot is sticky: later launches check it */ \
      return ::te::shim::fail(TE_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e__));                          \
    }                                                                                            \
  } while (0)

struct te_ctx {
  std::mutex mu;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipStream_t aux_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  te_params params;
  bool have_params = false, have_geo = false, have_elev = false, chain_done = false, footprint_done = false;
  float* poly_x = nullptr;  // traversability_x / traversability_rot (one allocation, made by the first te_run_polygon_footprint)
  float* poly_rot = nullptr;
  float* robot_slope = nullptr;  // layer robot_slope (checkInclination); allocated by its first upload, NaN until written
  bool have_robot_slope = false, check_inclination = false;  // footprint/check_robot_inclination (:114)
  unsigned* poly_stream = nullptr;        // offset tables of the two footprint polygons (device copy)
  size_t poly_stream_cap = 0;             // in words
  std::vector<unsigned> poly_stream_host;  // stays alive until the asynchronous upload has been consumed
  te::Geo geo;
  te::ChainParams cp;
  te::FootprintParams fp;
  te::Layers L;
  size_t layer_elems = 0;
  void* slab = nullptr;
  int16_t* d_spiral = nullptr;
  int* clip_table = nullptr;
  int* fp_clip_table = nullptr;
  bool combine_deferred = false;
  // the traversability layer was written from outside (upload, device pointer, a per-plugin combine of uploaded scores):
  // its values are then not bounded by the weights, and the fixed-point footprint kernel must not be used
  bool trav_external = false;
  // te_device_ptr handed out the traversability layer: the caller may write it at any time from then on, so only a
  // footprint pass that runs right behind a chain that rewrote EVERY cell (te_run_chain with the footprint flag) may
  // still assume the bound; te_run_footprint and region runs take the double kernel.  Reset with the layers.
  bool trav_ptr_out = false;
  // te_set_option: choices between kernels that give identical results (tests reach both; never read from the environment)
  int opt_fb_walk = 0, opt_fb_blocks_per_cu = 0, opt_polygon_per_cell = 0, opt_graph = 0, opt_bcast_rccl = 0, opt_rank_rule = 0;
  // invalid cells of the elevation layer as of the last whole upload (-1: unknown -- tiles, device pointer): see sparse_holes()
  long long invalid_cells = -1;
  long long invalid_runs = -1;  // runs of invalid cells in memory order (k_count_invalid); meaningful with invalid_cells >= 0
  unsigned long long* d_count = nullptr;
  char* hole_queue = nullptr;  // scratch of k_normals3's sparse-hole march (allocated when a launch first picks it)
  float* tie_scratch = nullptr;  // one float per cell: the step filter at a tie radius (allocated when a launch first needs it, freed with the layers)
  bool tables_ready = false;
  // the circular-footprint tables are built separately: a footprint this build cannot handle (more than 20 cells) must
  // not stop the filter chain or the per-plugin entry points, which never use them (the reference has no such coupling)
  bool fp_tables_ready = false;
  int fp_tables_rc = TE_OK;
  char fp_tables_err[256] = "";
  // the launch sequence of a whole-map run, captured once per (flags, parameters, geometry) and replayed
  static constexpr int kGraphs = 4;  // one per flag combination in use
  hipGraphExec_t graph_exec[kGraphs] = {nullptr, nullptr, nullptr, nullptr};
  unsigned graph_flags[kGraphs] = {0, 0, 0, 0};
  int graph_next = 0;
  bool graph_ok = true;  // cleared after a failed capture: direct launches from then on
  // streaming tiles (te_upload_tile_async / te_download_tile_async): copy streams, two device staging slots each way
  struct TileSlot {
    float* buf = nullptr;
    size_t cap = 0;                              // in floats
    hipEvent_t ready = nullptr, freed = nullptr;  // filled / consumed
    bool used = false;
  };
  hipStream_t in_stream = nullptr, out_stream = nullptr;
  TileSlot in_slot[2], out_slot[2];
  int in_next = 0, out_next = 0;
  bool tiles_pending = false;  // te_sync has copy streams to wait for
  te::HostStager stager;           // whole-layer transfers through pageable host buffers (te_stage.hip)
  // te_prefetch_layers: whole-layer uploads on a thread of their own, through a second staging ring and the second copy
  // pool, beside whatever the caller does meanwhile (a filter on other layers, the download of its output)
  te::HostStager prefetcher;
  hipStream_t prefetch_order = nullptr;  // stands in for the compute stream of HostStager::upload
  // (one worker per context, started by the first prefetch and kept: a new thread's first HIP call pays the runtime's
  // per-thread set-up, milliseconds that a 3 ms transfer cannot afford)
  std::thread prefetch_thread;
  std::mutex pf_mu;
  std::condition_variable pf_cv;
  std::function<void()> pf_job;
  bool pf_quit = false;
  bool prefetch_running = false, prefetch_elev = false;  // (prefetch_running: a job is queued or being worked on; under pf_mu)
  // bit TE_LAYER_* of every layer the prefetch in flight is writing (under mu): a call that runs beside a prefetch joins
  // it first if it reads or writes one of them (te_run_filter, te_download_layer*)
  unsigned prefetch_mask = 0;
  std::atomic<int> prefetch_rc{TE_OK};
};

namespace te {
namespace shim {
// joins a running prefetch (caller holds c->mu); its result stays in c->prefetch_rc until te_wait_prefetch reports it
void finish_prefetch_locked(te_ctx* c);
// Every entry point takes the context's mutex through this: a prefetch that is still running is finished first -- except
// in the calls that are meant to run beside one (te_run_filter, te_download_layer*, the parameter calls).
struct CtxLock {
  std::lock_guard<std::mutex> lk;
  // beside_prefetch: the call may run while a prefetch is in flight -- unless it touches one of the layers the prefetch is
  // writing (`touches`: bits TE_LAYER_*), in which case it joins it like every other call
  explicit CtxLock(te_ctx* c, bool beside_prefetch = false, unsigned touches = 0) : lk(c->mu) {
    if (!beside_prefetch || (touches & c->prefetch_mask)) finish_prefetch_locked(c);
  }
};
constexpr unsigned bit(int layer) { return (layer >= 0 && layer < 32) ? 1u << layer : 0u; }
// counts the invalid cells of the whole elevation layer on the context's stream and waits for the result
int count_invalid_elevation(te_ctx* c);
float* layer_ptr(te_ctx* c, int layer);
int ensure_input_layer(te_ctx* c, int layer);
int rebuild_tables(te_ctx* c);
void rebuild_footprint_tables(te_ctx* c);
int sync_tiles(te_ctx* c);  // waits for the copy streams of the streaming-tile calls
int run_whole_locked(te_ctx* c, unsigned flags);
}  // namespace shim
}  // namespace te
