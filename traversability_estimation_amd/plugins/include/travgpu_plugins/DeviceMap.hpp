/*
 * DeviceMap.hpp -- host-side glue between a grid_map::GridMap and the C-ABI of libtravgpu.so
 * (include/travgpu.h).  One process-wide context, serialised by a mutex: the ROS node calls the
 * filter chain from AsyncSpinner threads without a lock of its own
 * (traversability_estimation/src/TraversabilityMap.cpp:203-214).
 */
#ifndef TRAVGPU_PLUGINS_DEVICEMAP_HPP
#define TRAVGPU_PLUGINS_DEVICEMAP_HPP

#include <cstdint>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include <grid_map_core/GridMap.hpp>

#include "travgpu.h"

namespace travgpu_plugins {

class DeviceMap {
 public:
  static DeviceMap& instance();
  std::mutex& mutex() { return mutex_; }
  // All calls below expect the caller to hold mutex().  They return false and fill error() on failure.
  bool prepare(const grid_map::GridMap& map);                       // (re)sets the geometry if it changed, notes the start index
  bool params(te_params& p);                                        // current parameter set (to edit and pass back)
  bool setParams(const te_params& p);
  bool setOption(int option, int value);                            // te_set_option (e.g. TE_OPT_NORMALS_RANK_RULE)
  // Uploads a layer unless the device already holds exactly this one: the reference's chain hands every plugin a deep
  // copy of the whole map (SlopeFilter.cpp:62, StepFilter.cpp:105, RoughnessFilter.cpp:76), so StepFilter and
  // RoughnessFilter both arrive with the same `elevation`, SlopeFilter and RoughnessFilter with the same
  // `surface_normal_z`.  A device layer is identified by the map's time stamp, geometry and start index plus a hash of
  // the buffer (every cell; TRAVGPU_PLUGIN_HASH=sampled reads ~4096 cells and then relies on the stamp, TRAVGPU_PLUGIN_CACHE=0
  // always uploads).
  bool upload(const grid_map::GridMap& map, const std::string& layer, int te_layer);
  // Starts the upload of layers the NEXT plugins of the reference's chain will read (those of them `map` holds and the
  // device does not): it runs beside this plugin's filter and the download of its output (te_prefetch_layers).  The
  // plugin calls finishPrefetch() before its update() returns -- `map`'s buffers belong to the caller after that.
  bool prefetch(const grid_map::GridMap& map, const std::vector<std::pair<std::string, int>>& layers);
  bool finishPrefetch();
  // after a download into `layer`: the device layer and that host buffer are the same thing
  bool noteResident(const grid_map::GridMap& map, const std::string& layer, int te_layer);
  unsigned long uploads() const { return uploads_; }               // transfers actually made / avoided (tests, logging)
  unsigned long uploadsSkipped() const { return uploads_skipped_; }
  bool runFilter(int filter);
  bool runChain(unsigned flags);
  bool download(grid_map::GridMap& map, const std::string& layer, int te_layer);
  const std::string& error() const { return error_; }

 private:
  DeviceMap();
  ~DeviceMap();
  bool check(int rc);
  std::mutex mutex_;
  te_ctx* ctx_;
  int rows_, cols_;
  int start_row_, start_col_;  // GridMap::getStartIndex() of the map last passed to prepare()
  double res_, px_, py_;
  struct LayerKey {
    bool valid;
    uint64_t stamp, hash;
    size_t n;
    int start_row, start_col;
    bool prefetched;  // put there by prefetch(): the upload() it anticipated is the same transfer, not one avoided
  };
  static const int kLayers = 16;
  LayerKey resident_[kLayers];  // what each device layer holds, as far as the plugins put it there
  bool prefetching_;
  // prefetch() hashed this buffer: the upload() of the plugin that follows sees the same buffer under the same (non-zero)
  // stamp and takes the hash from here instead of reading 64 MB again (one use; a foreign filter in between hands the next
  // plugin the chain's OTHER buffer, which is hashed in full)
  struct HashMemo {
    const float* data;
    uint64_t stamp, hash;
    size_t n;
  };
  HashMemo memo_[kLayers];
  // a prefetched layer that no upload() asked for before the next prefetch() of it: the chain has no plugin that reads it
  // (SlopeFilter without a StepFilter behind it); after two such updates the layer is no longer prefetched
  int unused_prefetches_[kLayers];
  void forget();                // after a geometry change or a launch that overwrites input layers
  unsigned long uploads_, uploads_skipped_;
  std::string error_;
};

}  // namespace travgpu_plugins
#endif
