#include "travgpu_plugins/DeviceMap.hpp"

#include <cstdlib>
#include <cstring>

namespace travgpu_plugins {

DeviceMap& DeviceMap::instance() {
  static DeviceMap d;
  return d;
}

DeviceMap::DeviceMap()
    : ctx_(nullptr), rows_(0), cols_(0), start_row_(0), start_col_(0), res_(0), px_(0), py_(0), uploads_(0), uploads_skipped_(0), prefetching_(false) {
  for (int k = 0; k < kLayers; ++k) {
    memo_[k] = HashMemo{nullptr, 0, 0, 0};
    unused_prefetches_[k] = 0;
  }
  forget();
}

void DeviceMap::forget() {
  for (int k = 0; k < kLayers; ++k) {
    resident_[k].valid = false;
    memo_[k].data = nullptr;
  }
}

namespace {
// Checksum of a layer's bit patterns.  The default reads EVERY cell (four independent multiply-xorshift lanes over
// 64-bit words: memory speed, ~10 ms for a 4096^2 layer against 30 ms for its pageable upload): a cell edited in place
// under an unchanged time stamp (inpainting, a local elevation update, stamp 0) must never be taken for the resident
// layer.  sampled (TRAVGPU_PLUGIN_HASH=sampled, for hosts that guarantee a new stamp per content): ~4096 cells at an
// odd stride -- an even stride of n/4096 visits one row of a column-major power-of-two map only.
uint64_t mix64(uint64_t h, uint64_t w) {
  h = (h ^ w) * 0x9E3779B97F4A7C15ull;
  return h ^ (h >> 29);
}
uint64_t hash_layer(const float* d, size_t n, bool sampled) {
  uint64_t h0 = 0x243F6A8885A308D3ull, h1 = 0x13198A2E03707344ull, h2 = 0xA4093822299F31D0ull, h3 = 0x082EFA98EC4E6C89ull;
  if (sampled && n > 8192) {
    const size_t step = (n / 4096) | 1;
    for (size_t k = 0; k < n; k += step) {
      uint32_t w;
      memcpy(&w, d + k, 4);
      h0 = mix64(h0, w);
    }
    uint32_t w;
    memcpy(&w, d + (n - 1), 4);
    return mix64(h0, w) ^ (uint64_t)n;
  }
  const unsigned char* b = reinterpret_cast<const unsigned char*>(d);
  const size_t words = n / 2;  // 64-bit words
  size_t k = 0;
  for (; k + 4 <= words; k += 4) {
    uint64_t w[4];
    memcpy(w, b + 8 * k, 32);
    h0 = mix64(h0, w[0]);
    h1 = mix64(h1, w[1]);
    h2 = mix64(h2, w[2]);
    h3 = mix64(h3, w[3]);
  }
  for (size_t c = 2 * k; c < n; ++c) {  // the tail, cell by cell
    uint32_t w;
    memcpy(&w, d + c, 4);
    h0 = mix64(h0, w);
  }
  return mix64(mix64(mix64(h0, h1), h2), h3) ^ (uint64_t)n;
}
bool hash_sampled() {
  static const bool v = getenv("TRAVGPU_PLUGIN_HASH") && !strcmp(getenv("TRAVGPU_PLUGIN_HASH"), "sampled");
  return v;
}
}  // namespace

DeviceMap::~DeviceMap() {
  if (ctx_) te_destroy(ctx_);
}

bool DeviceMap::check(int rc) {
  if (rc == TE_OK) return true;
  error_ = te_last_error();
  return false;
}

bool DeviceMap::prepare(const grid_map::GridMap& map) {
  if (!ctx_ && !check(te_create(0, &ctx_))) return false;
  // a moved map is a circular buffer; the copies to and from the device undo / redo the rotation
  start_row_ = map.getStartIndex()(0);
  start_col_ = map.getStartIndex()(1);
  const int rows = map.getSize()(0), cols = map.getSize()(1);
  const double res = map.getResolution(), px = map.getPosition().x(), py = map.getPosition().y();
  if (rows != rows_ || cols != cols_ || res != res_ || px != px_ || py != py_) {
    if (!check(te_set_geometry(ctx_, rows, cols, 1, res, px, py))) return false;
    rows_ = rows; cols_ = cols; res_ = res; px_ = px; py_ = py;
    forget();
  }
  return true;
}

bool DeviceMap::params(te_params& p) {
  if (!ctx_ && !check(te_create(0, &ctx_))) return false;
  return check(te_get_params(ctx_, &p));
}

bool DeviceMap::setParams(const te_params& p) {
  if (!ctx_ && !check(te_create(0, &ctx_))) return false;
  return check(te_set_params(ctx_, &p));
}

bool DeviceMap::setOption(int option, int value) {
  if (!ctx_ && !check(te_create(0, &ctx_))) return false;
  return check(te_set_option(ctx_, option, value));
}

bool DeviceMap::upload(const grid_map::GridMap& map, const std::string& layer, int te_layer) {
  if (!map.exists(layer)) {
    error_ = "input layer '" + layer + "' is missing";
    return false;
  }
  static const bool cache_on = !(getenv("TRAVGPU_PLUGIN_CACHE") && atoi(getenv("TRAVGPU_PLUGIN_CACHE")) == 0);
  const float* data = map.get(layer).data();
  const size_t n = (size_t)rows_ * cols_;
  LayerKey key = {true, (uint64_t)map.getTimestamp(), 0, n, start_row_, start_col_, false};
  if (cache_on && te_layer >= 0 && te_layer < kLayers) {
    HashMemo& mm = memo_[te_layer];
    if (mm.data == data && mm.stamp == key.stamp && key.stamp != 0 && mm.n == n)
      key.hash = mm.hash;  // the previous plugin's prefetch() read this very buffer
    else
      key.hash = hash_layer(data, n, hash_sampled());
    mm.data = nullptr;
    const LayerKey& r = resident_[te_layer];
    // (a sampled hash only stands in for the content together with a real time stamp)
    if (r.valid && (!hash_sampled() || key.stamp != 0) && r.stamp == key.stamp && r.hash == key.hash && r.n == key.n && r.start_row == key.start_row && r.start_col == key.start_col) {
      if (r.prefetched) {
        resident_[te_layer].prefetched = false;  // (counted as an upload when it was started)
        unused_prefetches_[te_layer] = 0;
      } else
        ++uploads_skipped_;
      return true;
    }
  }
  const bool ok = (start_row_ || start_col_) ? check(te_upload_layer_circular(ctx_, te_layer, data, 0, start_row_, start_col_))
                                             : check(te_upload_layer(ctx_, te_layer, data, 0, 1));
  if (te_layer >= 0 && te_layer < kLayers) {
    resident_[te_layer] = key;
    resident_[te_layer].valid = ok && cache_on;
  }
  if (ok) ++uploads_;
  return ok;
}

bool DeviceMap::prefetch(const grid_map::GridMap& map, const std::vector<std::pair<std::string, int>>& layers) {
  static const bool cache_on = !(getenv("TRAVGPU_PLUGIN_CACHE") && atoi(getenv("TRAVGPU_PLUGIN_CACHE")) == 0);
  // On by default (TRAVGPU_PLUGIN_PREFETCH=0 switches it off): the 4096^2 three-plugin sequence takes 9.2 ms with the
  // prefetches against 11.7 ms one transfer at a time (bench.py, host_path.three_plugins_runs_ms: the two forms taking turns).
  static const bool prefetch_on = !(getenv("TRAVGPU_PLUGIN_PREFETCH") && atoi(getenv("TRAVGPU_PLUGIN_PREFETCH")) == 0);
  // (a moved map -- circular buffer -- takes the rectangle copies of te_upload_layer_circular: no prefetch; without the
  // cache the next plugin would upload again anyway)
  if (!cache_on || !prefetch_on || start_row_ || start_col_ || prefetching_) return true;
  const size_t n = (size_t)rows_ * cols_;
  int ids[8];
  const float* hosts[8];
  LayerKey keys[8];
  int m = 0;
  for (const auto& l : layers) {
    if (m >= 8 || l.second < 0 || l.second >= kLayers || !map.exists(l.first)) continue;
    const float* data = map.get(l.first).data();
    if (resident_[l.second].valid && resident_[l.second].prefetched) {  // the last prefetch of this layer found no reader
      if (++unused_prefetches_[l.second] >= 2) continue;
    }
    if (unused_prefetches_[l.second] >= 2) continue;
    const LayerKey key = {true, (uint64_t)map.getTimestamp(), hash_layer(data, n, hash_sampled()), n, start_row_, start_col_, true};
    memo_[l.second] = HashMemo{data, key.stamp, key.hash, n};
    const LayerKey& r = resident_[l.second];
    if (r.valid && (!hash_sampled() || key.stamp != 0) && r.stamp == key.stamp && r.hash == key.hash && r.n == key.n && r.start_row == key.start_row &&
        r.start_col == key.start_col)
      continue;  // the device holds it
    ids[m] = l.second;
    hosts[m] = data;
    keys[m] = key;
    ++m;
  }
  if (m == 0) return true;
  if (!check(te_prefetch_layers(ctx_, m, ids, hosts))) return false;
  prefetching_ = true;
  for (int k = 0; k < m; ++k) {
    resident_[ids[k]] = keys[k];  // (valid once finishPrefetch() has succeeded; cleared there otherwise)
    ++uploads_;
  }
  return true;
}

bool DeviceMap::finishPrefetch() {
  if (!prefetching_) return true;
  prefetching_ = false;
  if (check(te_wait_prefetch(ctx_))) return true;
  forget();
  return false;
}

bool DeviceMap::noteResident(const grid_map::GridMap& map, const std::string& layer, int te_layer) {
  static const bool cache_on = !(getenv("TRAVGPU_PLUGIN_CACHE") && atoi(getenv("TRAVGPU_PLUGIN_CACHE")) == 0);
  if (!cache_on || te_layer < 0 || te_layer >= kLayers || !map.exists(layer)) return true;
  const size_t n = (size_t)rows_ * cols_;
  const LayerKey key = {true, (uint64_t)map.getTimestamp(), hash_layer(map.get(layer).data(), n, hash_sampled()), n, start_row_, start_col_, false};
  resident_[te_layer] = key;
  return true;
}

bool DeviceMap::runFilter(int filter) {
  if (filter == TE_FILTER_NORMALS)  // writes the normal layers: whatever a plugin uploaded there is gone
    resident_[TE_LAYER_NORMAL_X].valid = resident_[TE_LAYER_NORMAL_Y].valid = resident_[TE_LAYER_NORMAL_Z].valid = false;
  return check(te_run_filter(ctx_, filter, 0));
}
bool DeviceMap::runChain(unsigned flags) {
  // the fused chain writes the normal layers itself (TE_RUN_KEEP_NORMALS): whatever a plugin uploaded there is gone
  resident_[TE_LAYER_NORMAL_X].valid = resident_[TE_LAYER_NORMAL_Y].valid = resident_[TE_LAYER_NORMAL_Z].valid = false;
  return check(te_run_chain(ctx_, flags));
}

bool DeviceMap::download(grid_map::GridMap& map, const std::string& layer, int te_layer) {
  if (start_row_ || start_col_) return check(te_download_layer_circular(ctx_, te_layer, map.get(layer).data(), 0, start_row_, start_col_));
  return check(te_download_layer(ctx_, te_layer, map.get(layer).data(), 0, 1));
}

}  // namespace travgpu_plugins
