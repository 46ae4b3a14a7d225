#include "travgpu_plugins/DeviceMap.hpp"

namespace travgpu_plugins {

DeviceMap& DeviceMap::instance() {
  static DeviceMap d;
  return d;
}

DeviceMap::DeviceMap() : ctx_(nullptr), rows_(0), cols_(0), start_row_(0), start_col_(0), res_(0), px_(0), py_(0) {}

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

bool DeviceMap::upload(const grid_map::GridMap& map, const std::string& layer, int te_layer) {
  if (!map.exists(layer)) {
    error_ = "input layer '" + layer + "' is missing";
    return false;
  }
  if (start_row_ || start_col_) return check(te_upload_layer_circular(ctx_, te_layer, map.get(layer).data(), 0, start_row_, start_col_));
  return check(te_upload_layer(ctx_, te_layer, map.get(layer).data(), 0, 1));
}

bool DeviceMap::runFilter(int filter) { return check(te_run_filter(ctx_, filter, 0)); }
bool DeviceMap::runChain(unsigned flags) { return check(te_run_chain(ctx_, flags)); }

bool DeviceMap::download(grid_map::GridMap& map, const std::string& layer, int te_layer) {
  if (start_row_ || start_col_) return check(te_download_layer_circular(ctx_, te_layer, map.get(layer).data(), 0, start_row_, start_col_));
  return check(te_download_layer(ctx_, te_layer, map.get(layer).data(), 0, 1));
}

}  // namespace travgpu_plugins
