// te_transfer.hip -- C-ABI of libtravgpu.so, the transfers (include/travgpu.h): whole layers, tiles (synchronous and on the
// copy streams), GridMap's circular-buffer order, grid_map_msgs/GridMap messages and bags, whole-layer prefetches beside
// the caller's own work, page-locking.  The context and the helpers shared with the other parts: te_ctx.h.
#include "te_ctx.h"

using namespace te;
using namespace te::shim;

extern "C" {

int te_upload_elevation(te_ctx* c, const float* host, int map0, int nmaps) {
  if (!c || !host) return fail(TE_ERR_INVALID_ARG, "te_upload_elevation: NULL");
  CtxLock lk(c);
  if (!c->have_geo) return fail(TE_ERR_NOT_READY, "te_upload_elevation: geometry not set");
  if (map0 < 0 || nmaps <= 0 || map0 + nmaps > c->geo.batch)
    return fail(TE_ERR_INVALID_ARG, "te_upload_elevation: maps [%d,%d) of batch %d", map0, map0 + nmaps, c->geo.batch);
  HIP_TRY(hipSetDevice(c->device));
  const size_t per = (size_t)c->geo.rows * c->geo.cols;
  HIP_TRY(c->stager.upload(c->L.elev + per * map0, host, per * nmaps * sizeof(float), c->stream));
  // (the count also waits for the copy: the host buffer may be reused as soon as we return)
  if (const int rc = count_invalid_elevation(c)) return rc;
  c->have_elev = true;
  c->chain_done = false;
  c->footprint_done = false;
  return TE_OK;
}

int te_upload_tile(te_ctx* c, const float* host_tile, int map, int row0, int col0, int h, int w) {
  if (!c || !host_tile) return fail(TE_ERR_INVALID_ARG, "te_upload_tile: NULL");
  CtxLock lk(c);
  if (!c->have_geo) return fail(TE_ERR_NOT_READY, "te_upload_tile: geometry not set");
  if (map < 0 || map >= c->geo.batch || row0 < 0 || col0 < 0 || h <= 0 || w <= 0 || row0 + h > c->geo.rows ||
      col0 + w > c->geo.cols)
    return fail(TE_ERR_INVALID_ARG, "te_upload_tile: tile (%d,%d)+(%d,%d) outside %dx%d", row0, col0, h, w,
                c->geo.rows, c->geo.cols);
  HIP_TRY(hipSetDevice(c->device));
  float* dst = c->L.elev + (size_t)map * c->geo.rows * c->geo.cols + (size_t)col0 * c->geo.rows + row0;
  // column-major: w columns of h contiguous rows each
  HIP_TRY(hipMemcpy2DAsync(dst, (size_t)c->geo.rows * sizeof(float), host_tile, (size_t)h * sizeof(float),
                           (size_t)h * sizeof(float), (size_t)w, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->have_elev = true;
  c->invalid_cells = -1;  // tiles are not counted: the count of the last whole upload says nothing about them (dense march)
  return TE_OK;
}

namespace {

int check_tile(te_ctx* c, const char* who, int map, int row0, int col0, int h, int w) {
  if (!c->have_geo) return fail(TE_ERR_NOT_READY, "%s: geometry not set", who);
  if (map < 0 || map >= c->geo.batch || row0 < 0 || col0 < 0 || h <= 0 || w <= 0 || row0 + h > c->geo.rows || col0 + w > c->geo.cols)
    return fail(TE_ERR_INVALID_ARG, "%s: tile (%d,%d)+(%d,%d) outside %dx%d", who, row0, col0, h, w, c->geo.rows, c->geo.cols);
  return TE_OK;
}

// a staging slot of at least n floats with its two events; growing one waits for whatever still uses it
int prepare_slot(te_ctx* c, te_ctx::TileSlot& sl, size_t n) {
  if (!sl.ready) HIP_TRY(hipEventCreateWithFlags(&sl.ready, hipEventDisableTiming));
  if (!sl.freed) HIP_TRY(hipEventCreateWithFlags(&sl.freed, hipEventDisableTiming));
  if (sl.cap < n) {
    if (sl.buf) {
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipFree(sl.buf));
      sl.buf = nullptr;
      sl.cap = 0;
      sl.used = false;
    }
    HIP_TRY(hipMalloc((void**)&sl.buf, n * sizeof(float)));
    sl.cap = n;
  }
  return TE_OK;
}

int tile_streams(te_ctx* c) {
  if (!c->in_stream) HIP_TRY(hipStreamCreateWithFlags(&c->in_stream, hipStreamNonBlocking));
  if (!c->out_stream) HIP_TRY(hipStreamCreateWithFlags(&c->out_stream, hipStreamNonBlocking));
  return TE_OK;
}

}  // namespace

int te_download_tile(te_ctx* c, int layer, int map, int row0, int col0, int h, int w, float* host_tile) {
  if (!c || !host_tile) return fail(TE_ERR_INVALID_ARG, "te_download_tile: NULL");
  CtxLock lk(c);
  if (const int rc = check_tile(c, "te_download_tile", map, row0, col0, h, w)) return rc;
  const float* p = layer_ptr(c, layer);
  if (!p) return fail(TE_ERR_INVALID_ARG, "te_download_tile: bad layer %d", layer);
  HIP_TRY(hipSetDevice(c->device));
  const float* src = p + (size_t)map * c->geo.rows * c->geo.cols + (size_t)col0 * c->geo.rows + row0;
  HIP_TRY(hipMemcpy2DAsync(host_tile, (size_t)h * sizeof(float), src, (size_t)c->geo.rows * sizeof(float), (size_t)h * sizeof(float),
                           (size_t)w, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return TE_OK;
}

int te_upload_tile_async(te_ctx* c, const float* host_tile, int map, int row0, int col0, int h, int w) {
  if (!c || !host_tile) return fail(TE_ERR_INVALID_ARG, "te_upload_tile_async: NULL");
  CtxLock lk(c);
  if (const int rc = check_tile(c, "te_upload_tile_async", map, row0, col0, h, w)) return rc;
  HIP_TRY(hipSetDevice(c->device));
  if (const int rc = tile_streams(c)) return rc;
  te_ctx::TileSlot& sl = c->in_slot[c->in_next];
  c->in_next ^= 1;
  if (const int rc = prepare_slot(c, sl, (size_t)h * w)) return rc;
  // PCIe into the slot on the copy-in stream, once the compute stream has consumed what the slot held before
  if (sl.used) HIP_TRY(hipStreamWaitEvent(c->in_stream, sl.freed, 0));
  HIP_TRY(hipMemcpyAsync(sl.buf, host_tile, (size_t)h * w * sizeof(float), hipMemcpyHostToDevice, c->in_stream));
  HIP_TRY(hipEventRecord(sl.ready, c->in_stream));
  // into the layer on the compute stream: ordered after every launch already queued there (they may still read the cells)
  HIP_TRY(hipStreamWaitEvent(c->stream, sl.ready, 0));
  float* dst = c->L.elev + (size_t)map * c->geo.rows * c->geo.cols + (size_t)col0 * c->geo.rows + row0;
  HIP_TRY(hipMemcpy2DAsync(dst, (size_t)c->geo.rows * sizeof(float), sl.buf, (size_t)h * sizeof(float), (size_t)h * sizeof(float),
                           (size_t)w, hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(hipEventRecord(sl.freed, c->stream));
  sl.used = true;
  c->tiles_pending = true;
  c->have_elev = true;
  c->invalid_cells = -1;  // (as te_upload_tile)
  return TE_OK;
}

int te_download_tile_async(te_ctx* c, int layer, int map, int row0, int col0, int h, int w, float* host_tile) {
  if (!c || !host_tile) return fail(TE_ERR_INVALID_ARG, "te_download_tile_async: NULL");
  CtxLock lk(c);
  if (const int rc = check_tile(c, "te_download_tile_async", map, row0, col0, h, w)) return rc;
  const float* p = layer_ptr(c, layer);
  if (!p) return fail(TE_ERR_INVALID_ARG, "te_download_tile_async: bad layer %d", layer);
  HIP_TRY(hipSetDevice(c->device));
  if (const int rc = tile_streams(c)) return rc;
  te_ctx::TileSlot& sl = c->out_slot[c->out_next];
  c->out_next ^= 1;
  if (const int rc = prepare_slot(c, sl, (size_t)h * w)) return rc;
  // the rectangle as the launches queued so far leave it, copied aside on the compute stream (the next tick may
  // overwrite it), once the slot's previous content has crossed PCIe
  if (sl.used) HIP_TRY(hipStreamWaitEvent(c->stream, sl.freed, 0));
  const float* src = p + (size_t)map * c->geo.rows * c->geo.cols + (size_t)col0 * c->geo.rows + row0;
  HIP_TRY(hipMemcpy2DAsync(sl.buf, (size_t)h * sizeof(float), src, (size_t)c->geo.rows * sizeof(float), (size_t)h * sizeof(float),
                           (size_t)w, hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(hipEventRecord(sl.ready, c->stream));
  HIP_TRY(hipStreamWaitEvent(c->out_stream, sl.ready, 0));
  HIP_TRY(hipMemcpyAsync(host_tile, sl.buf, (size_t)h * w * sizeof(float), hipMemcpyDeviceToHost, c->out_stream));
  HIP_TRY(hipEventRecord(sl.freed, c->out_stream));
  sl.used = true;
  c->tiles_pending = true;
  return TE_OK;
}

int te_device_ptr(te_ctx* c, int layer, void** dptr, size_t* bytes) {
  if (!c || !dptr) return fail(TE_ERR_INVALID_ARG, "te_device_ptr: NULL");
  CtxLock lk(c);
  if (!c->have_geo) return fail(TE_ERR_NOT_READY, "te_device_ptr: geometry not set");
  if (const int rc = ensure_input_layer(c, layer)) return rc;
  // (robot_slope: handing out the pointer does not make the layer present -- the buffer is all NaN until the caller
  // has filled it and says so with te_set_layer_present)
  float* p = layer_ptr(c, layer);
  if (!p) return fail(TE_ERR_INVALID_ARG, "te_device_ptr: bad layer %d", layer);
  *dptr = p;
  if (bytes) *bytes = c->layer_elems * sizeof(float);
  if (layer == TE_LAYER_TRAVERSABILITY) c->trav_external = c->trav_ptr_out = true;
  if (layer == TE_LAYER_ELEVATION) {  // caller fills the elevation in place (zero-copy producer)
    c->invalid_cells = -1;
    c->have_elev = true;
    c->chain_done = false;
    c->footprint_done = false;
  }
  return TE_OK;
}

int te_set_layer_present(te_ctx* c, int layer, int present) {
  if (!c) return fail(TE_ERR_INVALID_ARG, "te_set_layer_present: NULL ctx");
  CtxLock lk(c);
  if (layer != TE_LAYER_ROBOT_SLOPE) return fail(TE_ERR_INVALID_ARG, "te_set_layer_present: only the optional input layer robot_slope can be declared present / absent");
  if (present && !c->robot_slope) return fail(TE_ERR_NOT_READY, "te_set_layer_present: robot_slope was never uploaded nor handed out (te_device_ptr)");
  c->have_robot_slope = present != 0;
  return TE_OK;
}

int te_upload_layer(te_ctx* c, int layer, const float* host, int map0, int nmaps) {
  if (!c || !host) return fail(TE_ERR_INVALID_ARG, "te_upload_layer: NULL");
  if (layer == TE_LAYER_ELEVATION) return te_upload_elevation(c, host, map0, nmaps);
  CtxLock lk(c);
  if (!c->have_geo) return fail(TE_ERR_NOT_READY, "te_upload_layer: geometry not set");
  if (const int rc = ensure_input_layer(c, layer)) return rc;
  float* p = layer_ptr(c, layer);
  if (!p) return fail(TE_ERR_INVALID_ARG, "te_upload_layer: bad layer %d", layer);
  if (map0 < 0 || nmaps <= 0 || map0 + nmaps > c->geo.batch)
    return fail(TE_ERR_INVALID_ARG, "te_upload_layer: maps [%d,%d) of batch %d", map0, map0 + nmaps, c->geo.batch);
  HIP_TRY(hipSetDevice(c->device));
  const size_t per = (size_t)c->geo.rows * c->geo.cols;
  HIP_TRY(c->stager.upload(p + per * map0, host, per * nmaps * sizeof(float), c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (layer == TE_LAYER_ROBOT_SLOPE) c->have_robot_slope = true;
  if (layer == TE_LAYER_TRAVERSABILITY) c->trav_external = c->trav_ptr_out = true;
  return TE_OK;
}

namespace {
// GridMap layers are circular buffers: logical cell (i, j) is stored at ((i + si) % rows, (j + sj) % cols)
// (grid_map_core getBufferIndexFromIndex, (si, sj) = GridMap::getStartIndex()).  The device layers are in
// logical order, so a layer in buffer order moves as (up to) four rectangles.
// `host` may be unaligned (a payload inside a serialised message).
hipError_t copy_circular(te_ctx* c, float* dev, void* host, int si, int sj, bool to_device) {
  const int rows = c->geo.rows, cols = c->geo.cols;
  const size_t pitch = (size_t)rows * sizeof(float);
  const int i_split[3] = {0, rows - si, rows}, j_split[3] = {0, cols - sj, cols};
  for (int bj = 0; bj < 2; ++bj)
    for (int bi = 0; bi < 2; ++bi) {
      const int li = i_split[bi], lj = j_split[bj];                  // logical origin of the rectangle
      const int h = i_split[bi + 1] - li, w = j_split[bj + 1] - lj;  // rows x cols
      if (h <= 0 || w <= 0) continue;
      const int ri = (li + si) % rows, rj = (lj + sj) % cols;        // its origin in the buffer
      float* d = dev + (size_t)lj * rows + li;
      char* b = (char*)host + ((size_t)rj * rows + ri) * sizeof(float);
      const hipError_t e = to_device ? hipMemcpy2DAsync(d, pitch, b, pitch, (size_t)h * sizeof(float), (size_t)w, hipMemcpyHostToDevice, c->stream)
                                     : hipMemcpy2DAsync(b, pitch, d, pitch, (size_t)h * sizeof(float), (size_t)w, hipMemcpyDeviceToHost, c->stream);
      if (e != hipSuccess) return e;
    }
  return hipStreamSynchronize(c->stream);
}
}  // namespace

// expect_rows / expect_cols > 0: the caller laid out its host buffer for that shape (the message entry points read the
// geometry, drop the lock and come back here): fail instead of copying rows*cols cells of another shape
static int upload_layer_circular_checked(te_ctx* c, int layer, const float* host, int map, int start_row, int start_col,
                                         int expect_rows, int expect_cols);
static int download_layer_circular_checked(te_ctx* c, int layer, float* host, int map, int start_row, int start_col,
                                           int expect_rows, int expect_cols);

int te_upload_layer_circular(te_ctx* c, int layer, const float* host, int map, int start_row, int start_col) {
  return upload_layer_circular_checked(c, layer, host, map, start_row, start_col, 0, 0);
}

static int upload_layer_circular_checked(te_ctx* c, int layer, const float* host, int map, int start_row, int start_col,
                                         int expect_rows, int expect_cols) {
  if (!c || !host) return fail(TE_ERR_INVALID_ARG, "te_upload_layer_circular: NULL");
  CtxLock lk(c);
  if (!c->have_geo) return fail(TE_ERR_NOT_READY, "te_upload_layer_circular: geometry not set");
  if (expect_rows > 0 && (c->geo.rows != expect_rows || c->geo.cols != expect_cols))
    return fail(TE_ERR_NOT_READY, "the geometry changed to %dx%d under a %dx%d message transfer (another thread)", c->geo.rows,
                c->geo.cols, expect_rows, expect_cols);
  if (const int rc = ensure_input_layer(c, layer)) return rc;
  float* p = layer_ptr(c, layer);
  if (!p) return fail(TE_ERR_INVALID_ARG, "te_upload_layer_circular: bad layer %d", layer);
  if (map < 0 || map >= c->geo.batch || start_row < 0 || start_row >= c->geo.rows || start_col < 0 || start_col >= c->geo.cols)
    return fail(TE_ERR_INVALID_ARG, "te_upload_layer_circular: map %d, start index (%d,%d) of a %dx%d map", map, start_row,
                start_col, c->geo.rows, c->geo.cols);
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(copy_circular(c, p + (size_t)map * c->geo.rows * c->geo.cols, const_cast<float*>(host), start_row, start_col, true));
  if (layer == TE_LAYER_ELEVATION) {
    if (const int rc = count_invalid_elevation(c)) return rc;
    c->have_elev = true;
    c->chain_done = false;
    c->footprint_done = false;
  }
  if (layer == TE_LAYER_ROBOT_SLOPE) c->have_robot_slope = true;
  if (layer == TE_LAYER_TRAVERSABILITY) c->trav_external = c->trav_ptr_out = true;
  return TE_OK;
}

int te_download_layer_circular(te_ctx* c, int layer, float* host, int map, int start_row, int start_col) {
  return download_layer_circular_checked(c, layer, host, map, start_row, start_col, 0, 0);
}

static int download_layer_circular_checked(te_ctx* c, int layer, float* host, int map, int start_row, int start_col,
                                           int expect_rows, int expect_cols) {
  if (!c || !host) return fail(TE_ERR_INVALID_ARG, "te_download_layer_circular: NULL");
  CtxLock lk(c, /*beside_prefetch*/ true, bit(layer));
  if (!c->have_geo) return fail(TE_ERR_NOT_READY, "te_download_layer_circular: geometry not set");
  if (expect_rows > 0 && (c->geo.rows != expect_rows || c->geo.cols != expect_cols))
    return fail(TE_ERR_NOT_READY, "the geometry changed to %dx%d under a %dx%d message transfer (another thread)", c->geo.rows,
                c->geo.cols, expect_rows, expect_cols);
  float* p = layer_ptr(c, layer);
  if (!p) return fail(TE_ERR_INVALID_ARG, "te_download_layer_circular: bad layer %d", layer);
  if (map < 0 || map >= c->geo.batch || start_row < 0 || start_row >= c->geo.rows || start_col < 0 || start_col >= c->geo.cols)
    return fail(TE_ERR_INVALID_ARG, "te_download_layer_circular: map %d, start index (%d,%d) of a %dx%d map", map, start_row,
                start_col, c->geo.rows, c->geo.cols);
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(copy_circular(c, p + (size_t)map * c->geo.rows * c->geo.cols, host, start_row, start_col, false));
  return TE_OK;
}

int te_msg_parse(const void* m, size_t len, te_msg_info* info) {
  if (!m || !info) return fail(TE_ERR_INVALID_ARG, "te_msg_parse: NULL");
  msg::View v;
  std::string err;
  if (!msg::parse((const uint8_t*)m, len, v, err)) return fail(TE_ERR_INVALID_ARG, "te_msg_parse: %s", err.c_str());
  *info = v.info;
  return TE_OK;
}

int te_msg_layer(const void* m, size_t len, int k, char* name, size_t* data_offset) {
  if (!m || !name || !data_offset) return fail(TE_ERR_INVALID_ARG, "te_msg_layer: NULL");
  msg::View v;
  std::string err;
  if (!msg::parse((const uint8_t*)m, len, v, err)) return fail(TE_ERR_INVALID_ARG, "te_msg_layer: %s", err.c_str());
  if (k < 0 || k >= (int)v.layers.size()) return fail(TE_ERR_INVALID_ARG, "te_msg_layer: layer %d of %zu", k, v.layers.size());
  const msg::LayerView& l = v.layers[k];
  if (l.name_len >= TE_MSG_MAX_NAME) return fail(TE_ERR_INVALID_ARG, "te_msg_layer: layer name longer than %d", TE_MSG_MAX_NAME - 1);
  memcpy(name, l.name, l.name_len);
  name[l.name_len] = 0;
  *data_offset = l.data_off;
  return TE_OK;
}

int te_msg_write(const te_msg_info* info, int n_layers, const char* const* names, const float* const* layer_data, int n_basic,
                 const char* const* basic_names, void* out, size_t cap, size_t* written) {
  if (!info || !written || (n_layers > 0 && (!names || !layer_data))) return fail(TE_ERR_INVALID_ARG, "te_msg_write: NULL");
  const msg::Names ln = {n_layers, names}, bn = {n_basic, basic_names};
  *written = msg::message_size(*info, ln, bn);
  std::vector<size_t> off;
  std::string err;
  if (!msg::write_skeleton(*info, ln, bn, (uint8_t*)out, out ? cap : 0, off, err)) return fail(TE_ERR_INVALID_ARG, "te_msg_write: %s", err.c_str());
  for (int k = 0; k < n_layers; ++k) {
    if (!layer_data[k]) return fail(TE_ERR_INVALID_ARG, "te_msg_write: NULL layer data");
    memcpy((uint8_t*)out + off[k], layer_data[k], (size_t)info->rows * info->cols * sizeof(float));
  }
  return TE_OK;
}

int te_upload_msg(te_ctx* c, const void* m, size_t len, const char* layer_name, int layer, te_msg_info* info) {
  if (!c || !m || !layer_name) return fail(TE_ERR_INVALID_ARG, "te_upload_msg: NULL");
  msg::View v;
  std::string err;
  if (!msg::parse((const uint8_t*)m, len, v, err)) return fail(TE_ERR_INVALID_ARG, "te_upload_msg: %s", err.c_str());
  const msg::LayerView* l = nullptr;
  for (const msg::LayerView& k : v.layers)
    if (k.name_len == strlen(layer_name) && memcmp(k.name, layer_name, k.name_len) == 0) l = &k;
  // setElevationMap refuses a message without the elevation layers (TraversabilityMap.cpp:135-154)
  if (!l) return fail(TE_ERR_INVALID_ARG, "te_upload_msg: the message has no layer '%s'", layer_name);
  const te_msg_info& mi = v.info;
  bool same;
  {
    CtxLock lk(c);
    same = c->have_geo && c->geo.rows == mi.rows && c->geo.cols == mi.cols && c->geo.batch == 1 && c->geo.res == mi.resolution &&
           c->geo.pos_x == mi.pose[0] && c->geo.pos_y == mi.pose[1];
  }
  if (!same) {
    const int rc = te_set_geometry(c, mi.rows, mi.cols, 1, mi.resolution, mi.pose[0], mi.pose[1]);
    if (rc != TE_OK) return rc;
  }
  if (info) *info = mi;
  // the payload may be unaligned: it is only ever handed to the copy engine
  return upload_layer_circular_checked(c, layer, reinterpret_cast<const float*>((const uint8_t*)m + l->data_off), 0, mi.start_row,
                                       mi.start_col, mi.rows, mi.cols);
}

int te_download_msg(te_ctx* c, const te_msg_info* info, int n_layers, const int* layers, const char* const* names, int n_basic,
                    const char* const* basic_names, void* out, size_t cap, size_t* written) {
  if (!c || !info || !written || (n_layers > 0 && (!layers || !names))) return fail(TE_ERR_INVALID_ARG, "te_download_msg: NULL");
  te_msg_info mi = *info;
  {
    CtxLock lk(c);
    if (!c->have_geo) return fail(TE_ERR_NOT_READY, "te_download_msg: geometry not set");
    mi.rows = c->geo.rows;
    mi.cols = c->geo.cols;
    mi.resolution = c->geo.res;
    mi.length_x = c->geo.len_x;
    mi.length_y = c->geo.len_y;
    mi.pose[0] = c->geo.pos_x;
    mi.pose[1] = c->geo.pos_y;
  }
  const msg::Names ln = {n_layers, names}, bn = {n_basic, basic_names};
  *written = msg::message_size(mi, ln, bn);
  std::vector<size_t> off;
  std::string err;
  if (!msg::write_skeleton(mi, ln, bn, (uint8_t*)out, out ? cap : 0, off, err)) return fail(TE_ERR_INVALID_ARG, "te_download_msg: %s", err.c_str());
  for (int k = 0; k < n_layers; ++k) {
    const int rc = download_layer_circular_checked(c, layers[k], reinterpret_cast<float*>((uint8_t*)out + off[k]), 0, mi.start_row, mi.start_col,
                                                   mi.rows, mi.cols);
    if (rc != TE_OK) return rc;
  }
  return TE_OK;
}

int te_bag_find_message(const void* bag, size_t len, const char* topic, size_t* msg_offset, size_t* msg_len) {
  if (!bag || !topic || !msg_offset || !msg_len) return fail(TE_ERR_INVALID_ARG, "te_bag_find_message: NULL");
  std::string err;
  if (!msg::bag_find((const uint8_t*)bag, len, topic, *msg_offset, *msg_len, err)) return fail(TE_ERR_INVALID_ARG, "te_bag_find_message: %s", err.c_str());
  return TE_OK;
}

int te_bag_write(const void* m, size_t msg_len, const char* topic, uint32_t stamp_sec, uint32_t stamp_nsec, void* out, size_t cap,
                 size_t* written) {
  if (!m || !topic || !written) return fail(TE_ERR_INVALID_ARG, "te_bag_write: NULL");
  std::string err;
  if (!msg::bag_write((const uint8_t*)m, msg_len, topic, stamp_sec, stamp_nsec, (uint8_t*)out, out ? cap : 0, *written, err))
    return fail(TE_ERR_INVALID_ARG, "te_bag_write: %s", err.c_str());
  return TE_OK;
}

int te_download_layer(te_ctx* c, int layer, float* host, int map0, int nmaps) {
  if (!c || !host) return fail(TE_ERR_INVALID_ARG, "te_download_layer: NULL");
  CtxLock lk(c, /*beside_prefetch*/ true, bit(layer));
  if (!c->have_geo) return fail(TE_ERR_NOT_READY, "te_download_layer: geometry not set");
  float* p = layer_ptr(c, layer);
  if (!p) return fail(TE_ERR_INVALID_ARG, "te_download_layer: bad layer %d", layer);
  if (map0 < 0 || nmaps <= 0 || map0 + nmaps > c->geo.batch)
    return fail(TE_ERR_INVALID_ARG, "te_download_layer: maps [%d,%d) of batch %d", map0, map0 + nmaps, c->geo.batch);
  HIP_TRY(hipSetDevice(c->device));
  const size_t per = (size_t)c->geo.rows * c->geo.cols;
  HIP_TRY(c->stager.download(host, p + per * map0, per * nmaps * sizeof(float), c->stream));
  return TE_OK;
}

// Whole-layer uploads that run BESIDE the calls that follow (see travgpu.h).  The reference's chain hands every plugin the
// whole map (SlopeFilter.cpp:62-63, StepFilter.cpp:105-107, RoughnessFilter.cpp:76-77), so a plugin knows the layers its
// successors will read: their upload can cross PCIe host -> device while its own output crosses device -> host.
int te_prefetch_layers(te_ctx* c, int n, const int* layers, const float* const* hosts) {
  if (!c || n <= 0 || n > 8 || !layers || !hosts) return fail(TE_ERR_INVALID_ARG, "te_prefetch_layers: bad argument");
  CtxLock lk(c);  // (an earlier prefetch is finished first)
  if (!c->have_geo) return fail(TE_ERR_NOT_READY, "te_prefetch_layers: geometry not set");
  if (c->prefetch_rc.load() != TE_OK) return fail(TE_ERR_HIP, "te_prefetch_layers: an earlier prefetch failed (te_wait_prefetch reports it)");
  struct Job {
    float* dev;
    const float* host;
  };
  std::vector<Job> jobs;
  bool elev = false;
  unsigned mask = 0;
  for (int k = 0; k < n; ++k) {
    if (!hosts[k]) return fail(TE_ERR_INVALID_ARG, "te_prefetch_layers: NULL host buffer");
    if (layers[k] != TE_LAYER_ELEVATION) {
      if (const int rc = ensure_input_layer(c, layers[k])) return rc;
    }
    float* p = layer_ptr(c, layers[k]);
    if (!p) return fail(TE_ERR_INVALID_ARG, "te_prefetch_layers: bad layer %d", layers[k]);
    jobs.push_back(Job{p, hosts[k]});
    elev = elev || layers[k] == TE_LAYER_ELEVATION;
    mask |= bit(layers[k]);
  }
  HIP_TRY(hipSetDevice(c->device));
  if (!c->prefetch_order) HIP_TRY(hipStreamCreateWithFlags(&c->prefetch_order, hipStreamNonBlocking));
  // what the compute stream has queued so far may still read these layers
  HIP_TRY(hipEventRecord(c->ev0, c->stream));
  HIP_TRY(hipStreamWaitEvent(c->prefetch_order, c->ev0, 0));
  c->prefetcher.pool = 1;
  const size_t bytes = c->layer_elems * sizeof(float);
  const int device = c->device;
  auto work = [c, jobs, bytes, device] {
    int rc = TE_OK;
    if (hipSetDevice(device) != hipSuccess) rc = TE_ERR_HIP;
    for (size_t k = 0; k < jobs.size() && rc == TE_OK; ++k)
      if (c->prefetcher.upload(jobs[k].dev, jobs[k].host, bytes, c->prefetch_order) != hipSuccess) rc = TE_ERR_HIP;
    if (rc == TE_OK && hipStreamSynchronize(c->prefetch_order) != hipSuccess) rc = TE_ERR_HIP;
    if (rc != TE_OK) (void)hipGetLastError();
    c->prefetch_rc.store(rc);
  };
  c->prefetch_elev = elev;
  c->prefetch_mask = mask;
  if (!c->prefetch_thread.joinable()) {
    try {
      c->prefetch_thread = std::thread([c] {
        for (;;) {
          std::function<void()> job;
          {
            std::unique_lock<std::mutex> pl(c->pf_mu);
            c->pf_cv.wait(pl, [c] { return c->pf_quit || (bool)c->pf_job; });
            if (c->pf_quit) return;
            job.swap(c->pf_job);
          }
          job();
          {
            std::lock_guard<std::mutex> pl(c->pf_mu);
            c->prefetch_running = false;
          }
          c->pf_cv.notify_all();
        }
      });
    } catch (...) {  // no thread to be had: the uploads happen here and now
      work();
      finish_prefetch_locked(c);
      return TE_OK;
    }
  }
  {
    std::lock_guard<std::mutex> pl(c->pf_mu);
    c->pf_job = work;
    c->prefetch_running = true;
  }
  c->pf_cv.notify_all();
  return TE_OK;
}

int te_wait_prefetch(te_ctx* c) {
  if (!c) return fail(TE_ERR_INVALID_ARG, "te_wait_prefetch: NULL ctx");
  CtxLock lk(c);  // (joins the prefetch)
  const int rc = c->prefetch_rc.exchange(TE_OK);
  if (rc != TE_OK) return fail(rc, "te_prefetch_layers: a transfer failed");
  return TE_OK;
}

int te_pin_host(void* host, size_t bytes) {
  if (!host || !bytes) return fail(TE_ERR_INVALID_ARG, "te_pin_host: NULL or empty buffer");
  HIP_TRY(hipHostRegister(host, bytes, hipHostRegisterDefault));
  return TE_OK;
}

int te_unpin_host(void* host) {
  if (!host) return fail(TE_ERR_INVALID_ARG, "te_unpin_host: NULL");
  HIP_TRY(hipHostUnregister(host));
  return TE_OK;
}

}  // extern "C"
