// te_gridmap_msg.hip -- grid_map_msgs/GridMap (ROS1 serialisation) and rosbag V2.0, host side only.  See te_msg.h.
//
// Message layout (little endian; grid_map_msgs/GridMap.msg, GridMapInfo.msg, std_msgs/Float32MultiArray.msg):
//   Header{u32 seq; u32 sec; u32 nsec; string frame_id}  f64 resolution, length_x, length_y  Pose{7 x f64}
//   string[] layers  string[] basic_layers
//   Float32MultiArray[] data: { dim[]{string label; u32 size; u32 stride}  u32 data_offset  f32[] data }
//   u16 outer_start_index  u16 inner_start_index
// GridMapRosConverter writes every layer column-major: dim[0] = ("column_index", cols, rows*cols),
// dim[1] = ("row_index", rows, rows), value(i, j) = data[j*rows + i] -- byte for byte the Eigen matrix of the layer,
// and outer/inner_start_index = GridMap::getStartIndex()(0) / (1).
#include <math.h>
#include <string.h>

#include "te_msg.h"

namespace te {
namespace msg {
namespace {

const char kType[] = "grid_map_msgs/GridMap";
const char kMd5[] = "95681e052b1f73bf87b7eb984382b401";
// The message definition a bag's connection header carries (the .msg interface texts, comments dropped).
const char kDefinition[] =
    "GridMapInfo info\nstring[] layers\nstring[] basic_layers\nstd_msgs/Float32MultiArray[] data\n"
    "uint16 outer_start_index\nuint16 inner_start_index\n"
    "\n================================================================================\nMSG: grid_map_msgs/GridMapInfo\n"
    "Header header\nfloat64 resolution\nfloat64 length_x\nfloat64 length_y\ngeometry_msgs/Pose pose\n"
    "\n================================================================================\nMSG: std_msgs/Header\n"
    "uint32 seq\ntime stamp\nstring frame_id\n"
    "\n================================================================================\nMSG: geometry_msgs/Pose\n"
    "Point position\nQuaternion orientation\n"
    "\n================================================================================\nMSG: geometry_msgs/Point\n"
    "float64 x\nfloat64 y\nfloat64 z\n"
    "\n================================================================================\nMSG: geometry_msgs/Quaternion\n"
    "float64 x\nfloat64 y\nfloat64 z\nfloat64 w\n"
    "\n================================================================================\nMSG: std_msgs/Float32MultiArray\n"
    "MultiArrayLayout  layout\nfloat32[]         data\n"
    "\n================================================================================\nMSG: std_msgs/MultiArrayLayout\n"
    "MultiArrayDimension[] dim\nuint32 data_offset\n"
    "\n================================================================================\nMSG: std_msgs/MultiArrayDimension\n"
    "string label\nuint32 size\nuint32 stride\n";

struct In {
  const uint8_t* p;
  size_t n, at;
  bool ok;
  bool need(size_t k) {
    if (!ok || k > n - at) ok = false;
    return ok;
  }
  uint32_t u32() {
    uint32_t v = 0;
    if (need(4)) {
      memcpy(&v, p + at, 4);
      at += 4;
    }
    return v;
  }
  uint16_t u16() {
    uint16_t v = 0;
    if (need(2)) {
      memcpy(&v, p + at, 2);
      at += 2;
    }
    return v;
  }
  double f64() {
    double v = 0;
    if (need(8)) {
      memcpy(&v, p + at, 8);
      at += 8;
    }
    return v;
  }
  LayerView str() {
    LayerView s = {nullptr, 0, 0};
    const uint32_t k = u32();
    if (need(k)) {
      s.name = (const char*)p + at;
      s.name_len = k;
      at += k;
    }
    return s;
  }
};

bool is(const LayerView& s, const char* lit) { return s.name_len == strlen(lit) && memcmp(s.name, lit, s.name_len) == 0; }

struct Out {
  uint8_t* p;
  size_t cap, at;
  void raw(const void* s, size_t k) {
    if (p && at + k <= cap) memcpy(p + at, s, k);
    at += k;  // keeps counting past the end: at == the size needed
  }
  void u32(uint32_t v) { raw(&v, 4); }
  void u16(uint16_t v) { raw(&v, 2); }
  void u64(uint64_t v) { raw(&v, 8); }
  void f64(double v) { raw(&v, 8); }
  void str(const char* s, size_t k) {
    u32((uint32_t)k);
    raw(s, k);
  }
  void str(const char* s) { str(s, strlen(s)); }
  void skip(size_t k) { at += k; }
};

size_t skeleton(const te_msg_info& info, Names layers, Names basic, Out& o, std::vector<size_t>* payload_off) {
  const size_t cells = (size_t)info.rows * info.cols;
  o.u32(info.seq);
  o.u32(info.stamp_sec);
  o.u32(info.stamp_nsec);
  o.str(info.frame_id, strnlen(info.frame_id, sizeof(info.frame_id)));
  o.f64(info.resolution);
  o.f64(info.length_x);
  o.f64(info.length_y);
  for (int k = 0; k < 7; ++k) o.f64(info.pose[k]);
  o.u32((uint32_t)layers.n);
  for (int k = 0; k < layers.n; ++k) o.str(layers.v[k]);
  o.u32((uint32_t)basic.n);
  for (int k = 0; k < basic.n; ++k) o.str(basic.v[k]);
  o.u32((uint32_t)layers.n);
  for (int k = 0; k < layers.n; ++k) {
    o.u32(2);
    o.str("column_index");
    o.u32((uint32_t)info.cols);
    o.u32((uint32_t)cells);
    o.str("row_index");
    o.u32((uint32_t)info.rows);
    o.u32((uint32_t)info.rows);
    o.u32(0);  // data_offset
    o.u32((uint32_t)cells);
    if (payload_off) payload_off->push_back(o.at);
    o.skip(cells * sizeof(float));
  }
  o.u16((uint16_t)info.start_row);
  o.u16((uint16_t)info.start_col);
  return o.at;
}

bool check_info(const te_msg_info& info, Names layers, Names basic, std::string& err) {
  if (info.rows <= 0 || info.cols <= 0 || (uint64_t)info.rows * (uint64_t)info.cols > 0xFFFFFFFFull / 4) {
    err = "grid map message: bad size";
    return false;
  }
  if (info.start_row < 0 || info.start_row >= info.rows || info.start_col < 0 || info.start_col >= info.cols ||
      info.start_row > 0xFFFF || info.start_col > 0xFFFF) {
    err = "grid map message: start index outside the map (or beyond uint16)";
    return false;
  }
  if (layers.n < 0 || basic.n < 0 || (layers.n && !layers.v) || (basic.n && !basic.v)) {
    err = "grid map message: bad layer list";
    return false;
  }
  for (int k = 0; k < layers.n; ++k)
    if (!layers.v[k]) {
      err = "grid map message: NULL layer name";
      return false;
    }
  for (int k = 0; k < basic.n; ++k)
    if (!basic.v[k]) {
      err = "grid map message: NULL basic layer name";
      return false;
    }
  return true;
}

// ---- rosbag V2.0 ----
const char kMagic[] = "#ROSBAG V2.0\n";
enum { OP_MSG = 2, OP_BAG_HEADER = 3, OP_INDEX = 4, OP_CHUNK = 5, OP_CHUNK_INFO = 6, OP_CONNECTION = 7 };

struct Field {
  const uint8_t* v;
  uint32_t n;
};
struct Record {
  const uint8_t* hdr;
  uint32_t hdr_len;
  const uint8_t* data;
  uint32_t data_len;
  // value of header field `name`
  bool field(const uint8_t* h, uint32_t hn, const char* name, Field& f) const {
    const size_t nl = strlen(name);
    size_t at = 0;
    while (at + 4 <= hn) {
      uint32_t fl;
      memcpy(&fl, h + at, 4);
      at += 4;
      if (fl > hn - at) return false;
      if (fl > nl && memcmp(h + at, name, nl) == 0 && h[at + nl] == '=') {
        f.v = h + at + nl + 1;
        f.n = (uint32_t)(fl - nl - 1);
        return true;
      }
      at += fl;
    }
    return false;
  }
  bool field(const char* name, Field& f) const { return field(hdr, hdr_len, name, f); }
  int op() const {
    Field f;
    return field("op", f) && f.n == 1 ? f.v[0] : -1;
  }
};

bool next_record(const uint8_t* p, size_t n, size_t& at, Record& r) {
  uint32_t hl, dl;
  if (n - at < 4) return false;
  memcpy(&hl, p + at, 4);
  if (hl > n - at - 4 || n - at - 4 - hl < 4) return false;
  memcpy(&dl, p + at + 4 + hl, 4);
  if (dl > n - at - 8 - hl) return false;
  r.hdr = p + at + 4;
  r.hdr_len = hl;
  r.data = r.hdr + hl + 4;
  r.data_len = dl;
  at += 8 + (size_t)hl + dl;
  return true;
}

bool field_is(const Field& f, const char* s) { return f.n == strlen(s) && memcmp(f.v, s, f.n) == 0; }

struct Conn {
  uint32_t id;
  bool wanted;
};

void header_field(Out& o, const char* name, const void* v, size_t n) {
  o.u32((uint32_t)(strlen(name) + 1 + n));
  o.raw(name, strlen(name));
  o.raw("=", 1);
  o.raw(v, n);
}
template <class T>
void header_field(Out& o, const char* name, T v) {
  header_field(o, name, &v, sizeof(T));
}

// record = u32 header_len, header, u32 data_len, data; `body` writes the header fields
template <class H>
void begin_record(Out& o, H&& fields, uint32_t data_len) {
  Out probe = {nullptr, 0, 0};
  fields(probe);
  o.u32((uint32_t)probe.at);
  fields(o);
  o.u32(data_len);
}

size_t connection_record(Out& o, const char* topic) {
  const size_t at0 = o.at;
  auto data = [&](Out& d) {
    header_field(d, "md5sum", kMd5, sizeof(kMd5) - 1);
    header_field(d, "message_definition", kDefinition, sizeof(kDefinition) - 1);
    header_field(d, "type", kType, sizeof(kType) - 1);
  };
  Out probe = {nullptr, 0, 0};
  data(probe);
  begin_record(
      o,
      [&](Out& h) {
        header_field(h, "conn", (uint32_t)0);
        header_field(h, "op", (uint8_t)OP_CONNECTION);
        header_field(h, "topic", topic, strlen(topic));
      },
      (uint32_t)probe.at);
  data(o);
  return o.at - at0;
}

size_t bag_layout(Out& o, const uint8_t* message, size_t msg_len, const char* topic, uint32_t sec, uint32_t nsec) {
  if (sec == 0 && nsec == 0) nsec = 1;  // saveToBag: an unset timestamp is written as ros::TIME_MIN
  const uint64_t stamp = (uint64_t)sec | ((uint64_t)nsec << 32);
  Out probe = {nullptr, 0, 0};
  const size_t conn_len = connection_record(probe, topic);
  auto msg_fields = [&](Out& h) {
    header_field(h, "conn", (uint32_t)0);
    header_field(h, "op", (uint8_t)OP_MSG);
    header_field(h, "time", stamp);
  };
  Out mp = {nullptr, 0, 0};
  begin_record(mp, msg_fields, (uint32_t)msg_len);
  const size_t chunk_len = conn_len + mp.at + msg_len;
  auto chunk_fields = [&](Out& h) {
    header_field(h, "compression", "none", 4);
    header_field(h, "op", (uint8_t)OP_CHUNK);
    header_field(h, "size", (uint32_t)chunk_len);
  };
  auto index_fields = [&](Out& h) {
    header_field(h, "conn", (uint32_t)0);
    header_field(h, "count", (uint32_t)1);
    header_field(h, "op", (uint8_t)OP_INDEX);
    header_field(h, "ver", (uint32_t)1);
  };
  Out cp = {nullptr, 0, 0};
  begin_record(cp, chunk_fields, (uint32_t)chunk_len);
  Out ip = {nullptr, 0, 0};
  begin_record(ip, index_fields, 12);
  const uint64_t chunk_pos = sizeof(kMagic) - 1 + 4 + 4 + 4096;
  const uint64_t index_pos = chunk_pos + cp.at + chunk_len + ip.at + 12;

  o.raw(kMagic, sizeof(kMagic) - 1);
  auto bag_fields = [&](Out& h) {
    header_field(h, "chunk_count", (uint32_t)1);
    header_field(h, "conn_count", (uint32_t)1);
    header_field(h, "index_pos", index_pos);
    header_field(h, "op", (uint8_t)OP_BAG_HEADER);
  };
  Out bp = {nullptr, 0, 0};
  bag_fields(bp);
  const uint32_t pad = (uint32_t)(4096 - bp.at);  // header + padding = 4096 bytes
  begin_record(o, bag_fields, pad);
  for (uint32_t k = 0; k < pad; ++k) o.raw(" ", 1);

  begin_record(o, chunk_fields, (uint32_t)chunk_len);
  connection_record(o, topic);
  begin_record(o, msg_fields, (uint32_t)msg_len);
  o.raw(message, msg_len);

  begin_record(o, index_fields, 12);
  o.u64(stamp);
  o.u32((uint32_t)conn_len);  // offset of the message record inside the chunk

  connection_record(o, topic);
  begin_record(
      o,
      [&](Out& h) {
        header_field(h, "chunk_pos", chunk_pos);
        header_field(h, "count", (uint32_t)1);
        header_field(h, "end_time", stamp);
        header_field(h, "op", (uint8_t)OP_CHUNK_INFO);
        header_field(h, "start_time", stamp);
        header_field(h, "ver", (uint32_t)1);
      },
      8);
  o.u32(0);
  o.u32(1);
  return o.at;
}

}  // namespace

bool parse(const uint8_t* p, size_t n, View& v, std::string& err) {
  In in = {p, n, 0, true};
  v.layers.clear();
  v.basic_layers.clear();
  te_msg_info& info = v.info;
  memset(&info, 0, sizeof(info));
  info.seq = in.u32();
  info.stamp_sec = in.u32();
  info.stamp_nsec = in.u32();
  const LayerView frame = in.str();
  info.resolution = in.f64();
  info.length_x = in.f64();
  info.length_y = in.f64();
  for (int k = 0; k < 7; ++k) info.pose[k] = in.f64();
  if (!in.ok) {
    err = "grid map message: truncated header";
    return false;
  }
  if (frame.name_len >= sizeof(info.frame_id)) {
    err = "grid map message: frame_id longer than TE_MSG_MAX_NAME - 1";
    return false;
  }
  memcpy(info.frame_id, frame.name, frame.name_len);
  const uint32_t nl = in.u32();
  for (uint32_t k = 0; in.ok && k < nl; ++k) v.layers.push_back(in.str());
  const uint32_t nb = in.u32();
  for (uint32_t k = 0; in.ok && k < nb; ++k) v.basic_layers.push_back(in.str());
  const uint32_t nd = in.u32();
  if (!in.ok) {
    err = "grid map message: truncated layer lists";
    return false;
  }
  if (nd != nl) {  // GridMapRosConverter::fromMessage: "Different number of layers and data in grid map message."
    err = "grid map message: different number of layers and data";
    return false;
  }
  uint32_t rows = 0, cols = 0;
  for (uint32_t k = 0; k < nd; ++k) {
    const uint32_t ndim = in.u32();
    if (!in.ok || ndim != 2) {
      err = "grid map message: a layer is not a two-dimensional array";
      return false;
    }
    const LayerView l0 = in.str();
    const uint32_t s0 = in.u32();
    in.u32();
    const LayerView l1 = in.str();
    const uint32_t s1 = in.u32();
    in.u32();
    in.u32();  // data_offset
    const uint32_t cnt = in.u32();
    if (!in.ok) {
      err = "grid map message: truncated layer layout";
      return false;
    }
    if (!is(l0, "column_index") || !is(l1, "row_index")) {
      err = "grid map message: layer storage order is not column-major (column_index, row_index)";
      return false;
    }
    if (k == 0) {
      cols = s0;
      rows = s1;
    }
    if (s0 != cols || s1 != rows || (uint64_t)cnt != (uint64_t)rows * cols || rows == 0 || cols == 0 || rows > 0x7FFFFFFFu ||
        cols > 0x7FFFFFFFu) {
      err = "grid map message: layer sizes disagree";
      return false;
    }
    v.layers[k].data_off = in.at;
    if (!in.need((size_t)cnt * 4)) {
      err = "grid map message: truncated layer data";
      return false;
    }
    in.at += (size_t)cnt * 4;
  }
  info.start_row = in.u16();
  info.start_col = in.u16();
  if (!in.ok) {
    err = "grid map message: truncated start index";
    return false;
  }
  info.rows = (int32_t)rows;
  info.cols = (int32_t)cols;
  info.n_layers = (int32_t)nl;
  info.n_basic_layers = (int32_t)nb;
  if (nl) {
    // GridMap::setGeometry(length, resolution): size = round(length / resolution) must be what the layers hold
    if (!(info.resolution > 0.0) || !isfinite(info.resolution) || llround(info.length_x / info.resolution) != (long long)rows ||
        llround(info.length_y / info.resolution) != (long long)cols) {
      err = "grid map message: length / resolution does not match the layer size";
      return false;
    }
    if (info.start_row >= info.rows || info.start_col >= info.cols) {
      err = "grid map message: start index outside the map";
      return false;
    }
  }
  return true;
}

size_t message_size(const te_msg_info& info, Names layers, Names basic) {
  std::string err;
  if (!check_info(info, layers, basic, err)) return 0;
  Out o = {nullptr, 0, 0};
  return skeleton(info, layers, basic, o, nullptr);
}

bool write_skeleton(const te_msg_info& info, Names layers, Names basic, uint8_t* out, size_t cap, std::vector<size_t>& payload_off,
                    std::string& err) {
  if (!check_info(info, layers, basic, err)) return false;
  Out o = {out, cap, 0};
  payload_off.clear();
  if (skeleton(info, layers, basic, o, &payload_off) > cap) {
    err = "grid map message: output buffer too small";
    return false;
  }
  return true;
}

bool bag_find(const uint8_t* bag, size_t n, const char* topic, size_t& off, size_t& len, std::string& err) {
  const size_t m = sizeof(kMagic) - 1;
  if (n < m || memcmp(bag, kMagic, m) != 0) {
    err = "not a ROSBAG V2.0 file";
    return false;
  }
  std::vector<Conn> conns;
  auto note_connection = [&](const Record& r) {
    Field id, tp, ty;
    if (!r.field("conn", id) || id.n != 4 || !r.field("topic", tp)) return;
    Conn c;
    memcpy(&c.id, id.v, 4);
    c.wanted = field_is(tp, topic) && r.field(r.data, r.data_len, "type", ty) && field_is(ty, kType);
    for (const Conn& k : conns)
      if (k.id == c.id) return;
    conns.push_back(c);
  };
  bool found = false;
  size_t at = m;
  Record r;
  while (at < n) {
    if (!next_record(bag, n, at, r)) {
      err = "rosbag: truncated record";
      return false;
    }
    const int op = r.op();
    if (op == OP_CONNECTION) note_connection(r);
    if (op != OP_CHUNK) continue;
    Field comp;
    if (!r.field("compression", comp) || !field_is(comp, "none")) {
      err = "rosbag: compressed chunks (bz2 / lz4) are not supported";
      return false;
    }
    size_t cat = 0;
    Record q;
    while (cat < r.data_len) {
      if (!next_record(r.data, r.data_len, cat, q)) {
        err = "rosbag: truncated record in a chunk";
        return false;
      }
      const int qop = q.op();
      if (qop == OP_CONNECTION) note_connection(q);
      if (qop != OP_MSG) continue;
      Field id;
      uint32_t cid;
      if (!q.field("conn", id) || id.n != 4) continue;
      memcpy(&cid, id.v, 4);
      for (const Conn& k : conns)
        if (k.id == cid && k.wanted) {
          off = (size_t)(q.data - bag);
          len = q.data_len;
          found = true;
        }
    }
  }
  if (!found) err = std::string("rosbag: no grid_map_msgs/GridMap message under the topic '") + topic + "'";
  return found;
}

size_t bag_size(size_t msg_len, const char* topic) {
  Out o = {nullptr, 0, 0};
  return bag_layout(o, nullptr, msg_len, topic, 0, 0);
}

bool bag_write(const uint8_t* message, size_t msg_len, const char* topic, uint32_t sec, uint32_t nsec, uint8_t* out, size_t cap,
               size_t& written, std::string& err) {
  if (msg_len > 0xFFFF0000u) {
    err = "rosbag: message too large for one chunk";
    return false;
  }
  Out o = {out, cap, 0};
  written = bag_layout(o, message, msg_len, topic, sec, nsec);
  if (written > cap) {
    err = "rosbag: output buffer too small";
    return false;
  }
  return true;
}

}  // namespace msg
}  // namespace te
