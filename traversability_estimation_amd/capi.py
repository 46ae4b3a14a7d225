"""ctypes binding of the C-ABI in include/travgpu.h (libtravgpu.so).

This is the only way Python (tests, bench.py) reaches the HIP chain.  There is no fallback: if the
shared library is missing this module raises, and without a gfx950 device te_create() fails.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TRAVGPU_LIB") or os.path.join(_HERE, "libtravgpu.so")

TE_OK = 0
TE_ERR_INVALID_ARG, TE_ERR_BAD_PARAM, TE_ERR_NOT_READY, TE_ERR_HIP, TE_ERR_NO_DEVICE, TE_ERR_UNSUPPORTED = \
    -1, -2, -3, -4, -5, -6

LAYERS = dict(elevation=0, traversability_slope=1, traversability_step=2, traversability_roughness=3,
              traversability=4, traversability_footprint=5, surface_normal_x=6, surface_normal_y=7,
              surface_normal_z=8, slope_footprint=9, step_footprint=10, roughness_footprint=11,
              traversability_x=12, traversability_rot=13, robot_slope=14)
FILTERS = dict(slope=1, step=2, roughness=3, combine=4, normals=5)
RUN_KEEP_NORMALS = 0x1
RUN_FOOTPRINT = 0x2
RUN_GENERIC_KERNELS = 0x4
RUN_FOOTPRINT_MEMO = 0x8
RUN_SEQUENTIAL = 0x10
RUN_NORMALS_ONLY = 0x20
OPT_FP_BLOCKED_WALK, OPT_FP_BLOCKED_BLOCKS_PER_CU, OPT_POLYGON_PER_CELL, OPT_GRAPH_REPLAY, OPT_BCAST_RCCL, OPT_NORMALS_RANK_RULE = 1, 2, 3, 4, 5, 6

# every symbol include/travgpu.h declares (tests/test_cabi.py checks the library exports them all)
SYMBOLS = ["te_params_default", "te_params_validate", "te_device_count", "te_create", "te_destroy",
           "te_set_params", "te_get_params", "te_set_option", "te_set_geometry", "te_upload_elevation", "te_upload_tile", "te_download_tile",
           "te_upload_tile_async", "te_download_tile_async",
           "te_device_ptr", "te_set_layer_present", "te_upload_layer", "te_prefetch_layers", "te_wait_prefetch", "te_upload_layer_circular", "te_download_layer_circular", "te_run_filter", "te_run_chain", "te_run_chain_region", "te_run_footprint", "te_check_footprint_paths",
           "te_sync",
           "te_download_layer", "te_time_chain", "te_time_chain_samples", "te_last_error", "te_version",
           "te_msg_parse", "te_msg_layer", "te_msg_write", "te_upload_msg", "te_download_msg", "te_bag_find_message",
           "te_bag_write", "te_run_polygon_footprint", "te_polygons_traversable",
           "te_check_polygon_footprint_paths", "te_pin_host", "te_unpin_host", "te_path_polygons",
           "te_shard_range", "te_bcast_params", "te_run_chain_multi", "te_sync_multi",
           "te_set_check_robot_inclination", "te_check_inclination", "te_polygon_untraversable_hull"]
MSG_MAX_NAME = 64


def shard_range(batch, n_shards, k):
    first, count = C.c_int(), C.c_int()
    _check(load().te_shard_range(int(batch), int(n_shards), int(k), C.byref(first), C.byref(count)))
    return first.value, count.value


def _handles(ctxs):
    return (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])


def bcast_params(ctxs, root=0):
    """Every context receives the parameters of ctxs[root] (RCCL across devices, host copy within one)."""
    _check(load().te_bcast_params(_handles(ctxs), len(ctxs), int(root)))


def run_chain_multi(ctxs, flags=0):
    _check(load().te_run_chain_multi(_handles(ctxs), len(ctxs), int(flags)))


def sync_multi(ctxs):
    _check(load().te_sync_multi(_handles(ctxs), len(ctxs)))


def pack_paths(paths):
    """List of (n_i, 2) pose arrays -> (offsets int32[n+1], xy float64[total, 2])."""
    paths = [np.asarray(p, dtype=np.float64).reshape(-1, 2) for p in paths]
    n = len(paths)
    off = np.zeros(n + 1, np.int32)
    if n:
        off[1:] = np.cumsum([len(p) for p in paths])
    xy = np.ascontiguousarray(np.concatenate(paths) if n and off[-1] else np.zeros((1, 2)), dtype=np.float64)
    return off, xy


class TeMsgInfo(C.Structure):
    """te_msg_info: the non-layer part of a grid_map_msgs/GridMap message."""
    _fields_ = [("seq", C.c_uint32), ("stamp_sec", C.c_uint32), ("stamp_nsec", C.c_uint32),
                ("frame_id", C.c_char * MSG_MAX_NAME),
                ("resolution", C.c_double), ("length_x", C.c_double), ("length_y", C.c_double),
                ("pose", C.c_double * 7),
                ("rows", C.c_int32), ("cols", C.c_int32), ("start_row", C.c_int32), ("start_col", C.c_int32),
                ("n_layers", C.c_int32), ("n_basic_layers", C.c_int32)]


class TeParams(C.Structure):
    _fields_ = [("size", C.c_uint32), ("abi_version", C.c_uint32),
                ("normals_radius", C.c_double), ("normals_axis", C.c_int32), ("_pad0", C.c_int32),
                ("slope_critical", C.c_double),
                ("step_critical", C.c_double), ("step_radius1", C.c_double), ("step_radius2", C.c_double),
                ("step_ncrit", C.c_int32), ("_pad1", C.c_int32),
                ("rough_critical", C.c_double), ("rough_radius", C.c_double),
                ("w_scale", C.c_float), ("w_slope", C.c_float), ("w_step", C.c_float), ("w_rough", C.c_float),
                ("fp_radius", C.c_double), ("fp_offset", C.c_double), ("fp_default", C.c_double),
                ("fp_max_gap", C.c_double), ("fp_critical_step", C.c_double),
                ("fp_check_roughness", C.c_int32), ("_pad2", C.c_int32)]


class TeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"travgpu error {code}: {msg}")
        self.code = code


_lib = None


def load():
    """dlopen libtravgpu.so.  Raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -m traversability_estimation_amd.build` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        vp, pp = C.c_void_p, C.POINTER(TeParams)
        fp = C.POINTER(C.c_float)
        L.te_params_default.argtypes = [pp]
        L.te_params_validate.argtypes = [pp]
        L.te_device_count.argtypes = [C.POINTER(C.c_int)]
        L.te_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.te_destroy.argtypes = [vp]
        L.te_set_params.argtypes = [vp, pp]
        L.te_get_params.argtypes = [vp, pp]
        L.te_set_option.argtypes = [vp, C.c_int, C.c_int]
        L.te_set_geometry.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double]
        L.te_upload_elevation.argtypes = [vp, fp, C.c_int, C.c_int]
        L.te_upload_tile.argtypes = [vp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.te_download_tile.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp]
        L.te_upload_tile_async.argtypes = [vp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.te_download_tile_async.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp]
        L.te_device_ptr.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
        L.te_set_layer_present.argtypes = [vp, C.c_int, C.c_int]
        L.te_upload_layer.argtypes = [vp, C.c_int, fp, C.c_int, C.c_int]
        L.te_upload_layer_circular.argtypes = [vp, C.c_int, fp, C.c_int, C.c_int, C.c_int]
        if "TRAVGPU_LIB" not in os.environ or hasattr(L, "te_prefetch_layers"):  # (an A/B library of an earlier round lacks the pair)
            L.te_prefetch_layers.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(fp)]
            L.te_wait_prefetch.argtypes = [vp]
        L.te_download_layer_circular.argtypes = [vp, C.c_int, fp, C.c_int, C.c_int, C.c_int]
        L.te_run_filter.argtypes = [vp, C.c_int, C.c_uint]
        L.te_run_chain.argtypes = [vp, C.c_uint]
        L.te_run_chain_region.argtypes = [vp, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.te_run_footprint.argtypes = [vp]
        L.te_polygon_untraversable_hull.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_ubyte),
                                                    C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.te_set_check_robot_inclination.argtypes = [vp, C.c_int]
        L.te_check_inclination.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_ubyte), C.POINTER(C.c_int)]
        L.te_check_footprint_paths.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double),
                                               C.POINTER(C.c_ubyte), C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.te_run_polygon_footprint.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.c_double]
        L.te_polygons_traversable.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double),
                                              C.POINTER(C.c_ubyte), C.POINTER(C.c_double)]
        L.te_check_polygon_footprint_paths.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_int,
                                                       C.POINTER(C.c_double), C.POINTER(C.c_ubyte), C.POINTER(C.c_ubyte),
                                                       C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.te_pin_host.argtypes = [vp, C.c_size_t]
        L.te_unpin_host.argtypes = [vp]
        ip_, dp_ = C.POINTER(C.c_int), C.POINTER(C.c_double)
        L.te_path_polygons.argtypes = [C.c_int, ip_, dp_, C.c_int, dp_, C.POINTER(C.c_ubyte), C.c_int, C.c_int, ip_, ip_, ip_, ip_,
                                       dp_, dp_]
        L.te_sync.argtypes = [vp]
        L.te_download_layer.argtypes = [vp, C.c_int, fp, C.c_int, C.c_int]
        L.te_time_chain.argtypes = [vp, C.c_uint, C.c_int, C.c_int, C.POINTER(C.c_float)]
        L.te_time_chain_samples.argtypes = [vp, C.c_uint, C.c_int, C.c_int, C.POINTER(C.c_float)]
        L.te_shard_range.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.te_bcast_params.argtypes = [C.POINTER(vp), C.c_int, C.c_int]
        L.te_run_chain_multi.argtypes = [C.POINTER(vp), C.c_int, C.c_uint]
        L.te_sync_multi.argtypes = [C.POINTER(vp), C.c_int]
        szp, cpp = C.POINTER(C.c_size_t), C.POINTER(C.c_char_p)
        L.te_msg_parse.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(TeMsgInfo)]
        L.te_msg_layer.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_char_p, szp]
        L.te_msg_write.argtypes = [C.POINTER(TeMsgInfo), C.c_int, cpp, C.POINTER(fp), C.c_int, cpp, vp, C.c_size_t, szp]
        L.te_upload_msg.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_char_p, C.c_int, C.POINTER(TeMsgInfo)]
        L.te_download_msg.argtypes = [vp, C.POINTER(TeMsgInfo), C.c_int, C.POINTER(C.c_int), cpp, C.c_int, cpp, vp,
                                      C.c_size_t, szp]
        L.te_bag_find_message.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, szp, szp]
        L.te_bag_write.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_uint32, C.c_uint32, vp, C.c_size_t, szp]
        L.te_last_error.restype = C.c_char_p
        L.te_version.restype = C.c_char_p
        _lib = L
    return _lib


def _check(rc):
    if rc != TE_OK:
        raise TeError(rc, load().te_last_error().decode())


def default_params(**over):
    p = TeParams()
    _check(load().te_params_default(C.byref(p)))
    for k, v in over.items():
        if not hasattr(p, k):
            raise KeyError(k)
        setattr(p, k, v)
    return p


def validate_params(p):
    """The range checks of the reference's configure()s (te_params_validate); raises TeError with the reference's message."""
    _check(load().te_params_validate(C.byref(p)))
    return p


def params_to_bytes(p):
    return bytes(p)


def params_from_bytes(b):
    p = TeParams()
    assert len(b) == C.sizeof(p)
    C.memmove(C.byref(p), b, len(b))
    return p


def _names(names):
    arr = (C.c_char_p * max(len(names), 1))(*[n.encode() for n in names])
    return arr


def msg_parse(msg):
    """Validate a serialised grid_map_msgs/GridMap; returns (TeMsgInfo, {layer name: byte offset of its payload})."""
    info = TeMsgInfo()
    _check(load().te_msg_parse(msg, len(msg), C.byref(info)))
    layers = {}
    name = C.create_string_buffer(MSG_MAX_NAME)
    off = C.c_size_t()
    for k in range(info.n_layers):
        _check(load().te_msg_layer(msg, len(msg), k, name, C.byref(off)))
        layers[name.value.decode(errors="replace")] = off.value
    return info, layers


def msg_layer(msg, info, offset):
    """The payload at `offset` as a (cols, rows) float32 array in the message's own (circular) storage order."""
    return np.frombuffer(msg, dtype="<f4", count=info.rows * info.cols, offset=offset).reshape(info.cols, info.rows)


def msg_write(info, layers, basic_layers=()):
    """GridMapRosConverter::toMessage for host layers: {name: rows*cols float32 in storage order} -> bytes."""
    names = list(layers)
    arrs = [np.ascontiguousarray(layers[n], dtype=np.float32).reshape(-1) for n in names]
    for a in arrs:
        assert a.size == info.rows * info.cols, (a.size, info.rows, info.cols)
    fpp = C.POINTER(C.c_float)
    data = (fpp * max(len(arrs), 1))(*[a.ctypes.data_as(fpp) for a in arrs])
    need = C.c_size_t()
    L = load()
    L.te_msg_write(C.byref(info), len(names), _names(names), data, len(basic_layers), _names(list(basic_layers)), None, 0,
                   C.byref(need))
    out = C.create_string_buffer(max(need.value, 1))
    _check(L.te_msg_write(C.byref(info), len(names), _names(names), data, len(basic_layers), _names(list(basic_layers)), out,
                          need.value, C.byref(need)))
    return out.raw[:need.value]


def bag_find_message(bag, topic):
    """GridMapRosConverter::loadFromBag's pick: the last grid_map_msgs/GridMap message under `topic`."""
    off, n = C.c_size_t(), C.c_size_t()
    _check(load().te_bag_find_message(bag, len(bag), topic.encode(), C.byref(off), C.byref(n)))
    return bag[off.value:off.value + n.value]


def bag_write(msg, topic, stamp=(0, 0)):
    """GridMapRosConverter::saveToBag: a one-message rosbag V2.0 image."""
    need = C.c_size_t()
    L = load()
    L.te_bag_write(msg, len(msg), topic.encode(), int(stamp[0]), int(stamp[1]), None, 0, C.byref(need))
    out = C.create_string_buffer(max(need.value, 1))
    _check(L.te_bag_write(msg, len(msg), topic.encode(), int(stamp[0]), int(stamp[1]), out, need.value, C.byref(need)))
    return out.raw[:need.value]


def pin_host(array):
    """Page-lock a numpy buffer that is reused across frames (te_pin_host); unpin_host before dropping it."""
    _check(load().te_pin_host(array.ctypes.data_as(C.c_void_p), array.nbytes))


def unpin_host(array):
    _check(load().te_unpin_host(array.ctypes.data_as(C.c_void_p)))


def path_polygons(paths, points_xyz, conservative=None):
    """The polygons checkPolygonalFootprintPath evaluates (host computation, no device): returns a list per path of
    (vertices float64[n, 2], area) tuples."""
    paths = [np.asarray(p, dtype=np.float64).reshape(-1, 7) for p in paths]
    n = len(paths)
    off = np.zeros(n + 1, np.int32)
    if n:
        off[1:] = np.cumsum([len(p) for p in paths])
    poses = np.ascontiguousarray(np.concatenate(paths) if n and off[-1] else np.zeros((1, 7)), dtype=np.float64)
    pts = np.ascontiguousarray(points_xyz, dtype=np.float64).reshape(-1, 3)
    cons = None if conservative is None else np.ascontiguousarray(conservative, dtype=np.uint8)
    ip_, dp_ = C.POINTER(C.c_int), C.POINTER(C.c_double)
    args = (n, off.ctypes.data_as(ip_), poses.ctypes.data_as(dp_), len(pts), pts.ctypes.data_as(dp_),
            None if cons is None else cons.ctypes.data_as(C.POINTER(C.c_ubyte)))
    npoly, nvert = C.c_int(), C.c_int()
    L = load()
    L.te_path_polygons(*args, 0, 0, C.byref(npoly), C.byref(nvert), None, None, None, None)  # sizing call
    first = np.zeros(n + 1, np.int32)
    voff = np.zeros(npoly.value + 1, np.int32)
    xy = np.zeros((max(nvert.value, 1), 2), np.float64)
    area = np.zeros(max(npoly.value, 1), np.float64)
    _check(L.te_path_polygons(*args, npoly.value, nvert.value, C.byref(npoly), C.byref(nvert), first.ctypes.data_as(ip_),
                              voff.ctypes.data_as(ip_), xy.ctypes.data_as(dp_), area.ctypes.data_as(dp_)))
    return [[(xy[voff[q]:voff[q + 1]].copy(), float(area[q])) for q in range(first[k], first[k + 1])] for k in range(n)]


def device_count():
    n = C.c_int(0)
    rc = load().te_device_count(C.byref(n))
    return n.value if rc == TE_OK else 0


class Context:
    """One device context: owns the device-resident layers of a batch of maps (thin OO view of te_ctx)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _check(load().te_create(int(device), C.byref(self._h)))
        self.rows = self.cols = self.batch = 0

    def close(self):
        if self._h:
            load().te_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_params(self, p):
        _check(load().te_set_params(self._h, C.byref(p)))

    def set_option(self, option, value):
        _check(load().te_set_option(self._h, int(option), int(value)))

    def get_params(self):
        p = TeParams()
        _check(load().te_get_params(self._h, C.byref(p)))
        return p

    def set_geometry(self, rows, cols, batch, res, pos=(0.0, 0.0)):
        _check(load().te_set_geometry(self._h, int(rows), int(cols), int(batch), float(res), float(pos[0]), float(pos[1])))
        self.rows, self.cols, self.batch = int(rows), int(cols), int(batch)

    def upload_elevation(self, elev, map0=0):
        a = np.ascontiguousarray(elev, dtype=np.float32).reshape(-1)
        per = self.rows * self.cols
        assert a.size % per == 0, (a.size, per)
        _check(load().te_upload_elevation(self._h, a.ctypes.data_as(C.POINTER(C.c_float)), int(map0), a.size // per))

    def upload_tile(self, tile, map_index, row0, col0):
        """tile: array of shape (w, h) == [col][row] (column-major h x w tile)."""
        t = np.ascontiguousarray(tile, dtype=np.float32)
        w, h = t.shape
        _check(load().te_upload_tile(self._h, t.ctypes.data_as(C.POINTER(C.c_float)), int(map_index), int(row0),
                                     int(col0), int(h), int(w)))

    def download_tile(self, layer, map_index, row0, col0, h, w):
        """The h x w rectangle of a layer as an array of shape (w, h) == [col][row]."""
        t = np.empty((int(w), int(h)), np.float32)
        _check(load().te_download_tile(self._h, LAYERS[layer] if isinstance(layer, str) else int(layer), int(map_index), int(row0),
                                       int(col0), int(h), int(w), t.ctypes.data_as(C.POINTER(C.c_float))))
        return t

    def upload_tile_async(self, tile, map_index, row0, col0):
        """As upload_tile, but returns at once (copy-in stream + staging slot); `tile` must be a C-contiguous float32 array of
        shape (w, h) that stays alive and unchanged until sync() -- page-lock it (pin_host) for a truly asynchronous copy."""
        assert tile.dtype == np.float32 and tile.flags["C_CONTIGUOUS"], "upload_tile_async: float32, C-contiguous (no hidden copy)"
        w, h = tile.shape
        _check(load().te_upload_tile_async(self._h, tile.ctypes.data_as(C.POINTER(C.c_float)), int(map_index), int(row0), int(col0),
                                           int(h), int(w)))

    def download_tile_async(self, layer, map_index, row0, col0, out):
        """The rectangle of shape out.shape == (w, h) into `out` (float32, C-contiguous); valid after sync()."""
        assert out.dtype == np.float32 and out.flags["C_CONTIGUOUS"]
        w, h = out.shape
        _check(load().te_download_tile_async(self._h, LAYERS[layer] if isinstance(layer, str) else int(layer), int(map_index),
                                             int(row0), int(col0), int(h), int(w), out.ctypes.data_as(C.POINTER(C.c_float))))

    def device_ptr(self, layer):
        p, n = C.c_void_p(), C.c_size_t()
        _check(load().te_device_ptr(self._h, LAYERS[layer] if isinstance(layer, str) else int(layer), C.byref(p),
                                    C.byref(n)))
        return p.value, n.value

    def set_layer_present(self, layer, present):
        _check(load().te_set_layer_present(self._h, LAYERS[layer] if isinstance(layer, str) else int(layer), 1 if present else 0))

    def upload_layer(self, layer, data, map0=0):
        a = np.ascontiguousarray(data, dtype=np.float32).reshape(-1)
        per = self.rows * self.cols
        assert a.size % per == 0, (a.size, per)
        _check(load().te_upload_layer(self._h, LAYERS[layer] if isinstance(layer, str) else int(layer),
                                      a.ctypes.data_as(C.POINTER(C.c_float)), int(map0), a.size // per))

    def prefetch_layers(self, layers):
        """{layer: array}: whole-layer uploads that run beside the calls that follow, until wait_prefetch() (te_prefetch_layers).
        The arrays are kept alive by this object until then."""
        names = list(layers)
        arrs = [np.ascontiguousarray(layers[k], dtype=np.float32).reshape(-1) for k in names]
        for a in arrs:
            assert a.size == self.rows * self.cols * self.batch, (a.size, self.rows, self.cols, self.batch)
        ids = (C.c_int * len(names))(*[LAYERS[k] if isinstance(k, str) else int(k) for k in names])
        ptrs = (C.POINTER(C.c_float) * len(names))(*[a.ctypes.data_as(C.POINTER(C.c_float)) for a in arrs])
        self._prefetch_keep = arrs
        _check(load().te_prefetch_layers(self._h, len(names), ids, ptrs))

    def wait_prefetch(self):
        try:
            _check(load().te_wait_prefetch(self._h))
        finally:
            self._prefetch_keep = None

    def upload_layer_circular(self, layer, data, start_index, map_index=0):
        """Upload ONE map's layer given in GridMap buffer order (start_index = GridMap::getStartIndex())."""
        a = np.ascontiguousarray(data, dtype=np.float32).reshape(-1)
        assert a.size == self.rows * self.cols, (a.size, self.rows, self.cols)
        _check(load().te_upload_layer_circular(self._h, LAYERS[layer] if isinstance(layer, str) else int(layer),
                                               a.ctypes.data_as(C.POINTER(C.c_float)), int(map_index),
                                               int(start_index[0]), int(start_index[1])))

    def download_layer_circular(self, layer, start_index, map_index=0):
        """Download ONE map's layer into GridMap buffer order; returns a (cols, rows) array (row index fastest)."""
        out = np.empty((self.cols, self.rows), dtype=np.float32)
        _check(load().te_download_layer_circular(self._h, LAYERS[layer] if isinstance(layer, str) else int(layer),
                                                 out.ctypes.data_as(C.POINTER(C.c_float)), int(map_index),
                                                 int(start_index[0]), int(start_index[1])))
        return out

    def run_filter(self, which, flags=0):
        _check(load().te_run_filter(self._h, FILTERS[which] if isinstance(which, str) else int(which), int(flags)))

    def run_chain(self, flags=0):
        _check(load().te_run_chain(self._h, int(flags)))

    def run_chain_region(self, map_index, row0, col0, h, w, flags=0):
        _check(load().te_run_chain_region(self._h, int(flags), int(map_index), int(row0), int(col0), int(h), int(w)))

    def run_footprint(self):
        _check(load().te_run_footprint(self._h))

    def check_footprint_paths(self, paths, map_index=0):
        """paths: sequence of (n_i, 2) arrays of (x, y) poses.  Returns (is_safe[bool], traversability[float64], status[int32])
        like TraversabilityMap::checkFootprintPath for circular footprints, on the resident footprint layer."""
        off, xy = pack_paths(paths)
        return self.check_footprint_paths_packed(off, xy, map_index)

    def check_footprint_paths_packed(self, off, xy, map_index=0):
        """The same with the poses already packed: off int32[n+1] (off[0] == 0), xy float64[off[-1], 2]."""
        off = np.ascontiguousarray(off, dtype=np.int32)
        xy = np.ascontiguousarray(xy, dtype=np.float64)
        n = len(off) - 1
        safe = np.zeros(max(n, 1), np.uint8)
        trav = np.zeros(max(n, 1), np.float64)
        st = np.zeros(max(n, 1), np.int32)
        _check(load().te_check_footprint_paths(self._h, int(map_index), n, off.ctypes.data_as(C.POINTER(C.c_int)),
                                               xy.ctypes.data_as(C.POINTER(C.c_double)),
                                               safe.ctypes.data_as(C.POINTER(C.c_ubyte)),
                                               trav.ctypes.data_as(C.POINTER(C.c_double)),
                                               st.ctypes.data_as(C.POINTER(C.c_int))))
        return safe[:n].astype(bool), trav[:n], st[:n]

    def polygon_untraversable_hull(self, polygon, map_index=0, cap=4096):
        """isTraversable(polygon, computeUntraversablePolygon=True): (is_traversable, traversability, hull (k, 2))."""
        v = np.ascontiguousarray(polygon, dtype=np.float64).reshape(-1, 2)
        ok, val, nh = C.c_ubyte(), C.c_double(), C.c_int()
        hull = np.zeros((max(cap, 1), 2), np.float64)
        dp = C.POINTER(C.c_double)
        _check(load().te_polygon_untraversable_hull(self._h, int(map_index), len(v), v.ctypes.data_as(dp), C.byref(ok),
                                                    C.byref(val), int(cap), C.byref(nh), hull.ctypes.data_as(dp)))
        return bool(ok.value), val.value, hull[:nh.value].copy()

    def set_check_robot_inclination(self, enabled):
        """footprint/check_robot_inclination: the path checks then run checkInclination on the layer robot_slope."""
        _check(load().te_set_check_robot_inclination(self._h, int(bool(enabled))))

    def check_inclination(self, segments, map_index=0):
        """Batched TraversabilityMap::checkInclination: segments (n, 4) = start x y, end x y -> (ok[bool], status)."""
        seg = np.ascontiguousarray(segments, dtype=np.float64).reshape(-1, 4)
        n = len(seg)
        ok = np.zeros(max(n, 1), np.uint8)
        st = np.zeros(max(n, 1), np.int32)
        _check(load().te_check_inclination(self._h, int(map_index), n, seg.ctypes.data_as(C.POINTER(C.c_double)),
                                           ok.ctypes.data_as(C.POINTER(C.c_ubyte)), st.ctypes.data_as(C.POINTER(C.c_int))))
        return ok[:n].astype(bool), st[:n]

    def upload_msg(self, msg, layer_name="elevation", layer="elevation"):
        """fromMessage + upload: geometry from the message, layer `layer_name` into device layer `layer`."""
        info = TeMsgInfo()
        _check(load().te_upload_msg(self._h, msg, len(msg), layer_name.encode(),
                                    LAYERS[layer] if isinstance(layer, str) else int(layer), C.byref(info)))
        self.rows, self.cols, self.batch = info.rows, info.cols, 1
        return info

    def download_msg(self, info, layers, basic_layers=()):
        """toMessage: {message layer name: device layer} of map 0 -> serialised grid_map_msgs/GridMap bytes."""
        names = list(layers)
        ids = (C.c_int * max(len(names), 1))(*[LAYERS[v] if isinstance(v, str) else int(v) for v in layers.values()])
        need = C.c_size_t()
        L = load()
        args = (self._h, C.byref(info), len(names), ids, _names(names), len(basic_layers), _names(list(basic_layers)))
        L.te_download_msg(*args, None, 0, C.byref(need))
        out = C.create_string_buffer(max(need.value, 1))
        _check(L.te_download_msg(*args, out, need.value, C.byref(need)))
        return out.raw[:need.value]

    def run_polygon_footprint(self, points_xy, yaw):
        """traversabilityFootprint(footprintYaw): fills the layers traversability_x / traversability_rot."""
        pts = np.ascontiguousarray(points_xy, dtype=np.float64).reshape(-1, 2)
        _check(load().te_run_polygon_footprint(self._h, len(pts), pts.ctypes.data_as(C.POINTER(C.c_double)), float(yaw)))

    def polygons_traversable(self, polygons, map_index=0):
        """Batched isTraversable(polygon): list of (n_i, 2) vertex arrays -> (traversable bool[n], traversability float64[n])."""
        off, xy = pack_paths(polygons)
        n = len(off) - 1
        ok = np.zeros(max(n, 1), np.uint8)
        trav = np.zeros(max(n, 1), np.float64)
        _check(load().te_polygons_traversable(self._h, int(map_index), n, off.ctypes.data_as(C.POINTER(C.c_int)),
                                              xy.ctypes.data_as(C.POINTER(C.c_double)),
                                              ok.ctypes.data_as(C.POINTER(C.c_ubyte)),
                                              trav.ctypes.data_as(C.POINTER(C.c_double))))
        return ok[:n].astype(bool), trav[:n]

    def check_polygon_footprint_paths(self, paths, points_xyz, conservative=None, map_index=0):
        """Batched checkFootprintPath for a polygonal footprint: paths = list of (n_i, 7) pose arrays (position xyz,
        orientation xyzw); returns (is_safe bool[n], traversability[n], area[n], status[n])."""
        paths = [np.asarray(p, dtype=np.float64).reshape(-1, 7) for p in paths]
        n = len(paths)
        off = np.zeros(n + 1, np.int32)
        if n:
            off[1:] = np.cumsum([len(p) for p in paths])
        poses = np.concatenate(paths) if n and off[-1] else np.zeros((1, 7))
        return self.check_polygon_footprint_paths_packed(off, poses, points_xyz, conservative, map_index)

    def check_polygon_footprint_paths_packed(self, off, poses, points_xyz, conservative=None, map_index=0):
        """The same with the poses already packed: off int32[n+1] (off[0] == 0), poses float64[off[-1], 7]."""
        off = np.ascontiguousarray(off, dtype=np.int32)
        poses = np.ascontiguousarray(poses, dtype=np.float64)
        n = len(off) - 1
        pts = np.ascontiguousarray(points_xyz, dtype=np.float64).reshape(-1, 3)
        cons = None if conservative is None else np.ascontiguousarray(conservative, dtype=np.uint8)
        safe = np.zeros(max(n, 1), np.uint8)
        trav = np.zeros(max(n, 1), np.float64)
        area = np.zeros(max(n, 1), np.float64)
        st = np.zeros(max(n, 1), np.int32)
        dp = C.POINTER(C.c_double)
        _check(load().te_check_polygon_footprint_paths(
            self._h, int(map_index), n, off.ctypes.data_as(C.POINTER(C.c_int)), poses.ctypes.data_as(dp), len(pts),
            pts.ctypes.data_as(dp), None if cons is None else cons.ctypes.data_as(C.POINTER(C.c_ubyte)),
            safe.ctypes.data_as(C.POINTER(C.c_ubyte)), trav.ctypes.data_as(dp), area.ctypes.data_as(dp),
            st.ctypes.data_as(C.POINTER(C.c_int))))
        return safe[:n].astype(bool), trav[:n], area[:n], st[:n]

    def sync(self):
        _check(load().te_sync(self._h))

    def download(self, layer, map0=0, nmaps=None):
        nmaps = self.batch - map0 if nmaps is None else nmaps
        out = np.empty(nmaps * self.rows * self.cols, np.float32)
        _check(load().te_download_layer(self._h, LAYERS[layer] if isinstance(layer, str) else int(layer),
                                        out.ctypes.data_as(C.POINTER(C.c_float)), int(map0), int(nmaps)))
        return out

    def download_into(self, layer, out, map0=0, nmaps=None):
        """te_download_layer into a caller-owned float32 buffer (reused, possibly pinned)."""
        nmaps = self.batch - map0 if nmaps is None else nmaps
        assert out.dtype == np.float32 and out.flags.c_contiguous and out.size == nmaps * self.rows * self.cols
        _check(load().te_download_layer(self._h, LAYERS[layer] if isinstance(layer, str) else int(layer),
                                        out.ctypes.data_as(C.POINTER(C.c_float)), int(map0), int(nmaps)))
        return out

    def time_chain(self, flags=0, warmup=3, iters=10):
        ms = C.c_float()
        _check(load().te_time_chain(self._h, int(flags), int(warmup), int(iters), C.byref(ms)))
        return ms.value

    def time_chain_samples(self, flags=0, warmup=3, iters=100):
        """Device time of each of `iters` launches (HIP events on the context's stream), in ms."""
        ms = (C.c_float * int(iters))()
        _check(load().te_time_chain_samples(self._h, int(flags), int(warmup), int(iters), ms))
        return np.array(ms[:], dtype=np.float64)
