"""ctypes front-end of the CPU ORACLE (oracle/libte_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (traversability_estimation_amd) never does.  See oracle/te_oracle.h for what is restated.

Layers are float32 numpy arrays in grid_map storage order: flat, column-major,
value(i, j) = a[j * rows + i].  2-D views used here have shape (cols, rows) (C-contiguous), i.e.
``a2d[j, i]``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Geom(C.Structure):
    _fields_ = [("rows", C.c_int), ("cols", C.c_int), ("res", C.c_double), ("len_x", C.c_double),
                ("len_y", C.c_double), ("pos_x", C.c_double), ("pos_y", C.c_double)]


class Params(C.Structure):
    _fields_ = [("normals_radius", C.c_double), ("normals_axis", C.c_int), ("slope_critical", C.c_double),
                ("step_critical", C.c_double), ("step_radius1", C.c_double), ("step_radius2", C.c_double),
                ("step_ncrit", C.c_int), ("rough_critical", C.c_double), ("rough_radius", C.c_double),
                ("w_scale", C.c_float), ("w_slope", C.c_float), ("w_step", C.c_float), ("w_rough", C.c_float),
                ("fp_radius", C.c_double), ("fp_offset", C.c_double), ("fp_default", C.c_double),
                ("fp_max_gap", C.c_double), ("fp_critical_step", C.c_double), ("fp_check_roughness", C.c_int)]


def build(force=False):
    so = os.path.join(_HERE, "libte_oracle.so")
    src = os.path.join(_HERE, "te_oracle.c")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libte_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libte_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        fp = C.POINTER(C.c_float)
        gp, pp = C.POINTER(Geom), C.POINTER(Params)
        L.teo_geom_init.argtypes = [gp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double]
        L.teo_params_default.argtypes = [pp]
        L.teo_set_threads.argtypes = [C.c_int]
        L.teo_set_normals_rank_rule.argtypes = [C.c_int]
        L.teo_set_normals_rank_rule.restype = None
        L.teo_get_max_threads.restype = C.c_int
        L.teo_normals.argtypes = [gp, fp, C.c_double, C.c_int, fp, fp, fp]
        L.teo_slope.argtypes = [gp, fp, C.c_double, fp]
        L.teo_step.argtypes = [gp, fp, C.c_double, C.c_double, C.c_double, C.c_int, fp, fp]
        L.teo_roughness.argtypes = [gp, fp, fp, fp, fp, C.c_double, C.c_double, fp]
        L.teo_combine.argtypes = [C.c_long, fp, fp, fp, C.c_float, C.c_float, C.c_float, C.c_float, fp]
        L.teo_chain.argtypes = [gp, pp, fp, fp, fp, fp, fp, fp, fp, fp]
        L.teo_footprint.argtypes = [gp, pp, fp, fp, fp, fp, fp, fp, fp, fp, fp]
        L.teo_circle_count.argtypes = [gp, C.c_int, C.c_int, C.c_double]
        L.teo_check_circular_paths.argtypes = [gp, fp, C.c_double, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double),
                                               C.POINTER(C.c_ubyte), C.POINTER(C.c_double), C.POINTER(C.c_int)]
        ip = C.POINTER(C.c_int)
        dp_ = C.POINTER(C.c_double)
        L.teo_polygon_untraversable_hull.argtypes = [gp, pp, fp, fp, fp, fp, fp, C.c_int, dp_, C.POINTER(C.c_ubyte), dp_, C.c_int,
                                                     ip, dp_]
        L.teo_check_circular_paths_incl.argtypes = [gp, fp, C.c_double, fp, C.c_int, ip, dp_, C.POINTER(C.c_ubyte), dp_, ip]
        L.teo_check_inclination.argtypes = [gp, fp, C.c_int, dp_, C.POINTER(C.c_ubyte), ip]
        L.teo_check_polygon_paths_incl.argtypes = [gp, pp, fp, fp, fp, fp, fp, fp, C.c_int, ip, dp_, C.c_int, dp_,
                                                   C.POINTER(C.c_ubyte), C.POINTER(C.c_ubyte), dp_, dp_, ip]
        L.teo_polygons_traversable.argtypes = [gp, pp, fp, fp, fp, fp, fp, C.c_int, C.POINTER(C.c_int), dp_,
                                               C.POINTER(C.c_ubyte), dp_]
        L.teo_rotate_footprint.argtypes = [C.c_int, dp_, C.c_double, dp_]
        L.teo_rotate_footprint.restype = None
        L.teo_polygon_footprint.argtypes = [gp, pp, fp, fp, fp, fp, fp, C.c_int, dp_, C.c_double, fp, fp]
        L.teo_check_polygon_paths.argtypes = [gp, pp, fp, fp, fp, fp, fp, C.c_int, C.POINTER(C.c_int), dp_, C.c_int, dp_,
                                              C.POINTER(C.c_ubyte), C.POINTER(C.c_ubyte), dp_, dp_, C.POINTER(C.c_int)]
        L.teo_spiral_offsets.argtypes = [gp, C.c_int, C.c_int, C.c_double, ip, ip, ip, C.c_int]
        _LIB = L
    return _LIB


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def geom(rows, cols, res, pos=(0.0, 0.0)):
    g = Geom()
    lib().teo_geom_init(C.byref(g), int(rows), int(cols), float(res), float(pos[0]), float(pos[1]))
    return g


def default_params(**over):
    p = Params()
    lib().teo_params_default(C.byref(p))
    for k, v in over.items():
        if not hasattr(p, k):
            raise KeyError(k)
        setattr(p, k, v)
    return p


def set_threads(n):
    lib().teo_set_threads(int(n))


def set_normals_rank_rule(on):
    """NormalVectorsFilter's degenerate-plane rule of the filter that wrote the reference's bag (te_oracle.c); off by default."""
    lib().teo_set_normals_rank_rule(1 if on else 0)


def max_threads():
    return int(lib().teo_get_max_threads())


def _flat(a, n):
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
    assert a.size == n, (a.size, n)
    return a


def chain(g, p, elev, want_normals=False):
    """Run normals->slope->step->roughness->combine.  Returns dict of flat float32 layers."""
    n = g.rows * g.cols
    e = _flat(elev, n)
    out = {k: np.empty(n, np.float32) for k in ("traversability_slope", "traversability_step",
                                                "traversability_roughness", "traversability")}
    nrm = [np.empty(n, np.float32) for _ in range(3)] if want_normals else [None] * 3
    rc = lib().teo_chain(C.byref(g), C.byref(p), _f(e), _f(out["traversability_slope"]),
                         _f(out["traversability_step"]), _f(out["traversability_roughness"]),
                         _f(out["traversability"]), _f(nrm[0]), _f(nrm[1]), _f(nrm[2]))
    if rc:
        raise RuntimeError(f"teo_chain failed: {rc}")
    if want_normals:
        out["surface_normal_x"], out["surface_normal_y"], out["surface_normal_z"] = nrm
    return out


def step(g, elev, crit, r1, r2, ncrit, want_step_height=False):
    n = g.rows * g.cols
    e = _flat(elev, n)
    out = np.empty(n, np.float32)
    sh = np.empty(n, np.float32) if want_step_height else None
    rc = lib().teo_step(C.byref(g), _f(e), crit, r1, r2, int(ncrit), _f(out), _f(sh))
    if rc:
        raise RuntimeError(f"teo_step failed: {rc}")
    return (out, sh) if want_step_height else out


def footprint(g, p, elev, layers, want_memo=False):
    n = g.rows * g.cols
    e = _flat(elev, n)
    sl = _flat(layers["traversability_slope"], n)
    st = _flat(layers["traversability_step"], n)
    ro = _flat(layers["traversability_roughness"], n)
    tr = _flat(layers["traversability"], n)
    fp = np.empty(n, np.float32)
    memo = [np.empty(n, np.float32) for _ in range(3)] if want_memo else [None] * 3
    rc = lib().teo_footprint(C.byref(g), C.byref(p), _f(e), _f(sl), _f(st), _f(ro), _f(tr), _f(fp), _f(memo[0]),
                             _f(memo[1]), _f(memo[2]))
    if rc:
        raise RuntimeError(f"teo_footprint failed: {rc}")
    if want_memo:
        return fp, dict(slope_footprint=memo[0], step_footprint=memo[1], roughness_footprint=memo[2])
    return fp


def check_inclination(g, robot_slope, segments):
    """Batched TraversabilityMap::checkInclination(start, end): segments (n, 4) = sx sy ex ey -> (ok[bool], status)."""
    seg = np.ascontiguousarray(segments, dtype=np.float64).reshape(-1, 4)
    k = len(seg)
    ok = np.zeros(max(k, 1), np.uint8)
    st = np.zeros(max(k, 1), np.int32)
    rs = _flat(robot_slope, g.rows * g.cols)
    rc = lib().teo_check_inclination(C.byref(g), _f(rs), k, seg.ctypes.data_as(C.POINTER(C.c_double)),
                                     ok.ctypes.data_as(C.POINTER(C.c_ubyte)), st.ctypes.data_as(C.POINTER(C.c_int)))
    if rc:
        raise RuntimeError(f"teo_check_inclination failed: {rc}")
    return ok[:k].astype(bool), st[:k]


def check_circular_paths(g, footprint, fp_default, paths, robot_slope=None):
    """TraversabilityMap::checkFootprintPath (circular footprints) for a list of (n_i, 2) pose arrays; robot_slope:
    the layer checkInclination reads when footprint/check_robot_inclination is set (None: off)."""
    n = g.rows * g.cols
    fp = _flat(footprint, n)
    paths = [np.asarray(p, dtype=np.float64).reshape(-1, 2) for p in paths]
    k = len(paths)
    off = np.zeros(k + 1, np.int32)
    if k:
        off[1:] = np.cumsum([len(p) for p in paths])
    xy = np.ascontiguousarray(np.concatenate(paths) if k and off[-1] else np.zeros((1, 2)), dtype=np.float64)
    safe = np.zeros(max(k, 1), np.uint8)
    trav = np.zeros(max(k, 1), np.float64)
    st = np.zeros(max(k, 1), np.int32)
    rs = None if robot_slope is None else _flat(robot_slope, n)
    rc = lib().teo_check_circular_paths_incl(C.byref(g), _f(fp), C.c_double(fp_default), _f(rs), k,
                                        off.ctypes.data_as(C.POINTER(C.c_int)), xy.ctypes.data_as(C.POINTER(C.c_double)),
                                        safe.ctypes.data_as(C.POINTER(C.c_ubyte)), trav.ctypes.data_as(C.POINTER(C.c_double)),
                                        st.ctypes.data_as(C.POINTER(C.c_int)))
    if rc:
        raise RuntimeError(f"teo_check_circular_paths failed: {rc}")
    return safe[:k].astype(bool), trav[:k], st[:k]


def polygons_traversable(g, p, elev, slope, step, rough, trav, polygons):
    """Batched TraversabilityMap::isTraversable(polygon) for a list of (n_i, 2) vertex arrays."""
    n = g.rows * g.cols
    polys = [np.asarray(q, dtype=np.float64).reshape(-1, 2) for q in polygons]
    k = len(polys)
    off = np.zeros(k + 1, np.int32)
    off[1:] = np.cumsum([len(q) for q in polys])
    xy = np.ascontiguousarray(np.concatenate(polys), dtype=np.float64)
    ok = np.zeros(k, np.uint8)
    out = np.zeros(k, np.float64)
    rc = lib().teo_polygons_traversable(C.byref(g), C.byref(p), _f(_flat(elev, n)), _f(_flat(slope, n)), _f(_flat(step, n)),
                                        _f(_flat(rough, n)), _f(_flat(trav, n)), k, off.ctypes.data_as(C.POINTER(C.c_int)),
                                        xy.ctypes.data_as(C.POINTER(C.c_double)), ok.ctypes.data_as(C.POINTER(C.c_ubyte)),
                                        out.ctypes.data_as(C.POINTER(C.c_double)))
    if rc:
        raise RuntimeError(f"teo_polygons_traversable failed: {rc}")
    return ok.astype(bool), out


def polygon_untraversable_hull(g, p, elev, slope, step, rough, trav, polygon, cap=4096):
    """isTraversable(polygon, computeUntraversablePolygon=True): (is_traversable, traversability, hull (k, 2))."""
    n = g.rows * g.cols
    v = np.ascontiguousarray(polygon, dtype=np.float64).reshape(-1, 2)
    ok = C.c_ubyte()
    val = C.c_double()
    nh = C.c_int()
    hull = np.zeros((cap, 2), np.float64)
    dp_ = C.POINTER(C.c_double)
    rc = lib().teo_polygon_untraversable_hull(C.byref(g), C.byref(p), _f(_flat(elev, n)), _f(_flat(slope, n)), _f(_flat(step, n)),
                                              _f(_flat(rough, n)), _f(_flat(trav, n)), len(v), v.ctypes.data_as(dp_),
                                              C.byref(ok), C.byref(val), cap, C.byref(nh), hull.ctypes.data_as(dp_))
    if rc:
        raise RuntimeError(f"teo_polygon_untraversable_hull failed: {rc}")
    return bool(ok.value), val.value, hull[:nh.value].copy()


def rotate_footprint(points_xy, yaw):
    pts = np.ascontiguousarray(points_xy, dtype=np.float64).reshape(-1, 2)
    out = np.empty_like(pts)
    lib().teo_rotate_footprint(len(pts), pts.ctypes.data_as(C.POINTER(C.c_double)), C.c_double(yaw),
                               out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def polygon_footprint(g, p, elev, slope, step, rough, trav, points_xy, yaw):
    """traversabilityFootprint(footprintYaw): returns (traversability_x, traversability_rot)."""
    n = g.rows * g.cols
    pts = np.ascontiguousarray(points_xy, dtype=np.float64).reshape(-1, 2)
    tx = np.empty(n, np.float32)
    tr = np.empty(n, np.float32)
    rc = lib().teo_polygon_footprint(C.byref(g), C.byref(p), _f(_flat(elev, n)), _f(_flat(slope, n)), _f(_flat(step, n)),
                                     _f(_flat(rough, n)), _f(_flat(trav, n)), len(pts),
                                     pts.ctypes.data_as(C.POINTER(C.c_double)), C.c_double(yaw), _f(tx), _f(tr))
    if rc:
        raise RuntimeError(f"teo_polygon_footprint failed: {rc}")
    return tx, tr


def check_polygon_paths(g, p, elev, slope, step, rough, trav, paths, points_xyz, conservative=None, robot_slope=None):
    """checkPolygonalFootprintPath for a list of (n_i, 7) pose arrays (position xyz, orientation xyzw)."""
    n = g.rows * g.cols
    paths = [np.asarray(q, dtype=np.float64).reshape(-1, 7) for q in paths]
    k = len(paths)
    off = np.zeros(k + 1, np.int32)
    if k:
        off[1:] = np.cumsum([len(q) for q in paths])
    poses = np.ascontiguousarray(np.concatenate(paths) if k and off[-1] else np.zeros((1, 7)), dtype=np.float64)
    pts = np.ascontiguousarray(points_xyz, dtype=np.float64).reshape(-1, 3)
    cons = None if conservative is None else np.ascontiguousarray(conservative, dtype=np.uint8)
    safe = np.zeros(max(k, 1), np.uint8)
    out = np.zeros(max(k, 1), np.float64)
    area = np.zeros(max(k, 1), np.float64)
    st = np.zeros(max(k, 1), np.int32)
    dp_ = C.POINTER(C.c_double)
    rs = None if robot_slope is None else _flat(robot_slope, n)
    rc = lib().teo_check_polygon_paths_incl(C.byref(g), C.byref(p), _f(_flat(elev, n)), _f(_flat(slope, n)), _f(_flat(step, n)),
                                       _f(_flat(rough, n)), _f(_flat(trav, n)), _f(rs), k, off.ctypes.data_as(C.POINTER(C.c_int)),
                                       poses.ctypes.data_as(dp_), len(pts), pts.ctypes.data_as(dp_),
                                       None if cons is None else cons.ctypes.data_as(C.POINTER(C.c_ubyte)),
                                       safe.ctypes.data_as(C.POINTER(C.c_ubyte)), out.ctypes.data_as(dp_),
                                       area.ctypes.data_as(dp_), st.ctypes.data_as(C.POINTER(C.c_int)))
    if rc:
        raise RuntimeError(f"teo_check_polygon_paths failed: {rc}")
    return safe[:k].astype(bool), out[:k], area[:k], st[:k]


def circle_count(g, i, j, radius):
    return int(lib().teo_circle_count(C.byref(g), int(i), int(j), float(radius)))


def spiral_offsets(g, ci, cj, radius, cap=4096):
    di = (C.c_int * cap)()
    dj = (C.c_int * cap)()
    rg = (C.c_int * cap)()
    n = lib().teo_spiral_offsets(C.byref(g), int(ci), int(cj), float(radius), di, dj, rg, cap)
    if n < 0 or n > cap:
        raise RuntimeError(f"teo_spiral_offsets: {n}")
    return np.array(di[:n]), np.array(dj[:n]), np.array(rg[:n])
