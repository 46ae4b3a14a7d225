#!/usr/bin/env python3
"""Decode the reference's only fixture (a rosbag holding one grid_map_msgs/GridMap) into a small
.npz so that the known-answer test travels to the GPU box (``/root/reference`` does not).

Source : /root/reference/traversability_estimation/maps/elevation_map.bag
         (byte-identical copy: maps/traversability_map.bag) -- SURVEY.md F5/F6, Appendix A.
Output : tests/golden/bag_map.npz  (geometry + elevation input + the reference's own golden outputs
         of the default filter chain, traversability_estimation/config/robot_filter_parameter.yaml).

Run here (build container) only:  python tests/golden/make_golden.py
The decoder is dependency-free (no rosbag): ROSBAG V2.0 record walk + little-endian ROS
serialisation of grid_map_msgs/GridMap.
"""
import hashlib
import os
import struct
import sys

import numpy as np

BAG = "/root/reference/traversability_estimation/maps/elevation_map.bag"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bag_map.npz")


def _records(buf, pos, end):
    """Yield (header_dict, data_bytes) for rosbag-v2 records in buf[pos:end]."""
    while pos < end:
        (hlen,) = struct.unpack_from("<I", buf, pos)
        pos += 4
        hdr = {}
        hend = pos + hlen
        while pos < hend:
            (flen,) = struct.unpack_from("<I", buf, pos)
            pos += 4
            field = buf[pos:pos + flen]
            pos += flen
            name, _, value = field.partition(b"=")
            hdr[name.decode()] = value
        (dlen,) = struct.unpack_from("<I", buf, pos)
        pos += 4
        yield hdr, buf[pos:pos + dlen]
        pos += dlen


class _Reader:
    def __init__(self, b):
        self.b, self.p = b, 0

    def take(self, fmt):
        v = struct.unpack_from("<" + fmt, self.b, self.p)
        self.p += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def string(self):
        n = self.take("I")
        s = self.b[self.p:self.p + n].decode()
        self.p += n
        return s


def decode_gridmap(msg):
    r = _Reader(msg)
    info = {}
    info["seq"], info["sec"], info["nsec"] = r.take("III")
    info["frame_id"] = r.string()
    info["resolution"], info["length_x"], info["length_y"] = r.take("ddd")
    info["pose"] = r.take("7d")
    layers = [r.string() for _ in range(r.take("I"))]
    basic = [r.string() for _ in range(r.take("I"))]
    data = {}
    n = r.take("I")
    assert n == len(layers)
    for name in layers:
        ndim = r.take("I")
        dims = []
        for _ in range(ndim):
            label = r.string()
            size, stride = r.take("II")
            dims.append((label, size, stride))
        r.take("I")  # data_offset
        cnt = r.take("I")
        arr = np.frombuffer(r.b, dtype="<f4", count=cnt, offset=r.p).copy()
        r.p += 4 * cnt
        assert dims[0][0] == "column_index" and dims[1][0] == "row_index", dims
        cols, rows = dims[0][1], dims[1][1]
        # column-major: value(i, j) = data[j * rows + i]; keep the flat storage order.
        data[name] = (arr, rows, cols)
    info["outer_start"], info["inner_start"] = r.take("HH")
    info["layers"], info["basic_layers"] = layers, basic
    return info, data


def main():
    buf = open(BAG, "rb").read()
    assert buf.startswith(b"#ROSBAG V2.0\n")
    msg = None
    for hdr, dat in _records(buf, 13, len(buf)):
        op = hdr["op"][0]
        if op == 5:  # chunk
            assert hdr["compression"] == b"none"
            for h2, d2 in _records(dat, 0, len(dat)):
                if h2["op"][0] == 2:
                    msg = d2
    assert msg is not None
    info, data = decode_gridmap(msg)
    assert (info["outer_start"], info["inner_start"]) == (0, 0)
    rows, cols = data["elevation"][1:]
    out = dict(
        rows=np.int32(rows), cols=np.int32(cols), resolution=np.float64(info["resolution"]),
        length=np.array([info["length_x"], info["length_y"]], dtype=np.float64),
        position=np.array(info["pose"][:2], dtype=np.float64),
        bag_md5=np.bytes_(hashlib.md5(buf).hexdigest()),
    )
    keep = ["elevation", "traversability_slope", "traversability_step", "traversability_roughness",
            "traversability", "traversability_footprint", "slope_footprint", "step_footprint"]
    for k in keep:
        out[k] = data[k][0]  # flat, column-major storage order, float32
    np.savez_compressed(OUT, **out)
    print("layers:", info["layers"])
    print("rows x cols:", rows, cols, "res", info["resolution"], "len", info["length_x"], info["length_y"])
    for k in keep:
        a = out[k]
        print(f"  {k:28s} nan={int(np.isnan(a).sum()):6d} min={np.nanmin(a) if not np.isnan(a).all() else float('nan'):.8g} "
              f"max={np.nanmax(a) if not np.isnan(a).all() else float('nan'):.8g}")
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    sys.exit(main())
