"""grid_map_msgs/GridMap messages and rosbag V2.0 images either side of the chain (C-ABI te_msg_* / te_bag_* /
te_upload_msg / te_download_msg).  The host-side parsing and writing is checked on the CPU against an independent
pure-Python decoder (tests/golden/make_golden.py) and, when the reference checkout is present, against its own bag."""
import hashlib
import os
import struct
import sys

import numpy as np
import pytest

from tests.conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as G  # noqa: E402  (the dependency-free decoder that produced tests/golden/bag_map.npz)

REF_BAG = G.BAG


@pytest.fixture(scope="module")
def capi():
    from traversability_estimation_amd import build, capi
    build.build_lib()
    capi.load()
    return capi


def make_info(capi, rows, cols, res=0.05, start=(0, 0), pos=(1.5, -2.25), frame="odom", stamp=(1529564943, 122772932)):
    info = capi.TeMsgInfo()
    info.seq, info.stamp_sec, info.stamp_nsec = 7, stamp[0], stamp[1]
    info.frame_id = frame.encode()
    info.resolution, info.length_x, info.length_y = res, rows * res, cols * res
    info.pose[:] = [pos[0], pos[1], 0.25, 0.0, 0.0, 0.0, 1.0]
    info.rows, info.cols, info.start_row, info.start_col = rows, cols, start[0], start[1]
    return info


def random_layers(rows, cols, names, seed=0):
    rng = np.random.default_rng(seed)
    out = {}
    for n in names:
        a = rng.standard_normal(rows * cols).astype(np.float32)
        a[rng.random(rows * cols) < 0.05] = np.nan
        out[n] = a
    return out


def test_message_round_trip(capi):
    rows, cols = 37, 53
    info = make_info(capi, rows, cols, start=(11, 52))
    layers = random_layers(rows, cols, ["elevation", "variance", "a_layer_with_a_rather_long_name"])
    msg = capi.msg_write(info, layers, basic_layers=("elevation",))
    got, offs = capi.msg_parse(msg)
    assert list(offs) == list(layers)
    for f in ("seq", "stamp_sec", "stamp_nsec", "frame_id", "resolution", "length_x", "length_y", "rows", "cols",
              "start_row", "start_col"):
        assert getattr(got, f) == getattr(info, f), f
    assert list(got.pose) == list(info.pose)
    assert (got.n_layers, got.n_basic_layers) == (3, 1)
    for n, a in layers.items():
        b = capi.msg_layer(msg, got, offs[n]).reshape(-1)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), n
    # the independent decoder reads the same thing
    pinfo, pdata = G.decode_gridmap(msg)
    assert (pinfo["frame_id"], pinfo["resolution"], pinfo["outer_start"], pinfo["inner_start"]) == ("odom", 0.05, 11, 52)
    assert pinfo["layers"] == list(layers) and pinfo["basic_layers"] == ["elevation"]
    for n, a in layers.items():
        arr, r, c = pdata[n]
        assert (r, c) == (rows, cols) and np.array_equal(arr.view(np.uint32), a.view(np.uint32))


def test_message_without_layers_and_bad_arguments(capi):
    info = make_info(capi, 4, 5)
    msg = capi.msg_write(info, {})
    got, offs = capi.msg_parse(msg)
    assert offs == {} and got.n_layers == 0
    info.start_row = 4  # outside the map
    with pytest.raises(capi.TeError, match="start index"):
        capi.msg_write(info, {})


def test_malformed_messages_are_rejected(capi):
    rows, cols = 6, 9
    info = make_info(capi, rows, cols, start=(2, 3))
    msg = capi.msg_write(info, random_layers(rows, cols, ["elevation", "variance"]))
    capi.msg_parse(msg)
    for cut in list(range(0, 200, 7)) + [len(msg) - 1, len(msg) - 2, len(msg) - 5]:
        with pytest.raises(capi.TeError):
            capi.msg_parse(msg[:cut])
    with pytest.raises(capi.TeError, match="column-major"):
        capi.msg_parse(msg.replace(b"column_index", b"colomn_index", 1))
    # row-major labels (swapped) are a different storage order
    swapped = msg.replace(b"\x0c\x00\x00\x00column_index", b"\x09\x00\x00\x00row_index", 1)
    with pytest.raises(capi.TeError):
        capi.msg_parse(swapped)
    # layers / data count mismatch: drop the second name from the layer list (fromMessage's own check)
    pinfo, _ = G.decode_gridmap(msg)
    at = msg.index(b"\x02\x00\x00\x00\x09\x00\x00\x00elevation")
    broken = msg[:at] + b"\x01\x00\x00\x00\x09\x00\x00\x00elevation" + msg[at + 4 + 13 + 4 + 8:]
    with pytest.raises(capi.TeError, match="different number of layers"):
        capi.msg_parse(broken)
    # length / resolution inconsistent with the layer size
    bad = bytearray(msg)
    at = 12 + 4 + len("odom")
    struct.pack_into("<d", bad, at, 0.07)
    with pytest.raises(capi.TeError, match="resolution"):
        capi.msg_parse(bytes(bad))
    # start index outside the map
    bad = bytearray(msg)
    struct.pack_into("<H", bad, len(bad) - 4, rows)
    with pytest.raises(capi.TeError, match="start index"):
        capi.msg_parse(bytes(bad))


def test_bag_round_trip(capi):
    rows, cols = 10, 12
    info = make_info(capi, rows, cols)
    msg = capi.msg_write(info, random_layers(rows, cols, ["elevation"]))
    bag = capi.bag_write(msg, "grid_map", stamp=(info.stamp_sec, info.stamp_nsec))
    assert capi.bag_find_message(bag, "grid_map") == msg
    with pytest.raises(capi.TeError, match="no grid_map_msgs/GridMap message under the topic"):
        capi.bag_find_message(bag, "other_topic")
    with pytest.raises(capi.TeError, match="ROSBAG"):
        capi.bag_find_message(b"#ROSBAG V1.2\n" + bag[13:], "grid_map")
    with pytest.raises(capi.TeError):
        capi.bag_find_message(bag[:len(bag) // 2], "grid_map")
    # structure, read back with the independent record walker: bag header (4096 bytes of header + padding, index_pos),
    # one uncompressed chunk = connection + message, its index, then the connection and the chunk info
    assert bag.startswith(b"#ROSBAG V2.0\n")
    recs = list(G._records(bag, 13, len(bag)))
    assert [h["op"][0] for h, _ in recs] == [3, 5, 4, 7, 6]
    (bh, pad), (ch, chunk), (ih, idx), (kh, conn), (fh, cinfo) = recs
    assert set(pad) == {0x20} and sum(4 + len(k) + 1 + len(v) for k, v in bh.items()) + len(pad) == 4096
    index_pos = struct.unpack("<Q", bh["index_pos"])[0]
    chunk_pos = struct.unpack("<Q", fh["chunk_pos"])[0]
    assert chunk_pos == 13 + 4 + 4 + 4096
    assert bag[index_pos + 4:index_pos + 4 + 4 + 5] == struct.pack("<I", 9) + b"conn="  # a connection record starts there
    assert ch["compression"] == b"none" and struct.unpack("<I", ch["size"])[0] == len(chunk)
    inner = list(G._records(chunk, 0, len(chunk)))
    assert [h["op"][0] for h, _ in inner] == [7, 2]
    assert inner[0][0]["topic"] == b"grid_map" and inner[1][1] == msg
    stamp = struct.pack("<II", info.stamp_sec, info.stamp_nsec)
    assert inner[1][0]["time"] == stamp and fh["start_time"] == stamp and fh["end_time"] == stamp
    t, off = idx[:8], struct.unpack("<I", idx[8:])[0]
    assert t == stamp and chunk[off + 4:].startswith(struct.pack("<I", 9) + b"conn=")  # offset of the message record
    assert cinfo == struct.pack("<II", 0, 1) and conn == inner[0][1]
    assert b"md5sum=95681e052b1f73bf87b7eb984382b401" in conn and b"type=grid_map_msgs/GridMap" in conn
    # an unset timestamp is stored as ros::TIME_MIN (saveToBag)
    bag0 = capi.bag_write(msg, "grid_map")
    chunk0 = list(G._records(bag0, 13, len(bag0)))[1][1]
    inner0 = list(G._records(chunk0, 0, len(chunk0)))
    assert inner0[1][0]["time"] == struct.pack("<II", 0, 1)


def test_bag_with_several_messages_returns_the_last(capi):
    """loadFromBag calls fromMessage for every message of the view: the last one stays."""
    rows, cols = 5, 4
    info = make_info(capi, rows, cols)
    m1 = capi.msg_write(info, random_layers(rows, cols, ["elevation"], seed=1))
    m2 = capi.msg_write(info, random_layers(rows, cols, ["elevation"], seed=2))
    b1, b2 = capi.bag_write(m1, "grid_map", (5, 0)), capi.bag_write(m2, "grid_map", (6, 0))
    # splice the chunk + index records of the second bag behind those of the first (the reader walks records in
    # file order; the trailing index section is not needed for that)
    r2 = list(G._records(b2, 13, len(b2)))
    index_pos1 = struct.unpack("<Q", list(G._records(b1, 13, len(b1)))[0][0]["index_pos"])[0]
    index_pos2 = struct.unpack("<Q", r2[0][0]["index_pos"])[0]
    both = b1[:index_pos1] + b2[13 + 8 + 4096:index_pos2] + b1[index_pos1:]
    assert capi.bag_find_message(both, "grid_map") == m2


@pytest.mark.skipif(not os.path.exists(REF_BAG), reason="the reference checkout is not on this machine")
def test_reference_bag(capi, bag):
    """The reference's own fixture (maps/elevation_map.bag) through te_bag_find_message + te_msg_parse."""
    raw = open(REF_BAG, "rb").read()
    assert hashlib.md5(raw).hexdigest() == bytes(bag["bag_md5"]).decode()
    msg = capi.bag_find_message(raw, "grid_map")
    info, offs = capi.msg_parse(msg)
    assert (info.rows, info.cols, info.start_row, info.start_col) == (100, 133, 0, 0)
    assert info.frame_id == b"map" and info.resolution == 0.03 and (info.stamp_sec, info.stamp_nsec) == (1529564943, 122772932)
    assert abs(info.length_x - 3.0) < 1e-12 and abs(info.length_y - 3.99) < 1e-12
    assert list(info.pose) == [0, 0, 0, 0, 0, 0, 1]
    for k in ("elevation", "traversability_slope", "traversability_step", "traversability_roughness", "traversability"):
        got = capi.msg_layer(msg, info, offs[k]).reshape(-1)
        assert np.array_equal(got.view(np.uint32), bag[k].view(np.uint32)), k
    # re-serialising the parsed message reproduces it byte for byte, and so does the bag around it except for the
    # message definition text (ours is the interface without the comments)
    pinfo, pdata = G.decode_gridmap(msg)
    again = capi.msg_write(info, {n: pdata[n][0] for n in pinfo["layers"]}, basic_layers=pinfo["basic_layers"])
    assert again == msg
    ours = capi.bag_write(msg, "grid_map", (info.stamp_sec, info.stamp_nsec))
    theirs = list(G._records(raw, 13, len(raw)))
    mine = list(G._records(ours, 13, len(ours)))
    assert [h["op"] for h, _ in theirs] == [h["op"] for h, _ in mine]
    for (ht, dt), (hm, dm) in zip(theirs, mine):
        assert list(ht) == list(hm)  # same header fields in the same order
        for k in ht:
            if k not in ("index_pos", "size"):
                assert ht[k] == hm[k], k


# ---------------------------------------------------------------- on the device
@pytest.mark.gpu
@pytest.mark.parametrize("start", [(0, 0), (31, 77)])
def test_message_in_message_out(capi, bag, start):
    """fromMessage -> chain -> toMessage without a host GridMap: the golden bag map as a (circular) message in, the
    reference's golden outputs (rolled the same way) out."""
    from tests.helpers import OUT_LAYERS, compare_layer
    rows, cols = int(bag["rows"]), int(bag["cols"])
    si, sj = start
    info = make_info(capi, rows, cols, res=float(bag["resolution"]), start=start, pos=tuple(bag["position"]), frame="map")
    info.length_x, info.length_y = bag["length"]

    def roll(a):
        return np.roll(np.asarray(a).reshape(cols, rows), (sj, si), axis=(0, 1)).reshape(-1)

    msg = capi.msg_write(info, {"variance": np.zeros(rows * cols, np.float32), "elevation": roll(bag["elevation"])})
    with capi.Context(0) as ctx:
        ctx.set_params(capi.default_params())
        got_info = ctx.upload_msg(msg)
        assert (got_info.rows, got_info.cols, got_info.start_row, got_info.start_col) == (rows, cols, si, sj)
        assert np.array_equal(ctx.download("elevation").view(np.uint32), bag["elevation"].view(np.uint32))
        ctx.run_chain(0)
        ctx.sync()
        out = ctx.download_msg(got_info, {"elevation": "elevation", **{k: k for k in OUT_LAYERS}}, basic_layers=("elevation",))
        with pytest.raises(capi.TeError, match="no layer"):
            ctx.upload_msg(msg, layer_name="elevation_inpainted")
    oinfo, offs = capi.msg_parse(out)
    assert (oinfo.start_row, oinfo.start_col, oinfo.frame_id, oinfo.n_basic_layers) == (si, sj, b"map", 1)
    assert (oinfo.stamp_sec, oinfo.stamp_nsec, oinfo.resolution) == (info.stamp_sec, info.stamp_nsec, info.resolution)
    assert list(oinfo.pose) == list(info.pose)
    known = np.zeros((cols, rows), bool)
    for (i, j) in ((99, 117), (99, 118)):
        known[j, i] = True
    known = roll(known)
    assert np.array_equal(capi.msg_layer(out, oinfo, offs["elevation"]).reshape(-1).view(np.uint32),
                          roll(bag["elevation"]).view(np.uint32))
    for k in OUT_LAYERS:
        got = capi.msg_layer(out, oinfo, offs[k]).reshape(-1)
        n_bad, mx, _ = compare_layer(k, got[~known], roll(bag[k])[~known])
        assert n_bad == 0, (k, n_bad, mx)
    # and the written message survives a trip through a bag
    assert capi.bag_find_message(capi.bag_write(out, "traversability_map"), "traversability_map") == out


def test_parsers_survive_corrupted_input(capi):
    """Messages and bags come off the wire: any corruption must end in TE_OK or an error code, never in a crash or an
    out-of-bounds payload offset."""
    rows, cols = 7, 5
    info = make_info(capi, rows, cols, start=(3, 1))
    msg = capi.msg_write(info, random_layers(rows, cols, ["elevation", "variance"]), basic_layers=("elevation",))
    bag = capi.bag_write(msg, "grid_map", (12, 34))
    rng = np.random.default_rng(2024)
    parsed = rejected = 0
    for blob, is_bag in ((msg, False), (bag, True)):
        for trial in range(3000):
            b = bytearray(blob)
            kind = trial % 4
            if kind == 0:  # flip a few bytes
                for _ in range(int(rng.integers(1, 4))):
                    b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            elif kind == 1:  # overwrite a 32-bit length-like field with an extreme value
                at = int(rng.integers(0, max(1, len(b) - 4)))
                b[at:at + 4] = struct.pack("<I", int(rng.choice([0, 1, 0x7FFFFFFF, 0xFFFFFFFF, 0xFFFFFFF0, len(b), len(b) - at])))
            elif kind == 2:  # truncate
                b = b[:int(rng.integers(0, len(b)))]
            else:  # splice garbage in
                at = int(rng.integers(0, len(b)))
                b[at:at] = rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8).tobytes()
            b = bytes(b)
            try:
                if is_bag:
                    m = capi.bag_find_message(b, "grid_map")
                    assert len(m) <= len(b)
                else:
                    got, offs = capi.msg_parse(b)
                    for off in offs.values():
                        assert 0 <= off and off + 4 * got.rows * got.cols <= len(b)
                parsed += 1
            except capi.TeError:
                rejected += 1
    assert parsed > 100 and rejected > 1000
