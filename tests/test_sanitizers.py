"""AddressSanitizer + UndefinedBehaviorSanitizer on the CPU builds (the race-detection / sanitizer row;
GPU ASan is not available on this pool): the host-only parsers of grid_map_msgs/GridMap messages and rosbag V2.0 images
under 400 000 corrupted inputs (tools/fuzz_msg.cpp over te_gridmap_msg.hip, which is plain host C++), and the two CPU
harnesses over the kernels' own plan / routing headers.  Any sanitizer report aborts the program (-fno-sanitize-recover)."""
import os
import subprocess

from tests.conftest import ROOT

SAN = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"]
CSRC = os.path.join(ROOT, "traversability_estimation_amd", "csrc")


def _run(cmd, **kw):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, **kw)
    assert r.returncode == 0, " ".join(cmd) + "\n" + r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_message_and_bag_parsers_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "fuzz_msg")
    _run(["g++"] + SAN + ["-I", os.path.join(ROOT, "include"), "-I", CSRC, "-x", "c++", os.path.join(CSRC, "te_gridmap_msg.hip"),
                           os.path.join(ROOT, "tools", "fuzz_msg.cpp"), "-o", exe])
    out = _run([exe], env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    ok, rejected = (int(t.split("=")[1]) for t in out.split()[-2:])
    assert ok + rejected == 400000 and ok > 1000 and rejected > 1000, out[-300:]  # (both outcomes occur: the fuzzer reaches past the first length check)


def test_plan_and_routing_harnesses_under_asan_ubsan(tmp_path):
    for src in ("n3_plan_check.cpp", "hole_routing_check.cpp"):
        exe = str(tmp_path / src[:-4])
        _run(["g++"] + SAN + ["-Wall", "-Werror", "-I", CSRC, os.path.join(ROOT, "tests", "cpu", src), "-o", exe])
        out = _run([exe] + (["1048576", "1000", "1000"] if "routing" in src else []))
        assert ("0 failed checks" in out) if "plan" in src else out.startswith("sparse")


def test_polygon_tables_host_side_under_asan_ubsan(tmp_path):
    """te_polygon.hip's host side (build_path_polygons, build_polygon_table) on degenerate input, and the offset table against
    the per-cell crossing-number expression on 1.3e8 cell tests (tools/check_polygon_host.cpp) -- hipcc builds the host half
    of the translation unit with the sanitizers (-Xarch_host)."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    inc = ["-I", os.path.join(ROOT, "include"), "-I", CSRC]
    o1, o2, exe = str(tmp_path / "te_polygon_host.o"), str(tmp_path / "check_polygon_host.o"), str(tmp_path / "check_polygon_host")
    _run([hipcc, "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-Xarch_host", "-fsanitize=address,undefined", "-Xarch_host",
          "-fno-sanitize-recover=all"] + inc + ["-c", os.path.join(CSRC, "te_polygon.hip"), "-o", o1])
    _run([hipcc, "--cuda-host-only", "-x", "hip", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"] + inc +
         ["-c", os.path.join(ROOT, "tools", "check_polygon_host.cpp"), "-o", o2])
    _run([hipcc, "-fsanitize=address,undefined", o2, o1, "-o", exe])
    out = _run([exe])
    assert "cell tests=" in out and "path polygons=" in out, out[-500:]
