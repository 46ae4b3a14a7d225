"""Builds the plugin adapter library and its test driver with g++ against the stub ROS headers
(in-container check; a ROS host builds the same sources with catkin, see CMakeLists.txt)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
LIB = os.path.join(HERE, "libtraversability_estimation_filters.so")
TEST = os.path.join(HERE, "plugin_chain_test")
SRCS = ["src/DeviceMap.cpp", "src/SlopeFilter.cpp", "src/StepFilter.cpp", "src/RoughnessFilter.cpp",
        "src/FusedChainFilter.cpp", "src/SurfaceNormalsFilter.cpp", "src/TraversabilityMap.cpp", "stubs/pluginlib/registry.cpp"]


def build(verbose=False):
    inc = ["-I" + os.path.join(HERE, "include"), "-I" + os.path.join(HERE, "stubs"), "-I" + os.path.join(ROOT, "include")]
    common = ["g++", "-std=c++14", "-O2", "-fPIC", "-Wall", "-Wextra"] + inc
    cmd = common + ["-shared"] + [os.path.join(HERE, s) for s in SRCS] + [
        "-L" + PKG, "-ltravgpu", "-Wl,-rpath," + PKG, "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    cmd = common + ["-I" + os.path.join(ROOT, "oracle"), os.path.join(HERE, "test", "plugin_chain_test.cpp"),
                    "-L" + HERE, "-ltraversability_estimation_filters", "-L" + os.path.join(ROOT, "oracle"), "-lte_oracle",
                    "-Wl,-rpath," + HERE, "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-Wl,-rpath," + PKG,
                    "-Wl,--no-as-needed", "-o", TEST]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB, TEST


if __name__ == "__main__":
    print(build(verbose=True))
