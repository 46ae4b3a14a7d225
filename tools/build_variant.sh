#!/bin/bash
# Builds traversability_estimation_amd/libtravgpu_<name>.so with extra compiler flags for ONE source file
# (measurement aid for tools/ab_bench.sh):  tools/build_variant.sh <name> <source basename> [flags...]
set -e
NAME=$1; SRC=$2; shift 2
P=$(cd $(dirname $0)/../traversability_estimation_amd && pwd)
python -c "from traversability_estimation_amd import build as b; b.build_lib()" >/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function -I$P/../include -I$P/csrc -c $P/csrc/$SRC -o $P/_build/$SRC.$NAME.o "$@"
OBJS=$(ls $P/_build/*.hip.o | grep -v "/$SRC.o" | grep -v "/${SRC%.hip}\.p[0-9]*\.hip\.o")  # (sources built in parts: none of them)
hipcc --offload-arch=gfx950 -shared -fPIC -pthread $OBJS $P/_build/$SRC.$NAME.o -o $P/libtravgpu_$NAME.so
echo built $P/libtravgpu_$NAME.so
