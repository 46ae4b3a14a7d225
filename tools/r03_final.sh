#!/bin/bash
# Round 3, last GPU call: the record the docs quote (tests + driver bench line + kernel statistics + counters + traffic),
# the other configurations, the maps with holes and the maps with obstacles.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
ulimit -c 0
export TMPDIR=/tmp
bash $ROOT/tools/profile_round.sh r03 > $ROOT/gpurun_out/r03_profile_round.log 2>&1
tail -30 $ROOT/gpurun_out/r03_profile_round.log
(cd /tmp && timeout 600 python $ROOT/tools/bench_configs.py > $ROOT/gpurun_out/r03_configs.json 2> $ROOT/gpurun_out/r03_configs.err; echo "configs rc=$?")
bash $ROOT/tools/obstacles_bench.sh r03_obst > $ROOT/gpurun_out/r03_obst.log 2>&1; tail -8 $ROOT/gpurun_out/r03_obst.log
bash $ROOT/tools/tie_bench.sh r03_ties > $ROOT/gpurun_out/r03_ties.log 2>&1; tail -40 $ROOT/gpurun_out/r03_ties.log | cut -c1-200
bash $ROOT/tools/holes_bench.sh r03_holes > $ROOT/gpurun_out/r03_holes.log 2>&1; tail -10 $ROOT/gpurun_out/r03_holes.log
