#!/bin/bash
# Round 4, second half: the maps the headline does not show, with the final kernel sources -- invalid cells
# (tools/holes_bench.sh) and boxes (tools/obstacles_bench.sh), every timed configuration checked against the oracle.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
bash $ROOT/tools/holes_bench.sh r04b_holes > $ROOT/gpurun_out/r04b_holes.log 2>&1; tail -12 $ROOT/gpurun_out/r04b_holes.log | cut -c1-300
bash $ROOT/tools/obstacles_bench.sh r04b_obstacles > $ROOT/gpurun_out/r04b_obstacles.log 2>&1; tail -9 $ROOT/gpurun_out/r04b_obstacles.log | cut -c1-600
