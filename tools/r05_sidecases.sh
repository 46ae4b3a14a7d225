#!/bin/bash
# Round 5: the record the docs quote, part 3: obstacle maps, maps with invalid cells, tie radii (each row with its parity
# check where the tool has one) -> gpurun_out/r05_obstacles, r05_holes, r05_ties.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
bash $ROOT/tools/obstacles_bench.sh r05_obstacles > $ROOT/gpurun_out/r05_obstacles.log 2>&1
tail -8 $ROOT/gpurun_out/r05_obstacles.log | cut -c1-700
bash $ROOT/tools/holes_bench.sh r05_holes > $ROOT/gpurun_out/r05_holes.log 2>&1
tail -10 $ROOT/gpurun_out/r05_holes.log | cut -c1-400
# (tie radii: unchanged code, recorded once: profiles/r05_tie_radii.json)

