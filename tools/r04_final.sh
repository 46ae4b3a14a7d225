#!/bin/bash
# Round 4, the record the docs quote, part 1: tests + driver bench line + kernel statistics + SQ counters + HBM traffic
# (tools/profile_round.sh), the other BASELINE configurations and the tie radii.  Part 2 (tools/r04_record.sh) runs after
# profiles/r04_hbm_traffic.json of this call has been committed: the bench line then carries roofline.traffic.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
ulimit -c 0
export TMPDIR=/tmp
bash $ROOT/tools/profile_round.sh r04 > $ROOT/gpurun_out/r04_profile_round.log 2>&1
tail -34 $ROOT/gpurun_out/r04_profile_round.log | cut -c1-1500
(cd /tmp && timeout 600 python $ROOT/tools/bench_configs.py > $ROOT/gpurun_out/r04_configs.json 2> $ROOT/gpurun_out/r04_configs.err; echo "configs rc=$?")
bash $ROOT/tools/tie_bench.sh r04_ties > $ROOT/gpurun_out/r04_ties.log 2>&1; tail -40 $ROOT/gpurun_out/r04_ties.log | cut -c1-200
