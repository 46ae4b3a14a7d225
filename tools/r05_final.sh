#!/bin/bash
# Round 5: the record the docs quote, part 1: tests + driver bench line + kernel statistics + SQ counters + HBM traffic
# (tools/profile_round.sh) and the other BASELINE configurations.  Part 2 is tools/r05_record.sh.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
ulimit -c 0
export TMPDIR=/tmp
bash $ROOT/tools/profile_round.sh r05 > $ROOT/gpurun_out/r05_profile_round.log 2>&1
tail -40 $ROOT/gpurun_out/r05_profile_round.log | cut -c1-1800
(cd /tmp && timeout 600 python $ROOT/tools/bench_configs.py > $ROOT/gpurun_out/r05_configs.json 2> $ROOT/gpurun_out/r05_configs.err; echo "configs rc=$?")
