#!/bin/bash
# Round 6, the record the docs quote, part 1: GPU tests + driver bench line + kernel statistics (two-stream and sequential) +
# SQ counters + HBM traffic of the bench launch (tools/profile_round.sh).  The counter summaries go into profiles/ (with
# tools/valu_classes.py run on the build box, where the objects are); part 2 is tools/r06_record.sh.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
ulimit -c 0
export TMPDIR=/tmp
bash $ROOT/tools/profile_round.sh r06 > $ROOT/gpurun_out/r06_profile_round.log 2>&1
tail -40 $ROOT/gpurun_out/r06_profile_round.log | cut -c1-1800
