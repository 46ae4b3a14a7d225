#!/bin/bash
# Round 5, experiment 15: (a) the run count picks short strips for regions only (speckle back on long strips); (b) the sparse
# march's queue: h4 = the general tail in place on every row with a dirty row in its disc (no queue), it4 / it3 = 4 / 3 Newton
# steps in general_tail3 instead of 6 (timing only: what the flush's arithmetic weighs against its loads).
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_exp15
mkdir -p $OUT
P=$ROOT/traversability_estimation_amd
(cd $ROOT && timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_chain.py -m gpu -x -q -k "unobserved or prefetched or chain" 2>&1 | tail -3)
run() {  # tag, lib, args...
  local tag=$1 lib=$2; shift 2
  echo -n "$tag: "
  env TRAVGPU_LIB=$lib timeout 200 python $ROOT/tools/ab_chain.py --iters 60 --tag $tag "$@" 2>> $OUT/err.log | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['ms_median'], 4))"
}
for h in 0.003 0.01 0.55; do
  run new.normals.$h $P/libtravgpu.so --holes $h --normals-only
  run new.launch.$h $P/libtravgpu.so --holes $h
done
for h in 0.0003 0.001; do
  for v in "" _h4 _it4 _it3; do
    run sparse$v.normals.$h $P/libtravgpu$v.so --holes $h --normals-only
  done
done
run h4.launch.0.001 $P/libtravgpu_h4.so --holes 0.001
run new.launch.0.001 $P/libtravgpu.so --holes 0.001
echo -n "check h4 0.001: "
env TRAVGPU_LIB=$P/libtravgpu_h4.so timeout 300 python $ROOT/tools/ab_chain.py --holes 0.001 --iters 20 --tag check --check-whole 2>> $OUT/err.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); pc = d.get('parity_check', {})
print(round(d['ms_median'], 4), 'ok' if pc.get('ok') else 'MISMATCH', pc.get('mismatches'), pc.get('cells_per_layer'))"
tail -5 $OUT/err.log
