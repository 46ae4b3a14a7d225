#!/bin/bash
# Round 5, experiment 16: the void step as two branches (no second copy of the staging / stores); B = the library before.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_exp16
mkdir -p $OUT
P=$ROOT/traversability_estimation_amd
run() {  # tag, lib, args...
  local tag=$1 lib=$2; shift 2
  echo -n "$tag: "
  env TRAVGPU_LIB=$lib timeout 200 python $ROOT/tools/ab_chain.py --iters 60 --tag $tag "$@" 2>> $OUT/err.log | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['ms_median'], 4))"
}
for h in 0 0.003 0.01 0.55 0.7; do
  run B.normals.$h $P/libtravgpu_B.so --holes $h --normals-only
  run new.normals.$h $P/libtravgpu.so --holes $h --normals-only
  run B.launch.$h $P/libtravgpu_B.so --holes $h
  run new.launch.$h $P/libtravgpu.so --holes $h
done
tail -5 $OUT/err.log
(cd $ROOT && timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4)
