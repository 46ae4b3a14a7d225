#!/bin/bash
# Round 5, experiment 19: tail_ties with the ring cells at immediate offsets, the (+-R, 0) decisions and their x/y moments
# taken once per lane, the unclipped shape's moments read once per strip (B = before: four take_out calls through run-time
# ring slots, six table loads per row).
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_exp19
mkdir -p $OUT
P=$ROOT/traversability_estimation_amd
(cd $ROOT && timeout 600 python -m pytest tests -m gpu -x -q -k "tie or ties" 2>&1 | tail -3)
run() {  # tag, lib, args...
  local tag=$1 lib=$2; shift 2
  echo -n "$tag: "
  env TRAVGPU_LIB=$lib timeout 200 python $ROOT/tools/ab_chain.py --iters 40 --tag $tag "$@" 2>> $OUT/err.log | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['ms_median'], 4))"
}
for c in 9 5 7; do
  run B.normals.tie$c $P/libtravgpu_B.so --radius-cells $c --exact-chain --normals-only
  run new.normals.tie$c $P/libtravgpu.so --radius-cells $c --exact-chain --normals-only
  run B.launch.tie$c $P/libtravgpu_B.so --radius-cells $c --exact-chain
  run new.launch.tie$c $P/libtravgpu.so --radius-cells $c --exact-chain
done
tail -3 $OUT/err.log
