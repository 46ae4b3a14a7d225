#!/bin/bash
# Round 5, experiment 13: (a) the clean first attempt of k_normals3 hands over at the row it reached (B = before: the strip
# started again); (b) unobserved regions: the pass lasts as long as its slowest strip -- shorter strips, more blocks than
# resident slots (the hardware hands them out as slots free up): lab library, TE_N3_STRIP_ROWS.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_exp13
mkdir -p $OUT
P=$ROOT/traversability_estimation_amd
run() {  # tag, lib, args...
  local tag=$1 lib=$2; shift 2
  echo -n "$tag: "
  env TRAVGPU_LIB=$lib $ENVX timeout 200 python $ROOT/tools/ab_chain.py --normals-only --iters 60 --tag $tag "$@" 2>> $OUT/err.log | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['ms_median'], 4))"
}
for h in 0.0001 0.0003 0.001 0.55 0.7; do
  ENVX="X=1" run B.$h $P/libtravgpu_B.so --holes $h
  ENVX="X=1" run new.$h $P/libtravgpu.so --holes $h
done
for h in 0.55 0.6 0.7 0.01; do
  for r in 0 64 47 32 24; do
    ENVX="TE_N3_STRIP_ROWS=$r" run lab.rows$r.$h $P/libtravgpu_lab.so --holes $h
  done
done
for h in 0.0003 0.55; do
  echo -n "check $h: "
  timeout 300 python $ROOT/tools/ab_chain.py --holes $h --iters 20 --tag check --check-whole 2>> $OUT/err.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); pc = d.get('parity_check', {})
print(round(d['ms_median'], 4), 'ok' if pc.get('ok') else 'MISMATCH', pc.get('mismatches'), pc.get('cells_per_layer'))"
done
tail -5 $OUT/err.log
