#!/bin/bash
# Round 5, experiment 12: the sparse-hole march after (a) the clean first attempt hands over at the row it reached instead
# of starting the strip again, (b) the invalid cells are taken one at a time for all lanes (scalar loops over the row masks).
# old = the library before, h0 = after (shape 81 only); h1..h3 as in exp11, h4 = the general tail in place of the queue.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_exp12
mkdir -p $OUT
P=$ROOT/traversability_estimation_amd
run() {  # tag, lib, args...
  local tag=$1 lib=$2; shift 2
  echo -n "$tag: "
  env TRAVGPU_LIB=$lib timeout 200 python $ROOT/tools/ab_chain.py --normals-only --iters 60 --tag $tag "$@" 2>> $OUT/err.log | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['ms_median'], 4))"
}
for h in 0.0001 0.0003 0.001 0.003 0.01 0.55 0.6 0.7; do
  run old.$h $P/libtravgpu.so --holes $h
  run new.$h $P/libtravgpu_h0.so --holes $h
done
for v in 1 2 3 4; do run h$v.0.001 $P/libtravgpu_h$v.so --holes 0.001; done
for h in 0.0003 0.001 0.55; do
  echo -n "check $h: "
  env TRAVGPU_LIB=$P/libtravgpu_h0.so timeout 300 python $ROOT/tools/ab_chain.py --holes $h --iters 20 --tag check --check-whole 2>> $OUT/err.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); pc = d.get('parity_check', {})
print(round(d['ms_median'], 4), 'ok' if pc.get('ok') else 'MISMATCH', pc.get('mismatches'), pc.get('cells_per_layer'))"
done
tail -5 $OUT/err.log
