#!/bin/bash
# Round 5, experiment 14: unobserved regions -- dense march on strips of 32 rows (Layers::short_strips), steps inside a region
# that find no cell in the ring's window skipped.  B = the library before (HEAD of the morning: restart, long strips).
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_exp14
mkdir -p $OUT
P=$ROOT/traversability_estimation_amd
(cd $ROOT && timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_chain.py -m gpu -x -q -k "unobserved or clip or prefetched or chain" 2>&1 | tail -4)
(cd $ROOT && timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "holes_and_obstacles" 2>&1 | tail -3)
run() {  # tag, lib, args...
  local tag=$1 lib=$2; shift 2
  echo -n "$tag: "
  env TRAVGPU_LIB=$lib timeout 200 python $ROOT/tools/ab_chain.py --iters 60 --tag $tag "$@" 2>> $OUT/err.log | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['ms_median'], 4))"
}
for h in 0.003 0.01 0.55 0.6 0.7; do
  run B.normals.$h $P/libtravgpu_B.so --holes $h --normals-only
  run new.normals.$h $P/libtravgpu.so --holes $h --normals-only
  run B.launch.$h $P/libtravgpu_B.so --holes $h
  run new.launch.$h $P/libtravgpu.so --holes $h
done
for h in 0.55 0.7 0.01; do
  echo -n "check $h: "
  timeout 300 python $ROOT/tools/ab_chain.py --holes $h --iters 20 --tag check --check-whole 2>> $OUT/err.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); pc = d.get('parity_check', {})
print(round(d['ms_median'], 4), 'ok' if pc.get('ok') else 'MISMATCH', pc.get('mismatches'), pc.get('cells_per_layer'))"
done
tail -5 $OUT/err.log
