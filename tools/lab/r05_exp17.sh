#!/bin/bash
# Round 5, experiment 17: the sparse march looks at its queue before the step's row prefetch is issued (C = the library
# before: behind it), and the random sweep on the new hole marches (regions up to 240 cells on the large maps).
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_exp17
mkdir -p $OUT
P=$ROOT/traversability_estimation_amd
run() {  # tag, lib, args...
  local tag=$1 lib=$2; shift 2
  echo -n "$tag: "
  env TRAVGPU_LIB=$lib timeout 200 python $ROOT/tools/ab_chain.py --iters 60 --tag $tag "$@" 2>> $OUT/err.log | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['ms_median'], 4))"
}
for h in 0.0003 0.001 0.002; do
  run C.normals.$h $P/libtravgpu_C.so --holes $h --normals-only
  run new.normals.$h $P/libtravgpu.so --holes $h --normals-only
  run C.launch.$h $P/libtravgpu_C.so --holes $h
  run new.launch.$h $P/libtravgpu.so --holes $h
done
echo -n "check 0.001: "
timeout 300 python $ROOT/tools/ab_chain.py --holes 0.001 --iters 20 --tag check --check-whole 2>> $OUT/err.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); pc = d.get('parity_check', {})
print(round(d['ms_median'], 4), 'ok' if pc.get('ok') else 'MISMATCH', pc.get('mismatches'), pc.get('cells_per_layer'))"
tail -3 $OUT/err.log
cd $ROOT
TE_RANDOM_CASES="30000:5200" TE_RANDOM_REGION_CASES="9000:200" timeout 1200 python -m pytest tests/test_gpu_random.py -q -m gpu -n 16 > $OUT/pytest.log 2>&1
echo "rc=$?" >> $OUT/pytest.log
grep -E "^FAILED|passed|failed|rc=" $OUT/pytest.log | tail -12
