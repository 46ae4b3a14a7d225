#!/bin/bash
# Round 5, experiment 11: where the sparse-hole march of k_normals3 spends its time (0.1 % speckle, 4096^2, R = 9; normals pass
# alone).  Variant libraries (tools/build_variant.sh h<k> te_normals3.hip -DTE_N3_HWHATIF=<k> '-DTE_N3_SHAPES(X)=X(81)'; results
# wrong by construction): h1 the invalid cells of a disc are not looked at, h2 they are walked but no cell waits for a general
# tail, h3 the cells are queued but the queue is never read.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_exp11
mkdir -p $OUT
P=$ROOT/traversability_estimation_amd
run() {  # tag, lib, env..., -- args
  local tag=$1 lib=$2; shift 2
  echo -n "$tag: "
  env TRAVGPU_LIB=$lib "$@" timeout 200 python $ROOT/tools/ab_chain.py --normals-only --iters 60 --tag $tag $ARGS 2>> $OUT/err.log | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['ms_median'], 4))"
}
ARGS="--holes 0"     run clean      $P/libtravgpu.so    X=1
ARGS="--holes 0"     run clean.nonslim $P/libtravgpu.so TE_N3_NO_SLIM=1
for h in 0.001 0.0003; do
ARGS="--holes $h" run full.$h   $P/libtravgpu.so    X=1
ARGS="--holes $h" run h1.$h     $P/libtravgpu_h1.so X=1
ARGS="--holes $h" run h2.$h     $P/libtravgpu_h2.so X=1
ARGS="--holes $h" run h3.$h     $P/libtravgpu_h3.so X=1
ARGS="--holes $h" run dense.$h  $P/libtravgpu.so    TE_N3_HOLES=2
done
ARGS="--holes 0.01"  run sparse.0.01 $P/libtravgpu.so   TE_N3_HOLES=1
ARGS="--holes 0.01"  run h1.0.01   $P/libtravgpu_h1.so TE_N3_HOLES=1
ARGS="--holes 0.01"  run h2.0.01   $P/libtravgpu_h2.so TE_N3_HOLES=1
ARGS="--holes 0.01"  run h3.0.01   $P/libtravgpu_h3.so TE_N3_HOLES=1
ARGS="--holes 0.01"  run dense.0.01 $P/libtravgpu.so   X=1
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/kt -o p --output-format csv -- python $ROOT/tools/ab_chain.py --normals-only --iters 30 --holes 0.001 > $OUT/kt.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
python - <<PY
import csv, glob, re
for f in glob.glob("$OUT/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-60s calls %5s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
tail -5 $OUT/err.log
