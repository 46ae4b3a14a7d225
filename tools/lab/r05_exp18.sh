#!/bin/bash
# Round 5, experiment 18: the kernels of a launch at TIE radii (normals / roughness / step radii exactly 9 and 5 cells), alone
# on the GPU (sequential launch, rocprofv3 --kernel-trace --stats), against the tie-free launch.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_exp18
mkdir -p $OUT
for c in "free --radius-cells 9" "tie9 --radius-cells 9 --exact-chain" "tie5 --radius-cells 5 --exact-chain" "free5 --radius-cells 5"; do
  set -- $c; tag=$1; shift
  timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/kt_$tag -o p --output-format csv -- python $ROOT/tools/ab_chain.py --sequential --iters 30 "$@" > $OUT/kt_$tag.log 2>&1
  echo "== $tag: $(tail -1 $OUT/kt_$tag.log | python -c "import sys, json; print(round(json.loads(sys.stdin.read())['ms_median'], 4))")"
  python - <<PY
import csv, glob, re
for f in glob.glob("$OUT/kt_$tag/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["AverageNs"]) > 3000 and int(r["Calls"]) > 20: print("   %-70s calls %4s avg %9.1f us" % (re.sub(r"te::|\(anonymous namespace\)::|fast::", "", r["Name"])[:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
