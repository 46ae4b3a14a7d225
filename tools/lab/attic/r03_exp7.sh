#!/bin/bash
# k_fp_blocked hybrid (one disc per lane for long lists, per wavefront for short ones): stop at the first failure
TAG=${1:-r03_exp7}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
ulimit -c 0
AB="python $ROOT/tools/ab_chain.py"
timeout 90 $AB --tag boxes_30 --boxes 30 > $O/ab_boxes_30.json 2>&1 || { echo "boxes 30 failed"; tail -3 $O/ab_boxes_30.json; exit 1; }
cut -c1-300 $O/ab_boxes_30.json
(cd $ROOT && timeout 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_random.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -5) > $O/pytest.log
tail -3 $O/pytest.log
grep -q "failed\|error" $O/pytest.log && exit 1
for b in 3 300 3000 30000; do timeout 90 $AB --tag boxes_$b --boxes $b > $O/ab_boxes_$b.json 2>&1; cut -c1-300 $O/ab_boxes_$b.json | tail -1; done
for b in 300 3000; do
  timeout 120 rocprofv3 --kernel-trace --stats -d $O/kt_seq_$b -o p --output-format csv -- $AB --sequential --iters 30 --boxes $b > $O/kt_seq_$b.log 2>&1
done
python - <<PY
import csv, glob, re
for d in ("kt_seq_300", "kt_seq_3000"):
    for f in glob.glob("$O/" + d + "/**/*kernel_stats.csv", recursive=True):
        print("==", d)
        for r in csv.DictReader(open(f)):
            m = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", r["Name"])
            if m and "fp_" in m.group(0): print("  %-34s calls %4s avg %9.1f us" % (m.group(0), r["Calls"], float(r["AverageNs"]) / 1e3))
PY
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
