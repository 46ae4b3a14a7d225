#!/bin/bash
# Round 3, GPU call: the footprint pass's deferred blocked-disc list (k_fp_slide4 appends, k_fp_blocked walks):
# parity suite + random sweep, then the obstacle-density sweep (event-timed launch) and the per-kernel times.
TAG=${1:-r03_exp5}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
AB="python $ROOT/tools/ab_chain.py"
(cd $ROOT && timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -15) > $O/pytest.log
tail -4 $O/pytest.log
(cd $ROOT && TE_RANDOM_CASES=100:300 timeout 600 python -m pytest tests/test_gpu_random.py -m gpu -q -x 2>&1 | tail -5) > $O/pytest_random.log
tail -3 $O/pytest_random.log
for b in 0 3 30 300 3000 30000; do $AB --tag boxes_$b --boxes $b > $O/ab_boxes_$b.json 2>&1; done
TE_NO_F4=1 $AB --tag boxes_3000_nof4 --boxes 3000 > $O/ab_boxes_3000_nof4.json 2>&1
$AB --tag default --loops 1,20,100 > $O/ab_default.json 2>&1
cat $O/ab_*.json | cut -c1-420
for b in 3 300 3000; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_seq_$b -o p --output-format csv -- $AB --sequential --iters 30 --boxes $b > $O/kt_seq_$b.log 2>&1
done
python - <<PY
import csv, glob, re
for d in ("kt_seq_3", "kt_seq_300", "kt_seq_3000"):
    for f in glob.glob("$O/" + d + "/**/*kernel_stats.csv", recursive=True):
        print("==", d)
        for r in csv.DictReader(open(f)):
            m = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", r["Name"])
            if m: print("  %-34s calls %4s avg %9.1f us" % (m.group(0), r["Calls"], float(r["AverageNs"]) / 1e3))
PY
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
