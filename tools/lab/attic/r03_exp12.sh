#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/r03_exp12; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; ulimit -c 0
for rc in 9 5; do
python $ROOT/tools/ab_chain.py --tag exact_chain_$rc --exact-chain --radius-cells $rc --iters 20 | cut -c1-300
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_$rc -o p --output-format csv -- python $ROOT/tools/ab_chain.py --sequential --iters 10 --exact-chain --radius-cells $rc > $O/kt_$rc.log 2>&1
python - <<PY
import csv, glob, re
for f in glob.glob("$O/kt_$rc/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", r["Name"])
        if m: print("  %-34s calls %4s avg %9.1f us" % (m.group(0), r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
