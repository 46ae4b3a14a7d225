#!/bin/bash
# per-kernel times (sequential) of sparse obstacle maps
TAG=${1:-r03_exp6}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
AB="python $ROOT/tools/ab_chain.py"
for b in 3 30 300; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_seq_$b -o p --output-format csv -- $AB --sequential --iters 30 --boxes $b > $O/kt_seq_$b.log 2>&1
done
python - <<PY
import csv, glob, re
for d in ("kt_seq_3", "kt_seq_30", "kt_seq_300"):
    for f in glob.glob("$O/" + d + "/**/*kernel_stats.csv", recursive=True):
        print("==", d)
        for r in csv.DictReader(open(f)):
            m = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", r["Name"])
            if m: print("  %-34s calls %4s avg %9.1f us" % (m.group(0), r["Calls"], float(r["AverageNs"]) / 1e3))
PY
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
