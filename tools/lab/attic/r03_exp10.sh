#!/bin/bash
# the bag map's kernels (100 x 133 at 0.03 m, default YAML: footprint radius 15 cells)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/r03_exp10; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; ulimit -c 0
python $ROOT/tools/small_map_ab.py bagonly > $O/ab.json 2> $O/ab.err; cat $O/ab.json
TE_NO_F4=1 python $ROOT/tools/small_map_ab.py bagonly > $O/ab_nof4.json 2>> $O/ab.err; cat $O/ab_nof4.json
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt -o p --output-format csv -- python $ROOT/tools/small_map_ab.py bagonly > $O/kt.log 2>&1
python - <<PY
import csv, glob, re
for f in glob.glob("$O/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", r["Name"])
        if m: print("  %-34s calls %4s avg %9.1f us  min %9.1f" % (m.group(0), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
