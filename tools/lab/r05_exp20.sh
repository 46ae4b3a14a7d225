#!/bin/bash
# Round 5, experiment 20: the kernels of one launch on the reference's bag map (100 x 133, default YAML, chain + footprint):
# rocprofv3 --kernel-trace --stats of tools/small_map_ab.py bagonly, and the timeline of two launches.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_exp20
mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/kt -o p --output-format csv -- python $ROOT/tools/small_map_ab.py bagonly > $OUT/kt.log 2>&1
tail -1 $OUT/kt.log
python - <<PY
import csv, glob, re
for f in glob.glob("$OUT/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("   %-70s calls %4s avg %9.1f us" % (re.sub(r"te::|\(anonymous namespace\)::|fast::", "", r["Name"])[:70], r["Calls"], float(r["AverageNs"]) / 1e3))
for f in glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    tail = rows[-20:]
    t0 = int(tail[0]["Start_Timestamp"])
    for r in tail:
        m = re.search(r"k_[a-z0-9_]+", r["Kernel_Name"])
        print("%-28s start %9.1f  end %9.1f  dur %8.1f us  grid %s wg %s" % (m.group(0) if m else r["Kernel_Name"][:28], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?")))
PY
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
