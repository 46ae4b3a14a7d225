#!/bin/bash
# Runs on the GPU box: the bench map (4096^2, radius 9 cells, footprint 6 + 3 cells) with raised / lowered boxes of
# 4..40 cells a side -- kerbs, crates: untraversable edges, so the footprint pass meets discs it has to walk
# (k_fp_mask's full checkForStep, k_fp_slide4's list, k_fp_blocked).  Whole launch (event-timed) and the footprint
# kernels alone (rocprofv3 --kernel-trace --stats, sequential launch).
# Usage (gpurun): bash tools/obstacles_bench.sh <tag>  -> gpurun_out/<tag>/obstacles.json   (torch-free: tools/ab_chain.py)
TAG=${1:-obstacles}; ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
ulimit -c 0
echo "[" > $O/obstacles.json
first=1
for b in 0 3 30 300 3000; do
  timeout 120 python $ROOT/tools/ab_chain.py --boxes $b --tag chain > $O/b_$b.json 2> $O/b_$b.err
  # parity of the same map, EVERY cell, at the BASELINE resolution: the OpenMP oracle on the whole map (round 3-4 checked
  # crops at a dyadic resolution: checkForStep's geometric ties depend on the absolute cell positions at 0.05 m, which a
  # crop does not share with the map -- the whole map does)
  timeout 900 python $ROOT/tools/ab_chain.py --boxes $b --iters 20 --tag check --check-whole > $O/c_$b.json 2>> $O/b_$b.err
  timeout 180 rocprofv3 --kernel-trace --stats -d $O/kt_$b -o p --output-format csv -- python $ROOT/tools/ab_chain.py --sequential --iters 30 --boxes $b > $O/kt_$b.log 2>&1
  python - >> $O/obstacles.json <<PY
import csv, glob, json, re
d = json.loads(open("$O/b_$b.json").read().strip().splitlines()[-1])
k = {}
for f in glob.glob("$O/kt_$b/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_[a-z0-9_]+", r["Name"])
        if m and m.group(0) != "k_count_invalid": k[m.group(0)] = round(float(r["AverageNs"]) / 1e3, 1)
pc = json.loads(open("$O/c_$b.json").read().strip().splitlines()[-1]).get("parity_check", {})
print(("" if $first else ",") + json.dumps({"boxes": $b, "ms_per_launch": round(d["ms_median"], 4), "kernel_us_alone": k,
                                            "parity_ok_whole_map_res_0.05": pc.get("ok"), "parity_mismatches": sum(pc.get("mismatches", {"-": -1}).values()), "parity_cells_per_layer": pc.get("cells_per_layer")}))
PY
  first=0
done
echo "]" >> $O/obstacles.json
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
cat $O/obstacles.json
