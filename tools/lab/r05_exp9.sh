#!/bin/bash
# Round 5, ninth GPU call: where a face tile of k_fp_mask spends its time (lab library, TE_MASK_WHATIF: 1 no pair masks
# are evaluated, 2 the slow cells are not decided, 3 both; results wrong by construction) -- bench map with 3 boxes.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_exp9
mkdir -p $OUT
export TRAVGPU_LIB=$ROOT/traversability_estimation_amd/libtravgpu_lab.so
for w in 0 1 2 3; do
  TE_MASK_WHATIF=$w timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/kt_w$w -o p --output-format csv -- python $ROOT/tools/ab_chain.py --sequential --iters 30 --boxes 3 > $OUT/kt_w$w.log 2>&1
done
TE_MASK_WHATIF=0 timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/kt_300_w0 -o p --output-format csv -- python $ROOT/tools/ab_chain.py --sequential --iters 30 --boxes 300 > $OUT/kt_300_w0.log 2>&1
TE_MASK_WHATIF=1 timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/kt_300_w1 -o p --output-format csv -- python $ROOT/tools/ab_chain.py --sequential --iters 30 --boxes 300 > $OUT/kt_300_w1.log 2>&1
TE_MASK_WHATIF=3 timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/kt_300_w3 -o p --output-format csv -- python $ROOT/tools/ab_chain.py --sequential --iters 30 --boxes 300 > $OUT/kt_300_w3.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
python - <<PY
import csv, glob, re
for d in sorted(glob.glob("$OUT/kt_*/")):
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"k_fp_mask(<[^>]*>)?", r["Name"])
            if m: print("%-12s %-18s avg %9.1f us" % (d.rstrip("/").split("/")[-1], m.group(0), float(r["AverageNs"]) / 1e3))
PY
