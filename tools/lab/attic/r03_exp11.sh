#!/bin/bash
# tie radii through the fixed-point footprint kernel: parity, the bag map's kernels, a 4096^2 map at a whole-cell radius
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/r03_exp11; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; ulimit -c 0
(cd $ROOT && timeout 300 python -m pytest tests/test_gpu_round3.py -m gpu -q -x -k "tie or one_per_lane" 2>&1 | tail -12) > $O/pytest_tie.log
tail -6 $O/pytest_tie.log
grep -q "failed\|error" $O/pytest_tie.log && exit 1
(cd $ROOT && timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -5) > $O/pytest.log
tail -3 $O/pytest.log
python $ROOT/tools/small_map_ab.py bagonly > $O/ab.json 2> $O/ab.err; cat $O/ab.json
TE_F4_NO_TIES=1 python $ROOT/tools/small_map_ab.py bagonly > $O/ab_noties.json 2>> $O/ab.err; cat $O/ab_noties.json
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt -o p --output-format csv -- python $ROOT/tools/small_map_ab.py bagonly > $O/kt.log 2>&1
python - <<PY
import csv, glob, re
for f in glob.glob("$O/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", r["Name"])
        if m: print("  %-34s calls %4s avg %9.1f us  min %9.1f" % (m.group(0), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
# the bench map at exactly 9 + 6 + 3 cells (tie radii everywhere the parameters allow)
python $ROOT/tools/ab_chain.py --tag whole_cells --exact-cells > $O/ab_exact.json 2>&1; cut -c1-300 $O/ab_exact.json
TE_F4_NO_TIES=1 python $ROOT/tools/ab_chain.py --tag whole_cells_noties --exact-cells > $O/ab_exact_noties.json 2>&1; cut -c1-300 $O/ab_exact_noties.json
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_exact -o p --output-format csv -- python $ROOT/tools/ab_chain.py --sequential --iters 30 --exact-cells > $O/kt_exact.log 2>&1
python - <<PY
import csv, glob, re
for f in glob.glob("$O/kt_exact/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", r["Name"])
        if m: print("  %-34s calls %4s avg %9.1f us" % (m.group(0), r["Calls"], float(r["AverageNs"]) / 1e3))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
for b in 30 3000; do python $ROOT/tools/ab_chain.py --tag boxes_$b --boxes $b | cut -c1-250; done
