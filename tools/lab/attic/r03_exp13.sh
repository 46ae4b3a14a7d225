#!/bin/bash
# tie radii in the chain: parity, then the bench map at exactly 9 / 5 cells
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/r03_exp13; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; ulimit -c 0
(cd $ROOT && timeout 300 python -m pytest tests/test_gpu_round3.py -m gpu -q -x -k "tie" 2>&1 | tail -15) > $O/pytest_tie.log
tail -8 $O/pytest_tie.log
grep -q "failed\|error" $O/pytest_tie.log && exit 1
(cd $ROOT && timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -5) > $O/pytest.log
tail -3 $O/pytest.log
for rc in 9 5; do
python $ROOT/tools/ab_chain.py --tag exact_chain_$rc --exact-chain --radius-cells $rc --iters 30 | cut -c1-300
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
python $ROOT/tools/ab_chain.py --tag default | cut -c1-250
TE_RANDOM_CASES=100:300 TE_RANDOM_REGION_CASES=700:60 timeout 600 python -m pytest $ROOT/tests/test_gpu_random.py -m gpu -q -x 2>&1 | tail -3
