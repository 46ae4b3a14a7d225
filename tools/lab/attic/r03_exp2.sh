#!/bin/bash
# Round 3, GPU call: full GPU suite, A/B of the instruction diet (step emit masks, mask kernel, F4 single wait), step strip lengths.
TAG=${1:-r03_exp2}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
AB="python $ROOT/tools/ab_chain.py"
(cd $ROOT && timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/pytest.log
tail -4 $O/pytest.log
(cd $ROOT && RES=0.05 python tools/dbg/region_corner.py 2>&1 | grep -c "bad" ; python tools/dbg/region_corner.py 2>&1 | tail -6) > $O/region_dbg.log 2>&1
$AB --tag default --loops 20 > $O/ab_default.json 2>&1
TE_STEP_WAVES=2 $AB --tag step_waves2 > $O/ab_sw2.json 2>&1
TE_STEP_WAVES=4 $AB --tag step_waves4 > $O/ab_sw4.json 2>&1
$AB --tag fp_only --footprint-only > $O/ab_fp_only.json 2>&1
$AB --tag holes001 --holes 0.001 > $O/ab_holes001.json 2>&1
$AB --tag cfg2 --size 1024 --radius-cells 5 > $O/ab_cfg2.json 2>&1
cat $O/ab_*.json | cut -c1-330
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_seq -o p --output-format csv -- $AB --sequential --iters 30 > $O/kt_seq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_ovl -o p --output-format csv -- $AB --iters 30 > $O/kt_ovl.log 2>&1
python - <<PY
import csv, glob, re
for d in ("kt_seq", "kt_ovl"):
    for f in glob.glob("$O/" + d + "/**/*kernel_stats.csv", recursive=True):
        print("==", d)
        for r in csv.DictReader(open(f)):
            m = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", r["Name"])
            if m: print("  %-34s calls %4s avg %9.1f us" % (m.group(0), r["Calls"], float(r["AverageNs"]) / 1e3))
PY
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
