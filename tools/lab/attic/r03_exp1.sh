#!/bin/bash
# Round 3, GPU call 1: parity suite on the new kernels, the driver's bench command, A/B of the fixed-point footprint
# kernel and of the synchronisation polling, per-kernel times (sequential) and the overlapped timeline, new ubench ops.
TAG=${1:-r03_exp1}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
AB="python $ROOT/tools/ab_chain.py"

(cd $ROOT && timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > $O/pytest.log
tail -5 $O/pytest.log

# torch-free A/B (event-timed median of one launch; host loops = what bench.py's timed region does)
$AB --tag default --loops 1,5,20,100 > $O/ab_default.json 2> $O/ab_default.err
TE_NO_F4=1 $AB --tag no_f4 > $O/ab_no_f4.json 2>&1
TE_SYNC_SPIN_US=0 $AB --tag spin0 --loops 1,5,20,100 > $O/ab_spin0.json 2>&1
for b in 8 12 20; do TE_F4_BLOCKS_PER_CU=$b $AB --tag f4_blocks_$b --footprint-only > $O/ab_f4_b$b.json 2>&1; done
$AB --tag fp_only --footprint-only > $O/ab_fp_only.json 2>&1
TE_NO_F4=1 $AB --tag fp_only_no_f4 --footprint-only > $O/ab_fp_only_no_f4.json 2>&1
$AB --tag holes001 --holes 0.001 > $O/ab_holes001.json 2>&1
cat $O/ab_*.json | cut -c1-400

# per-kernel times: every kernel alone (sequential) and the default two-stream launch with its timeline
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
# timeline of the last overlapped launch: start / end of every kernel relative to the first one
for f in glob.glob("$O/kt_ovl/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [re.search(r"k_[a-z0-9_]+", r["Kernel_Name"]) for r in rows]
    # one launch = 6 kernels; print the last two launches
    tail = rows[-14:]
    t0 = int(tail[0]["Start_Timestamp"])
    with open("$O/timeline.txt", "w") as out:
        for r in tail:
            m = re.search(r"k_[a-z0-9_]+", r["Kernel_Name"])
            line = "%-28s start %9.1f  end %9.1f  dur %8.1f us  queue %s" % (m.group(0) if m else r["Kernel_Name"][:28], (int(r["Start_Timestamp"]) - t0) / 1e3,
                      (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"))
            print(line); out.write(line + "\n")
PY
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete

$ROOT/tools/ubench_valu 43 > $O/ubench_new_ops.txt 2>&1
cat $O/ubench_new_ops.txt

# the driver's command, last (it pays the torch import and the CPU baselines)
(cd $ROOT && timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?")
cut -c1-1500 $O/bench_driver.json
tail -3 $O/bench_driver.err
