#!/bin/bash
# Round 5, the record the docs quote, part 2: the driver's command with profiles/r05_hbm_traffic.json of the same kernel
# sources in place (the line then carries roofline.traffic), BASELINE configs[3] through bench.py (the 512-map batch, one
# GPU), and the timeline of one overlapped launch.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r05_record
mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
cd /tmp
(cd $ROOT && timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?")
cut -c1-2400 $O/bench_driver.json
(cd $ROOT && timeout 900 python bench.py --config cfg4 --gpus 1 > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "cfg4 rc=$?")
cut -c1-1600 $O/bench_cfg4.json
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_ovl -o p --output-format csv -- python $ROOT/tools/ab_chain.py --iters 30 > $O/kt_ovl.log 2>&1
python - <<PY
import csv, glob, re
for f in glob.glob("$O/kt_ovl/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    tail = rows[-16:]
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
