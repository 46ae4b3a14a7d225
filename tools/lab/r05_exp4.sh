#!/bin/bash
# Round 5, fourth GPU call: which walk of k_fp_blocked serves the lists the inner disc leaves (cells whose untraversable
# cells all lie beyond the inner radius: long walks, a few cells per row, so the one-disc-per-lane walk's loads no longer
# coalesce) -- te_set_option(TE_OPT_FP_BLOCKED_WALK) 1 (per wavefront) / 2 (per lane) on the box maps.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_exp4
mkdir -p $OUT
for b in 3 30 300 1000 3000; do for w in 1 2; do
  python $ROOT/tools/ab_chain.py --tag boxes$b.walk$w --boxes $b --fb-walk $w --iters 40 >> $OUT/lines.jsonl 2>> $OUT/err.log
done; done
for b in 300 3000; do for w in 1 2; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/kt_b${b}_w$w -o p --output-format csv -- python $ROOT/tools/ab_chain.py --sequential --iters 20 --boxes $b --fb-walk $w > $OUT/kt_b${b}_w$w.log 2>&1
done; done
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
python - <<PY
import json, csv, glob, re
for l in open("$OUT/lines.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    print(d.get("tag"), round(d["ms_median"], 4), round(d["ms_p10"], 4))
for d in sorted(glob.glob("$OUT/kt_*/")):
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        print("==", d.rstrip("/").split("/")[-1])
        for r in csv.DictReader(open(f)):
            m = re.search(r"k_fp_[a-z0-9_]+(<[^>]*>)?", r["Name"])
            if m: print("  %-34s calls %4s avg %9.1f us" % (m.group(0), r["Calls"], float(r["AverageNs"]) / 1e3))
PY
tail -3 $OUT/err.log
