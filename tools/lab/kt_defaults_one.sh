# the kernels of the default-parameter launch at ONE resolution (kt_defaults.sh mixes 0.05 and 0.03): bash kt_defaults_one.sh 0.05 4096
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/r06k; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for RES in "$@"; do
  rocprofv3 --kernel-trace --stats -d $O/kt_$RES -o d --output-format csv -- python $ROOT/tools/defaults_bench.py --one $RES 4096 profile sequential > $O/kt_$RES.log 2>&1
  echo "== res $RES"
  python - <<PY
import csv, glob, re
for f in glob.glob("$O/kt_$RES/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", r["Name"])
        if m: print("  %-34s calls %4s avg %9.1f us" % (m.group(0), r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
find $O -name "*kernel_trace.csv" -delete
