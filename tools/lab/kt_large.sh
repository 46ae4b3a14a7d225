ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/r06j; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for N in 16384 32768; do
  timeout 800 rocprofv3 --kernel-trace --stats -d $O/kt_$N -o d --output-format csv -- python $ROOT/tools/dbg/large_map.py $N > $O/kt_$N.log 2>&1
  echo "== $N"
  python - <<PY
import csv, glob, re
for f in glob.glob("$O/kt_$N/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", r["Name"])
        print("  %-40s calls %4s avg %9.1f us" % (m.group(0) if m else r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
find $O -name "*kernel_trace.csv" -delete
