# the kernels of the small launches one by one (cfg1 = the bag map, cfg2 = 1024^2; chain + footprint, one stream)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/r06h; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for C in cfg1 cfg2; do
  rocprofv3 --kernel-trace --stats -d $O/kt_$C -o d --output-format csv -- python $ROOT/bench.py --config $C --footprint --steps 200 --warmup 20 --sequential --no-cpu-baseline > $O/kt_$C.log 2>&1
  echo "== $C"
  python - <<PY
import csv, glob, re
for f in glob.glob("$O/kt_$C/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", r["Name"])
        if m: print("  %-34s calls %4s avg %9.1f us min %9.1f" % (m.group(0), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"])/1e3))
PY
  tail -2 $O/kt_$C.log | cut -c1-600
done
find $O -name "*kernel_trace.csv" -delete
