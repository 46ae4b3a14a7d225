# the kernels of the default-parameter launches one by one (4096^2; one stream: every kernel's own time)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/r06g; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
TE_SIZES=4096 rocprofv3 --kernel-trace --stats -d $O/kt -o d --output-format csv -- python $ROOT/tools/defaults_bench.py profile sequential > $O/kt.log 2>&1
python - <<PY
import csv, glob, re
for f in glob.glob("$O/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", r["Name"])
        if m: print("  %-34s calls %4s avg %9.1f us min %9.1f" % (m.group(0), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"])/1e3))
PY
cat $O/kt.log | tail -12
find $O -name "*kernel_trace.csv" -delete
