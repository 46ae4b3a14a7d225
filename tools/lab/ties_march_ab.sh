# the TIES normals march at 9 cells: variant libraries (one shape, Q = 81) against each other -- general_tail3<EARLY>, fewer
# Newton steps (what-if).  Kernel times from rocprofv3 on the sequential launch.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/r06i; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for V in "$@"; do
  LIB=$ROOT/traversability_estimation_amd/libtravgpu_$V.so
  TRAVGPU_LIB=$LIB python $ROOT/tools/ab_chain.py --iters 30 --tag $V --exact-chain --radius-cells 9 > $O/$V.json 2> $O/$V.err
  TRAVGPU_LIB=$LIB rocprofv3 --kernel-trace --stats -d $O/kt_$V -o d --output-format csv -- python $ROOT/tools/ab_chain.py --iters 10 --tag $V --exact-chain --radius-cells 9 --sequential > $O/kt_$V.log 2>&1
  echo "== $V: $(tail -1 $O/$V.json | cut -c1-200)"
  python - <<PY
import csv, glob, re
for f in glob.glob("$O/kt_$V/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", r["Name"])
        if m and "normals" in m.group(0): print("  %-40s calls %4s avg %9.1f us" % (m.group(0), r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
find $O -name "*kernel_trace.csv" -delete
