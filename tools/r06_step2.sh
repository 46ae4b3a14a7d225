ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/r06c; mkdir -p $O; cd $ROOT
(timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round3.py tests/test_gpu_chain.py -m gpu -x -q 2>&1 | tail -30) > $O/pytest6.log
python tools/defaults_bench.py > $O/defaults.json 2> $O/defaults.err
cd /tmp; export TMPDIR=/tmp
TE_SIZES=4096 rocprofv3 --kernel-trace --stats -d $O/defaults_kt -o d --output-format csv -- python $ROOT/tools/defaults_bench.py profile > $O/defaults_kt.log 2>&1
find $O/defaults_kt -name "*kernel_trace.csv" -delete
cat $O/pytest6.log; python - <<PY
import json, csv, re, glob
d=json.load(open("$O/defaults.json"))
for k,v in d.items(): print(k, {a:b["ms"] for a,b in v.items()})
for f in glob.glob("$O/defaults_kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m=re.search(r"k_[a-z0-9_]+(<[^>]*>)?", r["Name"])
        if m: print("  %-34s calls %4s avg %9.1f us" % (m.group(0), r["Calls"], float(r["AverageNs"]) / 1e3))
PY
