#!/bin/bash
# A/B of tools/small_map_ab.py on one GPU box: bash tools/small_map_ab.sh <tag>
TAG=${1:-smallab}; ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
(cd $ROOT && timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -5) > $O/pytest.log
P=$ROOT/traversability_estimation_amd
run() { env "$@" python $ROOT/tools/small_map_ab.py >> $O/ab.jsonl 2>> $O/ab.err; }
run TRAVGPU_LIB=$P/libtravgpu_A.so
run TRAVGPU_LIB=$P/libtravgpu_B.so
for m in 8 2; do
  run TRAVGPU_LIB=$P/libtravgpu_B.so TE_F3_MIN_STRIP=$m TE_FP_MIN_STRIP=$m
done
run TRAVGPU_LIB=$P/libtravgpu_A.so
run TRAVGPU_LIB=$P/libtravgpu_B.so
kt() {
  name=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats -d $O/kt_$name -o p --output-format csv -- python $ROOT/tools/small_map_ab.py bagonly > $O/kt_$name.log 2>&1
  echo "== bag map kernels, $name"
  python - <<PY
import csv, glob, re
for f in glob.glob("$O/kt_$name/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", r["Name"])
        if m: print("  %-34s calls %4s avg %9.1f us" % (m.group(0), r["Calls"], float(r["AverageNs"]) / 1e3))
PY
}
kt A TRAVGPU_LIB=$P/libtravgpu_A.so
kt B TRAVGPU_LIB=$P/libtravgpu_B.so
kt B_strip8 TRAVGPU_LIB=$P/libtravgpu_B.so TE_F3_MIN_STRIP=8 TE_FP_MIN_STRIP=8
for v in A B; do
  TRAVGPU_LIB=$P/libtravgpu_$v.so python $ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-host-path > $O/bench_$v.json 2> $O/bench_$v.err
  TRAVGPU_LIB=$P/libtravgpu_$v.so rocprofv3 --kernel-trace --stats -d $O/ktb_$v -o p --output-format csv -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-path --sequential > $O/ktb_$v.log 2>&1
  for h in 0.001 0.6; do
    TRAVGPU_LIB=$P/libtravgpu_$v.so python $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-path --holes $h > $O/bench_${v}_h$h.json 2> $O/bench_${v}_h$h.err
    python - <<PY
import json
d = json.loads(open("$O/bench_${v}_h$h.json").read().strip().splitlines()[-1]); r = d["roofline"]
print("   holes $h variant $v: launch %.4f ms, normals pass %.4f ms" % (r["ms_per_launch"], r["dominant_kernel"]["ms"]))
PY
  done
  echo "== bench map, variant $v: $(python -c "import json;d=json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline']['dominant_kernel']['ms'])")"
  python - <<PY
import csv, glob, re
for f in glob.glob("$O/ktb_$v/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", r["Name"])
        if m: print("  %-34s calls %4s avg %9.1f us" % (m.group(0), r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
find $O -name "*kernel_trace.csv" -delete
cat $O/pytest.log
python - <<PY
import json
for l in open("$O/ab.jsonl"):
    d = json.loads(l)
    print(d["lib"], d["env"], " ".join("%s=%.4g" % (k, v) for k, v in d.items() if k not in ("lib", "env")))
PY
tail -5 $O/ab.err
