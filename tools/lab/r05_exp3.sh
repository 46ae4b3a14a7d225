#!/bin/bash
# Round 5, third GPU call: the inner disc in k_fp_slide5 (only the discs whose untraversable cells all lie beyond the inner
# radius are listed), k_fp_blocked dealing out pages with one load per 64 page counts, k_fp_slide4 for tie radii only.
# Parity first, then obstacle maps against the round-4 library.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_exp3
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -q -m gpu -n 4 > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -8 $OUT/pytest.log
TE_RANDOM_CASES="${SWEEP_CASES:-9000:1500}" TE_RANDOM_REGION_CASES="${SWEEP_REGIONS:-4000:60}" timeout 900 python -m pytest tests/test_gpu_random.py -q -m gpu -n 16 > $OUT/sweep.log 2>&1
echo "sweep rc=$?" >> $OUT/sweep.log
grep -E "^FAILED|passed|failed|rc=" $OUT/sweep.log | tail -12
cd /tmp
P=$ROOT/traversability_estimation_amd
for v in B new; do
  unset TRAVGPU_LIB
  [ $v != new ] && export TRAVGPU_LIB=$P/libtravgpu_$v.so
  [ $v != new ] && [ ! -f $TRAVGPU_LIB ] && continue
  if [ $v = B ] || [ $v = new ]; then
    python $ROOT/tools/ab_chain.py --tag $v.full $( [ $v = new ] && echo --check ) >> $OUT/lines.jsonl 2>> $OUT/err.log
    python $ROOT/tools/ab_chain.py --tag $v.boxes3 --boxes 3 >> $OUT/lines.jsonl 2>> $OUT/err.log
    python $ROOT/tools/ab_chain.py --tag $v.boxes30 --boxes 30 >> $OUT/lines.jsonl 2>> $OUT/err.log
  fi
  python $ROOT/tools/ab_chain.py --tag $v.boxes300 --boxes 300 >> $OUT/lines.jsonl 2>> $OUT/err.log
  python $ROOT/tools/ab_chain.py --tag $v.boxes3000 --boxes 3000 --iters 40 >> $OUT/lines.jsonl 2>> $OUT/err.log
done
unset TRAVGPU_LIB
# parity of the obstacle maps at the dyadic resolution (the oracle sees crops: DESIGN.md section 2)
python $ROOT/tools/ab_chain.py --boxes 300 --res 0.0625 --iters 20 --tag check300 --check >> $OUT/lines.jsonl 2>> $OUT/err.log
python $ROOT/tools/ab_chain.py --boxes 3000 --res 0.0625 --iters 20 --tag check3000 --check >> $OUT/lines.jsonl 2>> $OUT/err.log
python $ROOT/tools/small_map_ab.py >> $OUT/small.jsonl 2>> $OUT/err.log
for b in 3 300 3000; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/kt_b$b -o p --output-format csv -- python $ROOT/tools/ab_chain.py --sequential --iters 30 --boxes $b > $OUT/kt_b$b.log 2>&1
done
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
python - <<PY
import json, csv, glob, re
for l in open("$OUT/lines.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    print(d.get("tag"), round(d["ms_median"], 4), round(d["ms_p10"], 4), d.get("parity_check", {}).get("ok"), d.get("parity_check", {}).get("mismatches"))
for l in open("$OUT/small.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    print(d["lib"], {k: round(v, 4) for k, v in d.items() if isinstance(v, float)})
for d in ("kt_b3", "kt_b300", "kt_b3000"):
    for f in glob.glob("$OUT/" + d + "/**/*kernel_stats.csv", recursive=True):
        print("==", d)
        for r in csv.DictReader(open(f)):
            m = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", r["Name"])
            if m: print("  %-34s calls %4s avg %9.1f us" % (m.group(0), r["Calls"], float(r["AverageNs"]) / 1e3))
PY
tail -5 $OUT/err.log
