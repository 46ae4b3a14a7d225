#!/bin/bash
# Round 5, tenth GPU call: the mask kernel's pair masks one PAIR per thread (r05_exp9.sh: the pair masks are 26 of the 37 us
# a face tile adds with 3 boxes, the slow cells 12-18).  Same measurements as r05_exp7.sh.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_exp10
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -q -m gpu -n 4 > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
TE_RANDOM_CASES="${SWEEP_CASES:-12000:1500}" TE_RANDOM_REGION_CASES="${SWEEP_REGIONS:-5000:60}" timeout 900 python -m pytest tests/test_gpu_random.py -q -m gpu -n 16 > $OUT/sweep.log 2>&1
echo "sweep rc=$?" >> $OUT/sweep.log
grep -E "^FAILED|passed|failed|rc=" $OUT/sweep.log | tail -12
cd /tmp
P=$ROOT/traversability_estimation_amd
for v in B new; do
  unset TRAVGPU_LIB
  [ $v = B ] && export TRAVGPU_LIB=$P/libtravgpu_B.so
  python $ROOT/tools/ab_chain.py --tag $v.full $( [ $v = new ] && echo --check ) --loops 100 >> $OUT/lines.jsonl 2>> $OUT/err.log
  for b in 3 30 300 1000 3000; do
    python $ROOT/tools/ab_chain.py --tag $v.boxes$b --boxes $b --iters 40 >> $OUT/lines.jsonl 2>> $OUT/err.log
  done
done
unset TRAVGPU_LIB
for b in 300 1000 3000; do for w in 1 2; do
  python $ROOT/tools/ab_chain.py --tag new.boxes$b.walk$w --boxes $b --fb-walk $w --iters 40 >> $OUT/lines.jsonl 2>> $OUT/err.log
done; done
python $ROOT/tools/ab_chain.py --boxes 300 --res 0.0625 --iters 20 --tag check300 --check >> $OUT/lines.jsonl 2>> $OUT/err.log
python $ROOT/tools/ab_chain.py --boxes 3000 --res 0.0625 --iters 20 --tag check3000 --check >> $OUT/lines.jsonl 2>> $OUT/err.log
python $ROOT/tools/ab_chain.py --boxes 3000 --res 0.0625 --iters 20 --fb-walk 1 --tag check3000.walk1 --check >> $OUT/lines.jsonl 2>> $OUT/err.log
python $ROOT/tools/small_map_ab.py >> $OUT/small.jsonl 2>> $OUT/err.log
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/kt_cfg2 -o p --output-format csv -- python $ROOT/tools/ab_chain.py --size 1024 --radius-cells 5 --iters 200 > $OUT/kt_cfg2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/kt_bag -o p --output-format csv -- python $ROOT/tools/small_map_ab.py bagonly > $OUT/kt_bag.log 2>&1
for b in 0 3 300 3000; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/kt_b$b -o p --output-format csv -- python $ROOT/tools/ab_chain.py --sequential --iters 30 --boxes $b > $OUT/kt_b$b.log 2>&1
done
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
python - <<PY
import json, csv, glob, re
for l in open("$OUT/lines.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    hl = d.get("host_loops", {})
    print(d.get("tag"), round(d["ms_median"], 4), round(d["ms_p10"], 4), d.get("parity_check", {}).get("ok"), {k: round(v["ms_per_step"], 4) for k, v in hl.items()})
for l in open("$OUT/small.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    print(d["lib"], {k: round(v, 4) for k, v in d.items() if isinstance(v, float)})
for d in ("kt_cfg2", "kt_bag", "kt_b0", "kt_b3", "kt_b300", "kt_b3000"):
    for f in glob.glob("$OUT/" + d + "/**/*kernel_stats.csv", recursive=True):
        print("==", d)
        for r in csv.DictReader(open(f)):
            m = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", r["Name"])
            if m: print("  %-34s calls %4s avg %9.1f us" % (m.group(0), r["Calls"], float(r["AverageNs"]) / 1e3))
PY
tail -5 $OUT/err.log
