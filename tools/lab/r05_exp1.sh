#!/bin/bash
# Round 5, first GPU call: (a) the GPU suite with the scores-at-their-clip escape and the memoised checkForStep,
# (b) a wide random sweep, (c) A/B against the round-4 library (libtravgpu_B.so): clean bench map, boxes, small maps,
# (d) the what-if builds of the "one footprint kernel" question (timing only).
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_exp1
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -x -q -m gpu -n 4 > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -8 $OUT/pytest.log
TE_RANDOM_CASES="${SWEEP_CASES:-6000:2500}" TE_RANDOM_REGION_CASES="${SWEEP_REGIONS:-3000:100}" timeout 900 python -m pytest tests/test_gpu_random.py -q -m gpu -n 16 > $OUT/sweep.log 2>&1
echo "sweep rc=$?" >> $OUT/sweep.log
grep -E "^FAILED|passed|failed|rc=" $OUT/sweep.log | tail -12
cd /tmp
P=$ROOT/traversability_estimation_amd
for rep in 1 2; do for v in B new; do
  unset TRAVGPU_LIB
  [ $v = B ] && export TRAVGPU_LIB=$P/libtravgpu_B.so
  python $ROOT/tools/ab_chain.py --tag $v.full $( [ $rep = 1 ] && echo --check ) --loops 100 >> $OUT/lines.jsonl 2>> $OUT/err.log
  if [ $rep = 1 ]; then
    python $ROOT/tools/ab_chain.py --tag $v.normals --normals-only >> $OUT/lines.jsonl 2>> $OUT/err.log
    python $ROOT/tools/ab_chain.py --tag $v.boxes3 --boxes 3 >> $OUT/lines.jsonl 2>> $OUT/err.log
    python $ROOT/tools/ab_chain.py --tag $v.boxes300 --boxes 300 >> $OUT/lines.jsonl 2>> $OUT/err.log
    python $ROOT/tools/ab_chain.py --tag $v.boxes3000 --boxes 3000 --iters 40 >> $OUT/lines.jsonl 2>> $OUT/err.log
    python $ROOT/tools/small_map_ab.py >> $OUT/small.jsonl 2>> $OUT/err.log
  fi
done; done
# what-ifs (results wrong by construction): mask and sum kernels side by side; the sum kernel staging four layers itself
unset TRAVGPU_LIB
for rep in 1 2; do
  TRAVGPU_LIB=$P/libtravgpu_lab.so python $ROOT/tools/ab_chain.py --tag lab.full >> $OUT/lines.jsonl 2>> $OUT/err.log
  TRAVGPU_LIB=$P/libtravgpu_lab.so TE_FP_WHATIF_CONCURRENT=1 python $ROOT/tools/ab_chain.py --tag whatif.concurrent >> $OUT/lines.jsonl 2>> $OUT/err.log
  [ -f $P/libtravgpu_f5fused.so ] && TRAVGPU_LIB=$P/libtravgpu_f5fused.so python $ROOT/tools/ab_chain.py --tag whatif.f5fused >> $OUT/lines.jsonl 2>> $OUT/err.log
done
# per-kernel times of the new library, every kernel alone and in the two-stream launch
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/kt_seq -o p --output-format csv -- python $ROOT/tools/ab_chain.py --sequential --iters 40 > $OUT/kt_seq.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/kt_ovl -o p --output-format csv -- python $ROOT/tools/ab_chain.py --iters 40 > $OUT/kt_ovl.log 2>&1
[ -f $P/libtravgpu_f5fused.so ] && TRAVGPU_LIB=$P/libtravgpu_f5fused.so timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/kt_fused -o p --output-format csv -- python $ROOT/tools/ab_chain.py --sequential --iters 40 > $OUT/kt_fused.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/kt_b3 -o p --output-format csv -- python $ROOT/tools/ab_chain.py --sequential --iters 30 --boxes 3 > $OUT/kt_b3.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/kt_b300 -o p --output-format csv -- python $ROOT/tools/ab_chain.py --sequential --iters 30 --boxes 300 > $OUT/kt_b300.log 2>&1
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
for d in ("kt_seq", "kt_ovl", "kt_fused", "kt_b3", "kt_b300"):
    for f in glob.glob("$OUT/" + d + "/**/*kernel_stats.csv", recursive=True):
        print("==", d)
        for r in csv.DictReader(open(f)):
            m = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", r["Name"])
            print("  %-34s calls %4s avg %9.1f us" % (m.group(0) if m else r["Name"][:34], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
tail -5 $OUT/err.log
