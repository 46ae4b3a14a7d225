#!/bin/bash
# k_fp_mask: tile staging by columns and the window maxima skipped in tiles without a lower step neighbour, against
# the shipped library of the round's first half (libtravgpu_A.so), same box.  Parity first.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04b_exp3
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_chain.py "tests/test_gpu_fullsize.py::test_full_size_holes_and_obstacles_against_oracle_bands" -x -q -m gpu > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
cd /tmp
for rep in 1 2; do for v in A new; do
  if [ $v = A ]; then export TRAVGPU_LIB=$ROOT/traversability_estimation_amd/libtravgpu_A.so; else unset TRAVGPU_LIB; fi
  python $ROOT/tools/ab_chain.py --tag $v.full $( [ $rep = 1 ] && echo --check ) --loops 20,100 >> $OUT/lines.jsonl 2>> $OUT/err.log
  python $ROOT/tools/ab_chain.py --tag $v.fp --footprint-only >> $OUT/lines.jsonl 2>> $OUT/err.log
  python $ROOT/tools/ab_chain.py --tag $v.boxes3 --boxes 3 >> $OUT/lines.jsonl 2>> $OUT/err.log
  python $ROOT/tools/ab_chain.py --tag $v.boxes300 --boxes 300 $( [ $rep = 1 ] && echo --check ) >> $OUT/lines.jsonl 2>> $OUT/err.log
  python $ROOT/tools/ab_chain.py --tag $v.1024 --size 1024 --radius-cells 5 >> $OUT/lines.jsonl 2>> $OUT/err.log
  python $ROOT/tools/ab_chain.py --tag $v.seq --sequential >> $OUT/lines.jsonl 2>> $OUT/err.log
done; done
python - <<PY
import json
for l in open("$OUT/lines.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    hl = d.get("host_loops", {})
    print(d.get("tag"), round(d["ms_median"], 4), round(d["ms_p10"], 4), d.get("parity_check", {}).get("ok"), d.get("parity_check", {}).get("max_abs_err"),
          {k: round(v["ms_per_step"], 4) for k, v in hl.items()})
PY
tail -5 $OUT/err.log
