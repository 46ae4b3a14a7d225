#!/bin/bash
# Fused emit of the second step pass against the round-4 launch sequence (TE_OPT_NO_FUSED_EMIT), same box, same library:
# parity tests first, then timings of the bench map (clean, boxes, holes) and of the smaller configurations.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04b_exp2
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_fused_emit.py tests/test_gpu_round3.py -x -q -m gpu > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
cd /tmp
for rep in 1 2; do for v in fused nofused; do
  F=$( [ $v = nofused ] && echo --no-fused )
  python $ROOT/tools/ab_chain.py --tag $v.full $F $( [ $rep = 1 ] && echo --check ) --loops 20,100 >> $OUT/lines.jsonl 2>> $OUT/err.log
  python $ROOT/tools/ab_chain.py --tag $v.boxes3 $F --boxes 3 >> $OUT/lines.jsonl 2>> $OUT/err.log
  python $ROOT/tools/ab_chain.py --tag $v.boxes300 $F --boxes 300 $( [ $rep = 1 ] && echo --check ) >> $OUT/lines.jsonl 2>> $OUT/err.log
  python $ROOT/tools/ab_chain.py --tag $v.1024 $F --size 1024 --radius-cells 5 >> $OUT/lines.jsonl 2>> $OUT/err.log
  python $ROOT/tools/ab_chain.py --tag $v.seq $F --sequential >> $OUT/lines.jsonl 2>> $OUT/err.log
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
