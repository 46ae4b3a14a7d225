#!/bin/bash
# Sparse holes: the clean first attempt skipped when nearly every strip holds an invalid cell (new) against the library
# that always attempts it (libtravgpu_noskip.so); then the wide random sweep on the new library.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04b_exp10
mkdir -p $OUT
for rep in 1 2; do for v in noskip new; do
  if [ $v = noskip ]; then export TRAVGPU_LIB=$ROOT/traversability_estimation_amd/libtravgpu_noskip.so; else unset TRAVGPU_LIB; fi
  for h in 0.0003 0.001 0.002; do
    python $ROOT/tools/ab_chain.py --tag $v.h$h --holes $h $( [ $rep = 1 ] && echo --check ) >> $OUT/lines.jsonl 2>> $OUT/err.log
    python $ROOT/tools/ab_chain.py --tag $v.h$h.normals --holes $h --normals-only >> $OUT/lines.jsonl 2>> $OUT/err.log
  done
done; done
python - <<PY
import json
for l in open("$OUT/lines.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    print(d.get("tag"), round(d["ms_median"], 4), round(d["ms_p10"], 4), d.get("parity_check", {}).get("ok"))
PY
tail -3 $OUT/err.log
unset TRAVGPU_LIB
SWEEP_CASES=5000:3000 SWEEP_REGIONS=9000:300 bash $ROOT/tools/lab/r04b_sweep.sh
