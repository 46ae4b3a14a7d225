#!/bin/bash
# What-if builds (measurement only, results wrong by construction -- no parity check): k_fp_slide5 without its mask-byte
# loads (f5nou), k_normals3 without its tail (n3wi1) / without its slide (n3wi2), against the current library (B).
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04b_exp4
mkdir -p $OUT
for rep in 1 2; do
  for v in B f5nou; do
    export TRAVGPU_LIB=$ROOT/traversability_estimation_amd/libtravgpu_$v.so
    python $ROOT/tools/ab_chain.py --tag $v.full >> $OUT/lines.jsonl 2>> $OUT/err.log
    python $ROOT/tools/ab_chain.py --tag $v.fp --footprint-only >> $OUT/lines.jsonl 2>> $OUT/err.log
  done
  for v in B n3wi1 n3wi2; do
    export TRAVGPU_LIB=$ROOT/traversability_estimation_amd/libtravgpu_$v.so
    python $ROOT/tools/ab_chain.py --tag $v.normals --normals-only >> $OUT/lines.jsonl 2>> $OUT/err.log
    python $ROOT/tools/ab_chain.py --tag $v.chain --no-footprint >> $OUT/lines.jsonl 2>> $OUT/err.log
  done
done
python - <<PY
import json
for l in open("$OUT/lines.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    print(d.get("tag"), round(d["ms_median"], 4), round(d["ms_p10"], 4))
PY
tail -5 $OUT/err.log
