#!/bin/bash
# k_normals3 register / priority variants against the shipped build, same box (libtravgpu_<v>.so built by tools/build_variant.sh):
#   A       the shipped library (151 VGPRs, static ring)
#   n3w4    ring declared extern: the compiler honours 4 waves per SIMD (127 VGPRs, no scratch)
#   n3p3    s_setprio 3 at kernel entry
#   n3w4p3  both
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04b_exp1
mkdir -p $OUT
for rep in 1 2; do for v in A n3w4 n3p3 n3w4p3; do
  export TRAVGPU_LIB=$ROOT/traversability_estimation_amd/libtravgpu_$v.so
  python $ROOT/tools/ab_chain.py --tag $v.full $( [ $rep = 1 ] && echo --check ) >> $OUT/lines.jsonl 2>> $OUT/err.log
  python $ROOT/tools/ab_chain.py --tag $v.normals --normals-only >> $OUT/lines.jsonl 2>> $OUT/err.log
  python $ROOT/tools/ab_chain.py --tag $v.chain --no-footprint >> $OUT/lines.jsonl 2>> $OUT/err.log
done; done
python - <<PY
import json
for l in open("$OUT/lines.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    print(d.get("tag"), round(d["ms_median"], 4), round(d["ms_p10"], 4), d.get("parity_check", {}).get("ok"), d.get("parity_check", {}).get("max_abs_err"))
PY
tail -5 $OUT/err.log
