#!/bin/bash
# Launch sizes re-swept on the final kernels (lab library): k_fp_slide5 / step kernels' waves per SIMD, k_normals3s blocks per CU.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04b_exp9
mkdir -p $OUT
export TRAVGPU_LIB=$ROOT/traversability_estimation_amd/libtravgpu_lab.so
run() { tag=$1; shift; env "$@" python $ROOT/tools/ab_chain.py --tag $tag --loops 100 >> $OUT/lines.jsonl 2>> $OUT/err.log; }
for rep in 1 2; do
  run base TE_X=0
  run f5w3 TE_F5_WAVES=3
  run f5w4 TE_F5_WAVES=4
  run f5w6 TE_F5_WAVES=6
  run f5w8 TE_F5_WAVES=8
  run stw3 TE_STEP_WAVES=3
  run stw5 TE_STEP_WAVES=5
  run n3b11 TE_N3_BLOCKS_PER_CU=11
  run n3b10 TE_N3_BLOCKS_PER_CU=10
  run noslim TE_N3_NO_SLIM=1
done
python - <<PY
import json
for l in open("$OUT/lines.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    hl = d.get("host_loops", {})
    print(d.get("tag"), round(d["ms_median"], 4), round(d["ms_p10"], 4), {k: round(v["ms_per_step"], 4) for k, v in hl.items()})
PY
tail -3 $OUT/err.log
