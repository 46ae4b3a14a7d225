#!/bin/bash
# k_fp_mask's "holds an untraversable cell" flags + k_fp_slide5 without mask-byte fetches on clean strips, against the
# recorded library (libtravgpu_B.so); 16-row mask tiles (lab library, TE_MASK_TILE=16).  Parity first.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04b_exp8
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py tests/test_gpu_chain.py "tests/test_gpu_fullsize.py::test_full_size_holes_and_obstacles_against_oracle_bands" "tests/test_gpu_fullsize.py::test_cfg4_true_size_batch_of_512_maps" "tests/test_gpu_fullsize.py::test_cfg5_true_size_streaming_8192" -x -q -m gpu > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
cd /tmp
for rep in 1 2; do for v in B new lab16; do
  unset TRAVGPU_LIB TE_MASK_TILE
  [ $v = B ] && export TRAVGPU_LIB=$ROOT/traversability_estimation_amd/libtravgpu_B.so
  [ $v = lab16 ] && export TRAVGPU_LIB=$ROOT/traversability_estimation_amd/libtravgpu_lab.so TE_MASK_TILE=16
  python $ROOT/tools/ab_chain.py --tag $v.full $( [ $rep = 1 ] && echo --check ) --loops 100 >> $OUT/lines.jsonl 2>> $OUT/err.log
  python $ROOT/tools/ab_chain.py --tag $v.fp --footprint-only >> $OUT/lines.jsonl 2>> $OUT/err.log
  python $ROOT/tools/ab_chain.py --tag $v.boxes3 --boxes 3 >> $OUT/lines.jsonl 2>> $OUT/err.log
  python $ROOT/tools/ab_chain.py --tag $v.boxes300 --boxes 300 $( [ $rep = 1 ] && echo --check ) >> $OUT/lines.jsonl 2>> $OUT/err.log
  python $ROOT/tools/ab_chain.py --tag $v.cfg4 --size 512 --batch 512 --radius-cells 5 --iters 30 >> $OUT/lines.jsonl 2>> $OUT/err.log
done; done
python - <<PY
import json
for l in open("$OUT/lines.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    hl = d.get("host_loops", {})
    print(d.get("tag"), round(d["ms_median"], 4), round(d["ms_p10"], 4), d.get("parity_check", {}).get("ok"), {k: round(v["ms_per_step"], 4) for k, v in hl.items()})
PY
tail -3 $OUT/err.log
