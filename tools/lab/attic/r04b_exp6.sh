#!/bin/bash
# k_normals3s (slim ring, 12 blocks per CU) against the library before it (libtravgpu_B.so), same box; parity first.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04b_exp6
mkdir -p $OUT
cd $ROOT
TE_RANDOM_CASES="300:110" timeout 1200 python -m pytest tests/test_gpu_round4.py tests/test_gpu_chain.py tests/test_gpu_random.py "tests/test_gpu_fullsize.py::test_full_map_fast_vs_generic_and_oracle_crops" "tests/test_gpu_fullsize.py::test_cfg4_true_size_batch_of_512_maps" -x -q -m gpu > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -12 $OUT/pytest.log
cd /tmp
for rep in 1 2; do for v in B new; do
  if [ $v = B ]; then export TRAVGPU_LIB=$ROOT/traversability_estimation_amd/libtravgpu_B.so; else unset TRAVGPU_LIB; fi
  python $ROOT/tools/ab_chain.py --tag $v.full $( [ $rep = 1 ] && echo --check ) --loops 20,100 >> $OUT/lines.jsonl 2>> $OUT/err.log
  python $ROOT/tools/ab_chain.py --tag $v.normals --normals-only >> $OUT/lines.jsonl 2>> $OUT/err.log
  python $ROOT/tools/ab_chain.py --tag $v.chain --no-footprint >> $OUT/lines.jsonl 2>> $OUT/err.log
  python $ROOT/tools/ab_chain.py --tag $v.1024 --size 1024 --radius-cells 5 >> $OUT/lines.jsonl 2>> $OUT/err.log
  python $ROOT/tools/ab_chain.py --tag $v.cfg4 --size 512 --batch 512 --radius-cells 5 --iters 30 >> $OUT/lines.jsonl 2>> $OUT/err.log
  python $ROOT/tools/ab_chain.py --tag $v.8192 --size 8192 --radius-cells 5 --iters 30 >> $OUT/lines.jsonl 2>> $OUT/err.log
done; done
python - <<PY
import json
for l in open("$OUT/lines.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    hl = d.get("host_loops", {})
    print(d.get("tag"), round(d["ms_median"], 4), round(d["ms_p10"], 4), d.get("parity_check", {}).get("ok"), {k: round(v["ms_per_step"], 4) for k, v in hl.items()})
PY
tail -5 $OUT/err.log
