#!/bin/bash
# k_normals3's strip height for grids that exceed one round of resident blocks (launch3's cost model) on the 512-map
# batch of BASELINE configs[3], against the library before the change (libtravgpu_B.so: strips of 512 rows); batch tests.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04b_exp5
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest "tests/test_gpu_fullsize.py::test_cfg4_true_size_batch_of_512_maps" "tests/test_gpu_fullsize.py::test_batch_with_sparse_holes" "tests/test_gpu_fullsize.py::test_batch_of_512_maps_shape" "tests/test_gpu_chain.py::test_batch_of_maps" tests/test_gpu_multi.py -x -q -m gpu > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
cd /tmp
for rep in 1 2; do for v in B new; do
  if [ $v = B ]; then export TRAVGPU_LIB=$ROOT/traversability_estimation_amd/libtravgpu_B.so; else unset TRAVGPU_LIB; fi
  python $ROOT/tools/ab_chain.py --tag $v.cfg4 --size 512 --batch 512 --radius-cells 5 --iters 30 >> $OUT/lines.jsonl 2>> $OUT/err.log
  python $ROOT/tools/ab_chain.py --tag $v.cfg4.normals --size 512 --batch 512 --radius-cells 5 --iters 30 --normals-only >> $OUT/lines.jsonl 2>> $OUT/err.log
  python $ROOT/tools/ab_chain.py --tag $v.cfg4.chain --size 512 --batch 512 --radius-cells 5 --iters 30 --no-footprint >> $OUT/lines.jsonl 2>> $OUT/err.log
  python $ROOT/tools/ab_chain.py --tag $v.b64 --size 512 --batch 64 --radius-cells 5 --iters 50 >> $OUT/lines.jsonl 2>> $OUT/err.log
  python $ROOT/tools/ab_chain.py --tag $v.full >> $OUT/lines.jsonl 2>> $OUT/err.log
done; done
python - <<PY
import json
for l in open("$OUT/lines.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    print(d.get("tag"), round(d["ms_median"], 4), round(d["ms_p10"], 4))
PY
tail -5 $OUT/err.log
