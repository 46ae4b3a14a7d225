#!/bin/bash
# The library as it will be recorded (slim ring at R = 9 / 10 only) against libtravgpu_B.so; the round-4 tests first.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04b_exp7
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_round4.py "tests/test_gpu_chain.py::test_perlin_radius_sweep" -x -q -m gpu > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
cd /tmp
for rep in 1 2; do for v in B new; do
  if [ $v = B ]; then export TRAVGPU_LIB=$ROOT/traversability_estimation_amd/libtravgpu_B.so; else unset TRAVGPU_LIB; fi
  python $ROOT/tools/ab_chain.py --tag $v.full --loops 100 >> $OUT/lines.jsonl 2>> $OUT/err.log
  python $ROOT/tools/ab_chain.py --tag $v.cfg4 --size 512 --batch 512 --radius-cells 5 --iters 30 >> $OUT/lines.jsonl 2>> $OUT/err.log
  python $ROOT/tools/ab_chain.py --tag $v.8192 --size 8192 --radius-cells 5 --iters 30 >> $OUT/lines.jsonl 2>> $OUT/err.log
done; done
python - <<PY
import json
for l in open("$OUT/lines.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    hl = d.get("host_loops", {})
    print(d.get("tag"), round(d["ms_median"], 4), round(d["ms_p10"], 4), {k: round(v["ms_per_step"], 4) for k, v in hl.items()})
PY
tail -3 $OUT/err.log
