#!/bin/bash
TAG=${1:-r03_exp9}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
ulimit -c 0
AB="python $ROOT/tools/ab_chain.py"
timeout 90 $AB --tag boxes_30 --boxes 30 > $O/ab_boxes_30.json 2>&1 || { echo "boxes 30 failed"; tail -3 $O/ab_boxes_30.json; exit 1; }
(cd $ROOT && timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -5) > $O/pytest.log
tail -3 $O/pytest.log
grep -q "failed\|error" $O/pytest.log && exit 1
(cd $ROOT && TE_RANDOM_CASES=100:300 timeout 600 python -m pytest tests/test_gpu_random.py -m gpu -q -x 2>&1 | tail -3) > $O/pytest_random.log
tail -2 $O/pytest_random.log
bash $ROOT/tools/obstacles_bench.sh $TAG/obst > $O/obst.log 2>&1
cat $O/obst/obstacles.json
