#!/bin/bash
TAG=${1:-r03_exp4}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
AB="python $ROOT/tools/ab_chain.py"
(cd $ROOT && timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15) > $O/pytest.log
tail -4 $O/pytest.log
(cd $ROOT && TE_RANDOM_CASES=100:300 timeout 600 python -m pytest tests/test_gpu_random.py -m gpu -q -x 2>&1 | tail -5) > $O/pytest_random.log
tail -3 $O/pytest_random.log
for h in 0 0.001 0.01 0.55; do $AB --tag holes_$h --holes $h > $O/ab_holes_$h.json 2>&1; done
$AB --tag normals_holes001 --holes 0.001 --normals-only > $O/ab_nholes001.json 2>&1
$AB --tag normals_clean --normals-only > $O/ab_nclean.json 2>&1
cat $O/ab_*.json | cut -c1-330
