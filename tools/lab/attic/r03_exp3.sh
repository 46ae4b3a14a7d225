#!/bin/bash
TAG=${1:-r03_exp3}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
AB="python $ROOT/tools/ab_chain.py"
(cd $ROOT && timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15) > $O/pytest.log
tail -4 $O/pytest.log
$AB --tag bands2 --loops 20 > $O/ab_bands2.json 2>&1
TE_FP_BANDS=1 $AB --tag bands1 --loops 20 > $O/ab_bands1.json 2>&1
$AB --tag bands2_again > $O/ab_bands2b.json 2>&1
TE_FP_BANDS=1 $AB --tag bands1_again > $O/ab_bands1b.json 2>&1
$AB --tag fp_only_b2 --footprint-only > $O/ab_fp_b2.json 2>&1
TE_FP_BANDS=1 $AB --tag fp_only_b1 --footprint-only > $O/ab_fp_b1.json 2>&1
cat $O/ab_*.json | cut -c1-420
