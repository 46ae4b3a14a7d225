#!/bin/bash
# round 4: the launch-wide list of slow mask cells + k_fp_mask_slow (one cell per wavefront): parity, then obstacles A/B
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/r04_exp6; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; ulimit -c 0
(cd $ROOT && timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -12) > $O/pytest.log
tail -5 $O/pytest.log
grep -q "failed\|error" $O/pytest.log && exit 1
bash $ROOT/tools/obstacles_bench.sh r04_obst_slowlist
export TRAVGPU_LIB=$ROOT/traversability_estimation_amd/libtravgpu_lab.so
for b in 0 3 300; do
  python $ROOT/tools/ab_chain.py --boxes $b --tag slowlist_$b | cut -c1-220
  TE_NO_SLOW_LIST=1 python $ROOT/tools/ab_chain.py --boxes $b --tag tilelist_$b | cut -c1-220
done
