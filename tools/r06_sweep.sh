#!/bin/bash
# Round 6: the seeded random sweep on the final kernel sources (k_normals_small / k_step_small take every launch of these
# map sizes whose discs reach at most two cells: about a sixth of the cases), a few thousand cases per call (tools/lab/README.md)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/r06_sweep; mkdir -p $O; cd $ROOT
FIRST=${1:-20000}; COUNT=${2:-4000}
(TE_RANDOM_CASES="$FIRST:$COUNT" TE_RANDOM_REGION_CASES="3000:400" timeout 2400 python -m pytest tests/test_gpu_random.py -m gpu -q -n 16 -p no:cacheprovider 2>&1 | tail -15) > $O/sweep_$FIRST.log
cat $O/sweep_$FIRST.log
