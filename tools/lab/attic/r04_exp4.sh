#!/bin/bash
# round 4: the footprint pass in bands (mask kernel of band k+1 beside the scatter-form sum of band k, lab build only)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/r04_exp4; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; ulimit -c 0
LAB=$ROOT/traversability_estimation_amd/libtravgpu_lab.so
export TRAVGPU_LIB=$LAB
AB="python $ROOT/tools/ab_chain.py"
$AB --tag base | tee $O/ab_base.json | cut -c1-230
for B in 2 3 4 6; do TE_FP_BANDS=$B $AB --tag bands$B | tee $O/ab_bands$B.json | cut -c1-230; done
for B in 2 4; do TE_FP_BANDS=$B TE_F5_WAVES=2 $AB --tag bands${B}_fw2 | tee $O/ab_bands${B}_fw2.json | cut -c1-230; done
$AB --tag base_fponly --footprint-only | tee $O/ab_base_fponly.json | cut -c1-230
for B in 2 4; do TE_FP_BANDS=$B $AB --tag bands${B}_fponly --footprint-only | tee $O/ab_b${B}_fponly.json | cut -c1-230; done
# parity of the banded pass: the bench's own check
TE_FP_BANDS=4 python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bands4 parity', d.get('parity_check'), d.get('ms_per_step'))"
