#!/bin/bash
# round 4, first look: this box's baseline and how much N independent contexts launched round-robin gain over one
# (the upper bound of cross-launch pipelining without touching the kernels)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/r04_exp1; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; ulimit -c 0
python $ROOT/tools/ab_chain.py --tag base --loops 20,100 | tee $O/base.json | cut -c1-400
for N in 1 2 3; do
  python $ROOT/tools/lab/two_ctx.py --nctx $N --tag n$N | tee $O/two_ctx_n$N.json | cut -c1-500
done
TE_N3_BLOCKS_PER_CU=8 python $ROOT/tools/lab/two_ctx.py --nctx 2 --tag n2_n3x8 | tee $O/two_ctx_n2_n3x8.json | cut -c1-500
TE_N3_BLOCKS_PER_CU=9 TE_F4_BLOCKS_PER_CU=8 python $ROOT/tools/lab/two_ctx.py --nctx 2 --tag n2_n3x9_f4x8 | tee $O/two_ctx_n2_b.json | cut -c1-500
python $ROOT/tools/lab/two_ctx.py --nctx 2 --no-footprint --tag n2_nofp | tee $O/two_ctx_n2_nofp.json | cut -c1-500
python $ROOT/tools/lab/two_ctx.py --nctx 1 --no-footprint --tag n1_nofp | tee $O/two_ctx_n1_nofp.json | cut -c1-500
