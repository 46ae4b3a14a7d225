#!/bin/bash
# PMC counters of the footprint kernels on an obstacle map (3000 boxes)
TAG=${1:-r03_exp8}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
ulimit -c 0
AB="python $ROOT/tools/ab_chain.py --sequential --iters 10 --boxes ${BOXES:-3000}"
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
P3="GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_ADD_F64 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"
i=0
for P in "$P1" "$P2" "$P3" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $P -d $O/pmc/p$i -o p --output-format csv -- $AB > $O/pmc_p$i.log 2>&1 || tail -3 $O/pmc_p$i.log
done
python $ROOT/tools/sq_counters.py $O/pmc > $O/sq_counters.json 2> $O/sq_counters.err
python - <<PY
import json
d = json.load(open("$O/sq_counters.json"))
for k, v in d.items():
    if "fp_" in k:
        print(k, json.dumps(v)[:1800])
PY
find $O -name "*agent_info.csv" -delete
