#!/bin/bash
# Runs on the GPU box (gpurun): GPU test suite, bench lines, rocprofv3 kernel statistics and PMC passes of the bench
# workload.  Usage: tools/profile_round.sh <tag> [quick]   -> gpurun_out/<tag>/...
# PMC passes are separate runs with --pmc only (never combined with trace domains).
TAG=${1:-r04}
QUICK=${2:-}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $ROOT/bench.py"
PROF_ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-check"  # (bench.py itself adds 2 x 100 event-timed launches)

(cd $ROOT && timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > $OUT/pytest.log
timeout 900 $BENCH --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err

timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_overlap -o p --output-format csv -- $BENCH $PROF_ARGS > $OUT/kt_overlap.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_seq -o p --output-format csv -- $BENCH $PROF_ARGS --sequential > $OUT/kt_seq.log 2>&1
if [ -z "$QUICK" ]; then
  P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
  P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
  P3="GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_ACTIVE_INST_VMEM"
  i=0
  for P in "$P1" "$P2" "$P3" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    SEQ=--sequential
    [ $i -ge 4 ] && SEQ=   # the traffic passes measure the default (two-stream, combine in the mask kernel) launch sequence
    timeout 600 rocprofv3 --pmc $P -d $OUT/pmc/p$i -o p --output-format csv -- $BENCH $PROF_ARGS $SEQ > $OUT/pmc_p$i.log 2>&1
  done
  python $ROOT/tools/sq_counters.py $OUT/pmc > $OUT/sq_counters.json 2> $OUT/sq_counters.err
  python $ROOT/tools/hbm_traffic.py $OUT/pmc/p4 $OUT/pmc/p5 > $OUT/hbm_traffic.json 2>> $OUT/sq_counters.err
fi
# keep what is merged back small: drop the raw per-dispatch traces, keep the statistics
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
ls -la $OUT
cat $OUT/pytest.log | tail -5
cat $OUT/bench_default.json
python - <<PY
import csv, glob, re
for d in ("kt_overlap", "kt_seq"):
    for f in glob.glob("$OUT/" + d + "/**/*kernel_stats.csv", recursive=True):
        print("==", d)
        for r in csv.DictReader(open(f)):
            m = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", r["Name"])
            print("  %-34s calls %4s avg %9.1f us" % (m.group(0) if m else r["Name"][:34], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
