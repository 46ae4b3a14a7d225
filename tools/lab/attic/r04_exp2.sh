#!/bin/bash
# round 4: the te_march5.h kernels (k_step_height5 / k_step_score5 / k_fp_slide5) -- parity first (product library, the whole
# GPU suite), then A/B against the round-3 kernels on the same box (lab library: TE_OLD_STEP / TE_NO_F5) and per-kernel times
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/r04_exp2; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; ulimit -c 0
(cd $ROOT && timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -25) > $O/pytest.log
tail -6 $O/pytest.log
LAB=$ROOT/traversability_estimation_amd/libtravgpu_lab.so
if [ -f $LAB ]; then
  export TRAVGPU_LIB=$LAB
  python $ROOT/tools/ab_chain.py --tag new | tee $O/ab_new.json | cut -c1-260
  TE_OLD_STEP=1 python $ROOT/tools/ab_chain.py --tag old_step | tee $O/ab_old_step.json | cut -c1-260
  TE_NO_F5=1 python $ROOT/tools/ab_chain.py --tag old_f4 | tee $O/ab_old_f4.json | cut -c1-260
  TE_OLD_STEP=1 TE_NO_F5=1 python $ROOT/tools/ab_chain.py --tag old_both | tee $O/ab_old_both.json | cut -c1-260
  for W in 3 5 6; do TE_STEP_WAVES=$W python $ROOT/tools/ab_chain.py --tag step_waves_$W | tee $O/ab_sw$W.json | cut -c1-200; done
  for W in 3 4 6 8; do TE_F5_WAVES=$W python $ROOT/tools/ab_chain.py --tag f5_waves_$W | tee $O/ab_fw$W.json | cut -c1-200; done
  python $ROOT/tools/ab_chain.py --tag new_seq --sequential | tee $O/ab_new_seq.json | cut -c1-200
  timeout 120 rocprofv3 --kernel-trace --stats -d $O/kt -o p --output-format csv -- python $ROOT/tools/ab_chain.py --sequential --iters 30 > $O/kt.log 2>&1
  TE_OLD_STEP=1 TE_NO_F5=1 timeout 120 rocprofv3 --kernel-trace --stats -d $O/kt_old -o p --output-format csv -- python $ROOT/tools/ab_chain.py --sequential --iters 30 > $O/kt_old.log 2>&1
  python - <<PY
import csv, glob, re
for d in ("kt", "kt_old"):
    for f in glob.glob("$O/" + d + "/**/*kernel_stats.csv", recursive=True):
        print("==", d)
        for r in csv.DictReader(open(f)):
            m = re.search(r"k_[a-z0-9_]+", r["Name"])
            if m: print("  %-28s calls %4s avg %8.1f us" % (m.group(0), r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  python $ROOT/tools/lab/two_ctx.py --nctx 2 --tag n2_new | tee $O/two_ctx_n2.json | cut -c1-400
  find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
fi
