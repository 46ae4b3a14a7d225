#!/bin/bash
# round 4: k_fp_slide5 on the bench map (the list's slack kept it from launching in exp2), strips / prefetch depth of the
# te_march5 kernels, and their SQ counters
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/r04_exp3; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; ulimit -c 0
(cd $ROOT && timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -25) > $O/pytest.log
tail -4 $O/pytest.log
LAB=$ROOT/traversability_estimation_amd/libtravgpu_lab.so
[ -f $LAB ] || exit 0
export TRAVGPU_LIB=$LAB
AB="python $ROOT/tools/ab_chain.py"
$AB --tag new | tee $O/ab_new.json | cut -c1-230
TE_NO_F5=1 $AB --tag old_f4 | tee $O/ab_old_f4.json | cut -c1-230
for W in 2 3; do for C in 2 3 4; do TE_STEP_WAVES=$W TE_M5_C=$C $AB --tag sw${W}_c$C | tee $O/ab_sw${W}_c$C.json | cut -c1-230; done; done
TE_STEP_WAVES=4 TE_M5_C=4 $AB --tag sw4_c4 | tee $O/ab_sw4_c4.json | cut -c1-230
for W in 2 3 4 6; do TE_STEP_WAVES=3 TE_F5_WAVES=$W $AB --tag sw3_fw$W | tee $O/ab_sw3_fw$W.json | cut -c1-230; done
kt() {  # name, env...
  local n=$1; shift
  env "$@" timeout 120 rocprofv3 --kernel-trace --stats -d $O/kt_$n -o p --output-format csv -- $AB --sequential --iters 30 > $O/kt_$n.log 2>&1
}
kt base TE_X=1
kt sw3_c4 TE_STEP_WAVES=3 TE_M5_C=4
kt sw2_c4 TE_STEP_WAVES=2 TE_M5_C=4
kt sw4_c4 TE_STEP_WAVES=4 TE_M5_C=4
python - <<PY
import csv, glob, re
for d in ("kt_base", "kt_sw3_c4", "kt_sw2_c4", "kt_sw4_c4"):
    for f in glob.glob("$O/" + d + "/**/*kernel_stats.csv", recursive=True):
        print("==", d)
        for r in csv.DictReader(open(f)):
            m = re.search(r"k_[a-z0-9_]+", r["Name"])
            if m: print("  %-28s calls %4s avg %8.1f us" % (m.group(0), r["Calls"], float(r["AverageNs"]) / 1e3))
PY
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
P3="GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_ADD_F64"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P -d $O/pmc/p$i -o p --output-format csv -- $AB --sequential --iters 20 > $O/pmc_p$i.log 2>&1
done
python $ROOT/tools/sq_counters.py $O/pmc > $O/sq_counters.json 2> $O/sq_counters.err
python - <<PY
import json
d = json.load(open("$O/sq_counters.json"))
for k, v in d.items():
    c, dv = v["counters"], v["derived"]
    if "SQ_WAVES" not in c: continue
    print("%-40s waves %6.0f vgpr %3d lds %6d | valu %6.0f salu %6.0f lds %5.0f /wave | wait_inst_any %.2f wait_any %.2f wait_lds %.2f resident %.1f" % (
        k, c["SQ_WAVES"], v["vgpr"], v["lds_bytes"], dv.get("insts_per_wave_valu", 0), dv.get("insts_per_wave_salu", 0), dv.get("insts_per_wave_lds", 0),
        dv.get("sq_wait_inst_any_over_wave_cycles", 0), dv.get("sq_wait_any_over_wave_cycles", 0), dv.get("sq_wait_inst_lds_over_wave_cycles", 0), dv.get("mean_resident_waves_per_busy_sq_cycle", 0)))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -delete
