# k_fp_mask at the default parameters, res 0.05 against res 0.03 (4096^2): why twice the time?  SQ counters of the kernel.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/r06l; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for RES in 0.05 0.03; do
  for P in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"; do
    T=$(echo $P | cut -d' ' -f1)
    timeout 300 rocprofv3 --pmc $P -d $O/pmc_${RES}_$T -o p --output-format csv -- python $ROOT/tools/defaults_bench.py --one $RES 4096 profile sequential > $O/pmc_${RES}_$T.log 2>&1
  done
  echo "== res $RES"
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$O/pmc_${RES}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_fp_mask" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
disp = {}
for k, (v, n) in sorted(acc.items()):
    print("  %-24s %14.0f per launch (%d records)" % (k, v / max(n, 1) * (n / max(1, len(set([1])))) / 1, n))
PY
done
find $O -name "*counter_collection.csv" -size +20M -delete
