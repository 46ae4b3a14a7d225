#!/bin/bash
# mask kernel without the list's barriers: parity, timing, and -- if green -- the traffic passes and the bench record again
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/r03_exp14; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; ulimit -c 0
(cd $ROOT && timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -5) > $O/pytest.log
tail -3 $O/pytest.log
grep -q "failed\|error" $O/pytest.log && exit 1
for b in 0 3 300; do
  python $ROOT/tools/ab_chain.py --boxes $b --tag boxes_$b | cut -c1-200
done
timeout 120 rocprofv3 --kernel-trace --stats -d $O/kt -o p --output-format csv -- python $ROOT/tools/ab_chain.py --sequential --iters 30 > $O/kt.log 2>&1
python - <<PY
import csv, glob, re
for f in glob.glob("$O/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_[a-z0-9_]+", r["Name"])
        if m: print("  %-28s avg %8.1f us" % (m.group(0), float(r["AverageNs"]) / 1e3))
PY
BENCH="python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-check"
i=3
for P in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P -d $O/pmc/p$i -o p --output-format csv -- $BENCH > $O/pmc_p$i.log 2>&1
done
python $ROOT/tools/hbm_traffic.py $O/pmc/p4 $O/pmc/p5 > $O/hbm_traffic.json 2> $O/hbm.err
head -c 300 $O/hbm_traffic.json
cp $O/hbm_traffic.json $ROOT/profiles/r03_hbm_traffic.json
(cd $ROOT && timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?")
cut -c1-330 $O/bench_driver.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
