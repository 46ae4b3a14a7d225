#!/bin/bash
# Round 6, the record the docs quote, part 2 (profiles/r06_hbm_traffic.json, r06_sq_counters.json and r06_valu_classes.json of
# the same kernel sources in place: the driver's line then carries roofline.traffic and the by-class ceiling): the driver's
# command; every other BASELINE configuration through bench.py --config; two gloo ranks through `bench.py --gpus 2` without
# a launcher; the reference's default parameters at common resolutions; the hole and tie maps of round 5 again (the kernels
# are unchanged: a stability check of the record); the roctx ranges; the timeline of one overlapped launch.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r06_record
mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
cd $ROOT
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"
cut -c1-1200 $O/bench_driver.json
for cfg in cfg1 cfg2 cfg5; do
  timeout 900 python bench.py --config $cfg > $O/bench_$cfg.json 2> $O/bench_$cfg.err; echo "$cfg rc=$?"
  timeout 900 python bench.py --config $cfg --footprint --no-cpu-baseline > $O/bench_${cfg}_footprint.json 2> $O/bench_${cfg}_footprint.err; echo "$cfg+fp rc=$?"
done
timeout 900 python bench.py --config cfg4 --gpus 1 > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "cfg4 rc=$?"
TE_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_gpus2_gloo.json 2> $O/bench_gpus2_gloo.err; echo "gpus2 rc=$?"
TE_DIST_BACKEND=gloo timeout 900 python bench.py --config cfg4 --gpus 2 --steps 10 --warmup 2 > $O/bench_cfg4_gpus2_gloo.json 2> $O/bench_cfg4_gpus2_gloo.err; echo "cfg4 gpus2 rc=$?"
python -c "from tests.test_params_yaml import SHIPPED, FOOTPRINT; open('/tmp/f.yaml','w').write(SHIPPED); open('/tmp/fp.yaml','w').write(FOOTPRINT)"
for n in 1024 4096; do
  timeout 900 python bench.py --yaml /tmp/f.yaml --footprint-yaml /tmp/fp.yaml --size $n --steps 50 --warmup 10 --no-cpu-all-cores > $O/bench_yaml_$n.json 2> $O/bench_yaml_$n.err; echo "yaml $n rc=$?"
done
timeout 600 python tools/defaults_bench.py > $O/defaults.json 2> $O/defaults.err
for h in 0.001 0.003 0.01 0.55 0.6 0.7; do
  timeout 600 python bench.py --holes $h --steps 50 --warmup 10 --no-cpu-baseline --no-host-path > $O/holes_$h.json 2> $O/holes_$h.err; echo "holes $h rc=$?"
done
cd /tmp
timeout 300 rocprofv3 --marker-trace --kernel-trace --stats -d $O/marker -o m --output-format csv -- python $ROOT/bench.py --config cfg2 --footprint --steps 5 --warmup 2 --no-cpu-baseline --no-check > $O/marker.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_ovl -o p --output-format csv -- python $ROOT/tools/ab_chain.py --iters 30 > $O/kt_ovl.log 2>&1
TE_SIZES=4096 timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_defaults -o d --output-format csv -- python $ROOT/tools/defaults_bench.py profile > $O/kt_defaults.log 2>&1
for c in cfg1 cfg2; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$c -o p --output-format csv -- python $ROOT/bench.py --config $c --footprint --steps 50 --warmup 5 --no-cpu-baseline --no-check > $O/kt_$c.log 2>&1
done
python - <<PY
import csv, glob, re
for f in glob.glob("$O/kt_ovl/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    tail = rows[-16:]
    t0 = int(tail[0]["Start_Timestamp"])
    with open("$O/timeline.txt", "w") as out:
        for r in tail:
            m = re.search(r"k_[a-z0-9_]+", r["Kernel_Name"])
            line = "%-28s start %9.1f  end %9.1f  dur %8.1f us  queue %s" % (m.group(0) if m else r["Kernel_Name"][:28], (int(r["Start_Timestamp"]) - t0) / 1e3,
                      (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"))
            print(line); out.write(line + "\n")
for d in ("kt_defaults", "kt_cfg1", "kt_cfg2"):
    for f in glob.glob("$O/" + d + "/**/*kernel_stats.csv", recursive=True):
        print("==", d)
        for r in csv.DictReader(open(f)):
            m = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", r["Name"])
            if m: print("  %-34s calls %4s avg %9.1f us" % (m.group(0), r["Calls"], float(r["AverageNs"]) / 1e3))
PY
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
