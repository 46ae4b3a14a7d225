#!/bin/bash
# Runs on the GPU box: GPU test suite, one bench line, per-kernel averages (overlapped and sequential).
# Usage (gpurun): bash tools/gpu_check.sh <tag> [extra env assignments for the profiled runs, e.g. TE_NO_F3=1]
TAG=${1:-check}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
(cd $ROOT && timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/pytest.log
env "$@" python $ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-host-path > $O/bench.json 2>$O/bench.err
for m in "" "--sequential"; do
  env "$@" rocprofv3 --kernel-trace --stats -d $O/kt$m -o p --output-format csv -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-path $m > $O/kt$m.log 2>&1
done
python - <<PY
import csv, glob, re
for d in ("kt","kt--sequential"):
    for f in glob.glob("$O/" + d + "/**/*kernel_stats.csv", recursive=True):
        print("==", d)
        for r in csv.DictReader(open(f)):
            m = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", r["Name"])
            if m: print("  %-34s calls %4s avg %9.1f us" % (m.group(0), r["Calls"], float(r["AverageNs"]) / 1e3))
PY
find $O -name "*kernel_trace.csv" -delete
tail -12 $O/pytest.log; cut -c1-330 $O/bench.json
