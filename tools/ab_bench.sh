#!/bin/bash
# A/B two (or more) builds of libtravgpu.so on the SAME GPU box: gpurun boxes differ by up to 10 %, so a kernel
# change is only judged against a baseline measured in the same call.
#
#   cp traversability_estimation_amd/libtravgpu.so traversability_estimation_amd/libtravgpu_A.so   # baseline
#   ... edit, python -m traversability_estimation_amd.build ...
#   cp traversability_estimation_amd/libtravgpu.so traversability_estimation_amd/libtravgpu_B.so
#   gpurun -- 'bash tools/ab_bench.sh A B'            # add --sequential etc. after the names via AB_FLAGS
#
# Prints ms per step / per launch and the per-kernel averages (rocprofv3 --kernel-trace --stats) per variant.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for v in "$@"; do
  export TRAVGPU_LIB=$ROOT/traversability_estimation_amd/libtravgpu_$v.so
  python $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-path $AB_FLAGS > /tmp/ab.log 2>&1
  line=$(tail -1 /tmp/ab.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_per_step=%.4f ms_per_launch=%.4f" % (d["ms_per_step"], d["roofline"]["ms_per_launch"]))')
  rocprofv3 --kernel-trace --stats -d /tmp/ab -o p --output-format csv -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-path $AB_FLAGS > /tmp/ab.log 2>&1
  kern=$(python - <<'PY'
import csv, re
out = []
for r in csv.DictReader(open('/tmp/ab/p_kernel_stats.csv')):
    m = re.search(r'k_[a-z_]+', r['Name'])
    if m:
        out.append('%s=%.1f' % (m.group(0)[2:], float(r['AverageNs']) / 1e3))
print(' '.join(out))
PY
)
  echo "variant=$v $line  $kern"
done; done
