cd /tmp && export TMPDIR=/tmp
for h in 0 0.001 0.55 0.7; do
rocprofv3 --kernel-trace --stats -d /tmp/hh -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --sequential --holes $h > /tmp/hh.log 2>&1
echo "holes=$h"
python - <<'PY'
import csv,re
for r in csv.DictReader(open('/tmp/hh/p_kernel_stats.csv')):
    m=re.search(r'k_[a-z_]+(<[-\d, ]+>)?',r['Name'])
    if m and 'normals' in m.group(0): print('   ',m.group(0), round(float(r['AverageNs'])/1e3,1))
PY
done
