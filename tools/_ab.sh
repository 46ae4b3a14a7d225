cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for v in A B; do
export TRAVGPU_LIB=$GRAFT_REPO_ROOT/traversability_estimation_amd/libtravgpu_$v.so
rocprofv3 --kernel-trace --stats -d /tmp/hh -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-path > /tmp/hh.log 2>&1
echo "variant=$v $(tail -1 /tmp/hh.log | grep -o '"ms_per_launch": [0-9.]*') $(python - <<'PY'
import csv,re
out=[]
for r in csv.DictReader(open('/tmp/hh/p_kernel_stats.csv')):
    m=re.search(r'k_[a-z_]+',r['Name'])
    if m: out.append('%s=%.1f'%(m.group(0)[2:12], float(r['AverageNs'])/1e3))
print(' '.join(out))
PY
)"
done; done
