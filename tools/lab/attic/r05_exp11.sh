cd /tmp
R=$GRAFT_REPO_ROOT
for v in plain torch omp; do
  unset PREFETCH_AB_TORCH PREFETCH_AB_OMP
  [ $v = torch ] && export PREFETCH_AB_TORCH=1
  [ $v = omp ] && export PREFETCH_AB_OMP=1
  echo "== $v"; python $R/tools/lab/prefetch_ab.py 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('download_beside_prefetch_ms','three_plugins_ms','three_plugins_prefetch_ms')})"
done
cd $R
for k in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench no-check', {k:round(v,2) for k,v in d['host_path'].items() if k.endswith('ms')})"; done
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench with check', {k:round(v,2) for k,v in d['host_path'].items() if k.endswith('ms')})"
