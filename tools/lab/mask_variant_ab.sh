ROOT=$GRAFT_REPO_ROOT; cd $ROOT
for L in "" "$ROOT/traversability_estimation_amd/libtravgpu_maskv.so"; do
  echo "== lib: ${L:-product}"
  for i in 1 2; do TRAVGPU_LIB=$L python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(' headline ms/step', round(d['ms_per_step'],4), 'launch', round(d['latency_ms_per_launch'],4), 'parity', d['parity_check']['ok'])"; done
  TRAVGPU_LIB=$L TE_SIZES=4096 python tools/defaults_bench.py 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items(): print(' ',k, {a:round(b['ms'],4) for a,b in v.items() if 'generic' not in a})"
done
