cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06f
for f in 1 2 4 8; do python bench.py --config cfg5 --steps 512 --warmup 32 --in-flight $f --no-check --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in-flight $f', 'ms_per_step', round(d['ms_per_step'],4), 'sync tick', round(d['tick_latency_ms'],4))"; done
