#!/bin/bash
# Round 5, eighth GPU call: te_prefetch_layers (the next plugin's inputs on their way beside this plugin's kernel and
# download): the GPU suite incl. its parity test and the plugin driver, then bench.py's host_path block (three plugins
# with and without the prefetches), three times.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_exp8
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -q -m gpu -n 4 > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -8 $OUT/pytest.log
for k in 1 2 3; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_$k.json 2> $OUT/bench_$k.err
  python - <<PY
import json
d = json.loads(open("$OUT/bench_$k.json").read().strip().splitlines()[-1])
print("ms_per_step", round(d["ms_per_step"], 4), "host_path", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in d["host_path"].items() if k.endswith("ms")})
PY
done
tail -3 $OUT/bench_1.err
