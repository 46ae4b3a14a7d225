#!/bin/bash
# Runs on the GPU box: the normals / slope / roughness pass (alone) and the whole launch on maps with invalid cells.
# Usage (gpurun): bash tools/holes_bench.sh <tag>  -> gpurun_out/<tag>/holes.json
TAG=${1:-holes}; ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
echo "[" > $O/holes.json
first=1
for h in 0 0.001 0.01 0.55 0.6 0.7; do
  python $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-path --holes $h > $O/h_$h.json 2> $O/h_$h.err
  python - >> $O/holes.json <<PY
import json
d = json.loads(open("$O/h_$h.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(("" if $first else ",") + json.dumps({"holes": $h, "ms_per_launch": round(r["ms_per_launch"], 4), "normals_pass_ms": round(r["dominant_kernel"]["ms"], 4)}))
PY
  first=0
done
echo "]" >> $O/holes.json
cat $O/holes.json
