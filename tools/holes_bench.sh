#!/bin/bash
# Runs on the GPU box: the whole launch and the normals / slope / roughness pass alone on maps with invalid cells
# (speckle for fractions < 0.5, solid unobserved rectangles covering (fraction - 0.5) of the map above).
# Every timed configuration is checked against the oracle on EVERY cell of the map (ab_chain.py --check-whole).
# Usage (gpurun): bash tools/holes_bench.sh <tag>  -> gpurun_out/<tag>/holes.json   (torch-free: tools/ab_chain.py)
TAG=${1:-holes}; ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
echo "[" > $O/holes.json
first=1
for h in 0 0.001 0.003 0.01 0.55 0.6 0.7; do
  python $ROOT/tools/ab_chain.py --holes $h --tag chain --check-whole > $O/h_$h.json 2> $O/h_$h.err
  python $ROOT/tools/ab_chain.py --holes $h --normals-only --tag normals > $O/hn_$h.json 2>> $O/h_$h.err
  python - >> $O/holes.json <<PY
import json
d = json.loads(open("$O/h_$h.json").read().strip().splitlines()[-1])
n = json.loads(open("$O/hn_$h.json").read().strip().splitlines()[-1])
pc = d.get("parity_check", {})
print(("" if $first else ",") + json.dumps({"holes": $h, "ms_per_launch": round(d["ms_median"], 4), "normals_pass_ms": round(n["ms_median"], 4),
                                            "parity_ok": pc.get("ok"), "parity_mismatches": sum(pc.get("mismatches", {"-": -1}).values()), "parity_cells_per_layer": pc.get("cells_per_layer")}))
PY
  first=0
done
echo "]" >> $O/holes.json
cat $O/holes.json
