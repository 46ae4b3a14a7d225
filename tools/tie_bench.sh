#!/bin/bash
# Runs on the GPU box: the bench map (4096^2) at tie radii -- radius a whole number of cells, where the reference decides
# the cells exactly on the circle per centre -- through the tie paths of this round and through the generic kernels
# that served them before (TE_N3_NO_TIES / TE_STEP_NO_TIES / TE_F4_NO_TIES).  Event-timed launch, chain + footprint.
# Usage (gpurun): bash tools/tie_bench.sh <tag>  -> gpurun_out/<tag>/tie_radii.json   (torch-free: tools/ab_chain.py)
TAG=${1:-ties}; ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp; cd /tmp; ulimit -c 0
AB="python $ROOT/tools/ab_chain.py --iters 30"
run() { name=$1; shift; env "$@" > /dev/null 2>&1; }
one() {  # name, env..., -- args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  # (the switches are read by the lab build of the library only: libtravgpu.so never reads the environment)
  lib=(); [ ${#envs[@]} -gt 0 ] && [ -f $ROOT/traversability_estimation_amd/libtravgpu_lab.so ] && lib=(TRAVGPU_LIB=$ROOT/traversability_estimation_amd/libtravgpu_lab.so)
  env "${lib[@]}" "${envs[@]}" $AB --tag "$name" "$@" > $O/$name.json 2> $O/$name.err
}
one tie_free -- 
one footprint_9_cells -- --exact-cells
one footprint_9_cells_before TE_F4_NO_TIES=1 -- --exact-cells
one chain_9_cells -- --exact-chain --radius-cells 9
one chain_9_cells_before TE_N3_NO_TIES=1 TE_STEP_NO_TIES=1 -- --exact-chain --radius-cells 9
one chain_5_cells -- --exact-chain --radius-cells 5
one chain_5_cells_before TE_N3_NO_TIES=1 TE_STEP_NO_TIES=1 -- --exact-chain --radius-cells 5
one all_9_cells -- --exact-chain --exact-cells --radius-cells 9
one all_9_cells_before TE_N3_NO_TIES=1 TE_STEP_NO_TIES=1 TE_F4_NO_TIES=1 -- --exact-chain --exact-cells --radius-cells 9
python - > $O/tie_radii.json <<PY
import json
out = {}
for n in ("tie_free", "footprint_9_cells", "footprint_9_cells_before", "chain_9_cells", "chain_9_cells_before", "chain_5_cells", "chain_5_cells_before",
          "all_9_cells", "all_9_cells_before"):
    try:
        d = json.loads(open("$O/" + n + ".json").read().strip().splitlines()[-1])
        out[n] = {"ms_per_launch": round(d["ms_median"], 4), "cells_per_s": round(d["cells_per_s"]), "env": d["env"]}
    except Exception as e:
        out[n] = {"error": str(e)}
out["what"] = ("4096 x 4096, chain + footprint pass, median of 30 event-timed launches; *_cells: the named radii are exactly that many cells "
               "(normals / roughness / step radii for 'chain', radius + offset = 6 + 3 cells for 'footprint'), the others tie-free; "
               "*_before: the same with the tie paths switched off (generic kernels, as in round 2)")
print(json.dumps(out, indent=1))
PY
cat $O/tie_radii.json
