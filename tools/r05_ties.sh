#!/bin/bash
# Round 5: the tie-radius rows of profiles/r05_tie_radii.json with the shipped library (the *_before rows -- generic kernels,
# lab library -- are tools/tie_bench.sh's and were recorded once).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/r05_ties; mkdir -p $O
export TMPDIR=/tmp; cd /tmp; ulimit -c 0
for c in "tie_free" "footprint_9_cells --exact-cells" "chain_9_cells --exact-chain --radius-cells 9" "chain_5_cells --exact-chain --radius-cells 5" "all_9_cells --exact-chain --exact-cells --radius-cells 9"; do
  set -- $c; name=$1; shift
  python $ROOT/tools/ab_chain.py --iters 30 --tag $name "$@" > $O/$name.json 2> $O/$name.err
  echo "$name: $(tail -1 $O/$name.json | python -c "import sys, json; print(round(json.loads(sys.stdin.read())['ms_median'], 4))")"
done
