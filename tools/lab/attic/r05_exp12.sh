#!/bin/bash
# Why do bench.py's three-plugin lines disagree with tools/lab/prefetch_ab.py?  The same sequence with the differences one at a time.
cd /tmp
R=$GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python $R/tools/lab/prefetch_ab.py 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in d if 'three' in k})"; }
run A=1
run PREFETCH_AB_SKIP_MICRO=1
run PREFETCH_AB_SKIP_MICRO=1 PREFETCH_AB_ORDER=TF
run PREFETCH_AB_SKIP_MICRO=1 PREFETCH_AB_HOSTPATH=1
run PREFETCH_AB_SKIP_MICRO=1 PREFETCH_AB_HOSTPATH=1 PREFETCH_AB_TORCH=1
