#!/bin/bash
# bag map: 64 x 4 against 64 x 8 mask tiles (TE_MASK_SMALL_TILES=8 forces the latter); GPU tests first
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/maskab; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
(cd $ROOT && timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3)
for rep in 1 2; do
  python $ROOT/tools/small_map_ab.py bagonly
  TE_MASK_SMALL_TILES=8 python $ROOT/tools/small_map_ab.py bagonly
done
for e in "" "TE_MASK_SMALL_TILES=8"; do
  env $e rocprofv3 --kernel-trace --stats -d $O/kt -o p --output-format csv -- python $ROOT/tools/small_map_ab.py bagonly > $O/kt.log 2>&1
  echo "== $e"; python - <<PY
import csv, glob, re
for f in glob.glob("$O/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_fp_[a-z0-9_]+(<[^>]*>)?", r["Name"])
        if m: print("  %-34s calls %4s avg %9.1f us" % (m.group(0), r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $O/kt
done
