ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/r06g; mkdir -p $O; cd $ROOT
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30) > $O/pytest.log
python tools/defaults_bench.py > $O/defaults.json 2> $O/defaults.err
cat $O/pytest.log | tail -8; python - <<PY
import json
d=json.load(open("$O/defaults.json"))
for k,v in d.items(): print(k, {a:b["ms"] for a,b in v.items()})
PY
