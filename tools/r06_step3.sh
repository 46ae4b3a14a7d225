ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/r06d; mkdir -p $O; cd $ROOT
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30) > $O/pytest.log
python tools/defaults_bench.py > $O/defaults.json 2> $O/defaults.err
python bench.py --config cfg1 --steps 200 --warmup 20 --no-cpu-baseline > $O/cfg1.json 2> $O/cfg1.err
python bench.py --config cfg1 --footprint --steps 200 --warmup 20 --no-cpu-baseline > $O/cfg1_fp.json 2> $O/cfg1_fp.err
cat $O/pytest.log; python - <<PY
import json
d=json.load(open("$O/defaults.json"))
for k,v in d.items(): print(k, {a:b["ms"] for a,b in v.items()})
for f in ("cfg1","cfg1_fp"):
    d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["latency_ms_per_launch"], d["parity_check"]["ok"])
PY
