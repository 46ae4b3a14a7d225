#!/bin/bash
# round 4: the whole GPU suite with the new full-size parity tests (timed), then the driver's bench command (host path
# through the pinned staging ring, CPU thread ladder)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/r04_exp5; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; ulimit -c 0
(cd $ROOT && timeout 1500 python -m pytest tests -m gpu -q -x --durations=12 2>&1 | tail -40) > $O/pytest.log
tail -25 $O/pytest.log
(cd $ROOT && timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?")
python - <<PY
import json
d = json.load(open("$O/bench_driver.json"))
print({k: d[k] for k in ("value", "ms_per_step", "latency_ms_per_launch")})
print("host_path", d.get("host_path"))
print("cpu", d.get("cpu_baseline"), d.get("cpu_baseline_all_cores"))
print("parity", d.get("parity_check", {}).get("ok"), {k: v["mismatches"] for k, v in d.get("parity_check", {}).get("layers", {}).items()})
print("roofline", d["roofline"]["frac"], d["roofline"]["dominant_kernel"]["ms"])
PY
tail -3 $O/bench_driver.err
