#!/bin/bash
# Round 5: the random parity sweep on the final kernel sources (tests/test_gpu_random.py: whole-map cases and batches with
# dirty regions, memo layers' NaN patterns included) -> gpurun_out/r05_sweep/; then the driver's bench line once more.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r05_sweep
mkdir -p $O
cd $ROOT
TE_RANDOM_CASES="${SWEEP_CASES:-20000:5200}" TE_RANDOM_REGION_CASES="${SWEEP_REGIONS:-8000:200}" timeout 1200 python -m pytest tests/test_gpu_random.py -q -m gpu -n 16 > $O/pytest.log 2>&1
echo "rc=$?" >> $O/pytest.log
grep -E "^FAILED|passed|failed|rc=" $O/pytest.log | tail -12
grep -E "mismatches=[1-9]" $O/pytest.log | sort | uniq -c | sort -rn | head -12
export TMPDIR=/tmp
(timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?")
cut -c1-600 $O/bench_driver.json
