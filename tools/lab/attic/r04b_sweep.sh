#!/bin/bash
# Wider random parity sweep on the final kernels (tests/test_gpu_random.py): whole-map cases and batches with dirty regions.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04b_sweep
TE_RANDOM_CASES="${SWEEP_CASES:-1000:500}" TE_RANDOM_REGION_CASES="${SWEEP_REGIONS:-2000:80}" timeout 1500 python -m pytest tests/test_gpu_random.py -q -m gpu -n 8 > gpurun_out/r04b_sweep/pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r04b_sweep/pytest.log
grep -E "^FAILED|passed|failed" gpurun_out/r04b_sweep/pytest.log | tail -12; grep -E "mismatches=[1-9]" gpurun_out/r04b_sweep/pytest.log | sort | uniq -c | sort -rn | head -12
