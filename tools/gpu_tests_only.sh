#!/bin/bash
# Runs on the GPU box: the GPU test suite only.  Usage (gpurun): bash tools/gpu_tests_only.sh <tag> [pytest args]
TAG=${1:-tests}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/$TAG; mkdir -p $O
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -q "$@" 2>&1 | tail -40 | tee $O/pytest.log
