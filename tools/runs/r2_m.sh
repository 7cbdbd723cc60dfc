#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 python tools/pair_timing.py > gpurun_out/r2m_timing.log 2>&1 || { tail -5 gpurun_out/r2m_timing.log; exit 1; }
cat gpurun_out/r2m_timing.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_shapes.py -x -q -p no:cacheprovider > gpurun_out/r2m_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2m_pytest.log
