#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_tail.py -x -q -p no:cacheprovider > gpurun_out/r2h_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -15 gpurun_out/r2h_pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/tail_timing.py > gpurun_out/r2h_timing.log 2>&1 || { tail -5 gpurun_out/r2h_timing.log; exit 1; }
cat gpurun_out/r2h_timing.log
