#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_emulation.py -x -q -s -p no:cacheprovider -k "trailing or emulated_fit" > gpurun_out/r2j_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -25 gpurun_out/r2j_pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/emul_fit_timing.py > gpurun_out/r2j_timing.log 2>&1 || { tail -5 gpurun_out/r2j_timing.log; exit 1; }
cat gpurun_out/r2j_timing.log
