#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_emulation.py -x -q -p no:cacheprovider > gpurun_out/r2o_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -8 gpurun_out/r2o_pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/emul_timing.py > gpurun_out/r2o_timing.log 2>&1 || { tail -5 gpurun_out/r2o_timing.log; exit 1; }
grep -v "max |mean" gpurun_out/r2o_timing.log
grep "max |mean" gpurun_out/r2o_timing.log | sort | uniq -c
timeout -k 10 300 python tools/emul_fit_timing.py > gpurun_out/r2o_fit.log 2>&1 || { tail -5 gpurun_out/r2o_fit.log; exit 1; }
cat gpurun_out/r2o_fit.log
