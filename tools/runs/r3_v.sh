#!/bin/bash
# covariance builders with the in-house exp and the unrolled distance loop: parity, then the phase times
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r3v}
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu -k "not emul and not fullsize" > gpurun_out/${tag}_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/${tag}_pytest.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 200 python3 tools/step_breakdown.py 2>&1 | tail -12
exit 0
