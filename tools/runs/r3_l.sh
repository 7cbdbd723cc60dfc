#!/bin/bash
# after moving the inverse-panel builds beside the factorisation: parity, then fused / separate / fit timings
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r3l}
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_emulation.py -x -q -m gpu > gpurun_out/${tag}_parity.log 2>&1; rc=$?; echo "parity rc=$rc"; tail -3 gpurun_out/${tag}_parity.log
[ $rc -ne 0 ] && exit 1
for st in 3 4; do timeout -k 10 120 python3 tools/fused_sweep.py pipe_stages=$st 2>&1 | tail -1; done | tee gpurun_out/${tag}_fused.txt
timeout -k 10 200 python3 tools/emul_fit_timing.py > gpurun_out/${tag}_emul.txt 2>&1; grep "^emulate\|fused" gpurun_out/${tag}_emul.txt | cut -c1-200
timeout -k 10 300 python3 tools/configs_timing.py 2>&1 | tee gpurun_out/${tag}_configs.txt | cut -c1-250
exit 0
