#!/bin/bash
# emulated candidate solve pipelined behind the emulated factorisation: parity (both modes), then timings
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r3n}
mkdir -p gpurun_out
timeout -k 10 200 python3 tools/emul_fit_timing.py > gpurun_out/${tag}_emul.txt 2>&1; echo "emul rc=$?"; grep "^emulate\|fused\|rror" gpurun_out/${tag}_emul.txt | cut -c1-220
timeout -k 10 700 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_emulation.py tests/test_gpu_random_shapes.py -x -q -m gpu > gpurun_out/${tag}_parity.log 2>&1; rc=$?; echo "parity rc=$rc"; tail -5 gpurun_out/${tag}_parity.log
exit 0
