#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 800 python tools/emul_sweep.py 300 > gpurun_out/r2t.log 2>&1 || { tail -20 gpurun_out/r2t.log; exit 1; }
head -60 gpurun_out/r2t.log
