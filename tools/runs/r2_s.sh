#!/bin/bash
# the other BASELINE.json configurations (C2, one C4 shard, C5) in true fp64 and with GPHIP_EMULATE_FP64=1
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
echo "== true fp64" > gpurun_out/r2s.log
timeout -k 10 400 python tools/configs_timing.py >> gpurun_out/r2s.log 2>&1 || { tail -5 gpurun_out/r2s.log; exit 1; }
echo "== GPHIP_EMULATE_FP64=1" >> gpurun_out/r2s.log
GPHIP_EMULATE_FP64=1 timeout -k 10 400 python tools/configs_timing.py >> gpurun_out/r2s.log 2>&1 || { tail -5 gpurun_out/r2s.log; exit 1; }
cat gpurun_out/r2s.log
