#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R="$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2d_kt -o e -- python3 $R/tools/emul_once.py > $R/gpurun_out/r2d.log 2>&1 || { tail -5 $R/gpurun_out/r2d.log; exit 1; }
cd $R && head -12 gpurun_out/r2d_kt/e_kernel_stats.csv | cut -c1-200
find gpurun_out/r2d_kt -name "*kernel_trace.csv" -size +30M -delete
exit 0
