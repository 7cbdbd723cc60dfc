#!/bin/bash
# kernel trace of the fused step (true fp64): where no long launch is running
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r3k}
mkdir -p gpurun_out
export TMPDIR=/tmp
( cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${tag}_kt" -o kt -- python3 "$GRAFT_REPO_ROOT/tools/fused_once.py" > "$GRAFT_REPO_ROOT/gpurun_out/${tag}_kt.log" 2>&1 )
echo "trace rc=$?"
python3 tools/trace_gaps.py gpurun_out/${tag}_kt 1000 40 > gpurun_out/${tag}_gaps.txt 2>&1
python3 tools/trace_fused.py gpurun_out/${tag}_kt > gpurun_out/${tag}_fused.txt 2>&1
find gpurun_out/${tag}_kt -name "*kernel_trace.csv" -size +20M -delete
exit 0
