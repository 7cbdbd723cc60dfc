#!/bin/bash
# kernel trace of the emulated fused step (gp_fit_predict, emulate_fp64 = 1)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r3o}
mkdir -p gpurun_out
export TMPDIR=/tmp
( cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${tag}_kt" -o kt -- python3 "$GRAFT_REPO_ROOT/tools/fused_once.py" emulate_fp64=1 "${@:2}" > "$GRAFT_REPO_ROOT/gpurun_out/${tag}_kt.log" 2>&1 )
echo "trace rc=$?"
python3 tools/trace_list.py gpurun_out/${tag}_kt kbuild 100 > gpurun_out/${tag}_list.txt 2>&1
find gpurun_out/${tag}_kt -name "*kernel_trace.csv" -size +20M -delete
exit 0
