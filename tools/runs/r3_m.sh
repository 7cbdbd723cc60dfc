#!/bin/bash
# kernel trace of the emulated candidate solve (one predict at C3)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r3m}
mkdir -p gpurun_out
export TMPDIR=/tmp
( cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${tag}_kt" -o kt -- python3 "$GRAFT_REPO_ROOT/tools/emul_once.py" > "$GRAFT_REPO_ROOT/gpurun_out/${tag}_kt.log" 2>&1 )
echo "trace rc=$?"
python3 tools/trace_list.py gpurun_out/${tag}_kt cross_k > gpurun_out/${tag}_list.txt 2>&1
find gpurun_out/${tag}_kt -name "*kernel_trace.csv" -size +20M -delete
exit 0
