#!/bin/bash
# potrf polish check, then the GPU suite + bench (tools/runs/r3_suite.sh)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r3g}
mkdir -p gpurun_out
timeout -k 10 60 tools/micro/potrf_check > gpurun_out/${tag}_potrf.txt 2>&1
rc=$?; echo "potrf_check rc=$rc"; cat gpurun_out/${tag}_potrf.txt
if [ $rc -ne 0 ]; then exit 1; fi
bash tools/runs/r3_suite.sh $tag
