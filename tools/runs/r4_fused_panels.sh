#!/bin/bash
# per-panel chain timeline of one gp_fit_predict (kernel trace): r4_fused_panels.sh <tag> <panels>
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; panels=$2
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
( cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$out/kt" -o kt -- python3 "$GRAFT_REPO_ROOT/tools/fused_once.py" > "$GRAFT_REPO_ROOT/$out/kt.log" 2>&1 )
python3 tools/trace_panels.py $out/kt 6 $panels > $out/fused_panels.txt 2>&1
find $out/kt -name "*kernel_trace.csv" -delete
head -28 $out/fused_panels.txt | cut -c1-200
