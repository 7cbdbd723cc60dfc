#!/bin/bash
# kernel-trace timeline of one gp_fit at the headline size: r4_trace.sh <tag> <panels> [opt=val ...]
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; panels=$2; shift 2
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
( cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$out/kt" -o kt -- python3 "$GRAFT_REPO_ROOT/tools/fit_once.py" "$@" > "$GRAFT_REPO_ROOT/$out/kt.log" 2>&1 )
python3 tools/trace_panels.py $out/kt 6 $panels > $out/timeline.txt 2>&1
find $out/kt -name "*kernel_trace.csv" -delete
head -30 $out/timeline.txt
