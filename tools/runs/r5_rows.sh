#!/bin/bash
# one-row call latencies after the fused path (csrc/onerow.hip): table, BO iteration, kernel trace
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r5rows
mkdir -p $out
timeout -k 10 300 python3 tools/gpbench.py rows_trace 512 400 > $out/trace.txt 2>&1 &&
timeout -k 10 300 python3 tools/gpbench.py rows_trace 2048 400 >> $out/trace.txt 2>&1 &&
timeout -k 10 300 python3 tools/gpbench.py rows_trace 16384 200 >> $out/trace.txt 2>&1 &&
timeout -k 10 300 python3 tools/gpbench.py bo_iteration_timing > $out/bo_iter.txt 2>&1 &&
timeout -k 10 500 python3 bench.py --small-calls > $out/small_calls.txt 2> $out/small_calls.err &&
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof -o rows -- python3 $GRAFT_REPO_ROOT/tools/gpbench.py rows_trace 16384 50 > $GRAFT_REPO_ROOT/$out/prof.log 2>&1)
cat $out/trace.txt $out/bo_iter.txt
tail -5 $out/small_calls.txt
exit 0
