#!/bin/bash
# clocks and power of the card while one phase of the bench step runs in a loop (rocm-smi sampled beside it)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r3clk}
out=gpurun_out/$tag.txt
mkdir -p gpurun_out
: > $out
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -v "^$" | head -40 >> $out
for what in predict fit fused emulated; do
  echo "=== $what" >> $out
  timeout -k 10 120 python3 tools/load_loop.py $what 6 >> $out 2>&1 &
  pid=$!
  sleep 3
  for i in 1 2 3 4 5 6; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" | tr -s ' ' | tr '\n' ';' >> $out; echo >> $out
    sleep 0.5
  done
  wait $pid
done
cat $out | cut -c1-300
exit 0
