#!/bin/bash
# measurements after the new diagonal-tile kernel: configuration timings, fit timelines (fp64 + emulated), fused sweep
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r3e}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 python3 tools/configs_timing.py > gpurun_out/${tag}_configs.txt 2>&1; echo "configs rc=$?"
cat gpurun_out/${tag}_configs.txt
for mode in 0 1; do
  ( cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${tag}_kt$mode" -o kt -- python3 "$GRAFT_REPO_ROOT/tools/fit_once.py" emulate_fp64=$mode > "$GRAFT_REPO_ROOT/gpurun_out/${tag}_kt$mode.log" 2>&1 )
  echo "trace mode $mode rc=$?"
  python3 tools/trace_panels.py gpurun_out/${tag}_kt$mode 6 2,12,19 > gpurun_out/${tag}_panels$mode.txt 2>&1
  find gpurun_out/${tag}_kt$mode -name "*kernel_trace.csv" -size +20M -delete
done
head -40 gpurun_out/${tag}_panels0.txt
for st in 1 2 3 4 5; do for pct in 30 40 55; do
  timeout -k 10 120 python3 tools/fused_sweep.py pipe_stages=$st pipe_start_pct=$pct 2>&1 | tail -1 | sed "s/^/stages=$st pct=$pct /"
done; done | tee gpurun_out/${tag}_fused_sweep.txt
timeout -k 10 200 python3 tools/emul_fit_timing.py > gpurun_out/${tag}_emul.txt 2>&1; tail -12 gpurun_out/${tag}_emul.txt
exit 0
