#!/bin/bash
# Round-4 baseline / evidence: bench line (no CPU baseline), fit timelines of both modes, configuration timings
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r4a}
shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python3 bench.py --no-cpu-baseline "$@" > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
for mode in 0 1; do
  ( cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$out/kt$mode" -o kt -- python3 "$GRAFT_REPO_ROOT/tools/fit_once.py" emulate_fp64=$mode > "$GRAFT_REPO_ROOT/$out/kt$mode.log" 2>&1 )
  python3 tools/trace_panels.py $out/kt$mode 6 2,12,19 > $out/fit_panel_timeline_mode$mode.txt 2>&1
  find $out/kt$mode -name "*kernel_trace.csv" -delete
done
timeout -k 10 300 python3 tools/configs_timing.py > $out/configs.txt 2>&1; cat $out/configs.txt
timeout -k 10 200 python3 tools/emul_fit_timing.py > $out/emul.txt 2>&1; grep "^emulate" $out/emul.txt | cut -c1-180
python3 - <<PY
import json
d=json.load(open("$out/bench.json"))
r=d["roofline"]; e=d.get("emulated_fp64_second_line") or {}
print("ms_per_step", d["ms_per_step"], "value", d["value"], "frac", r["frac"], "step_frac", r["step_frac"], "chol", d["config"]["cholesky_tflops"], "cand", d["config"]["cand_solve_tflops"])
print("chain_gemm", d["chain_gemm"]["frac"], d["chain_gemm"]["launches_per_step"], d["chain_gemm"]["kernel_ms_per_step"])
print("emulated", e.get("ms_per_step"), (e.get("int8_gemm") or {}).get("frac"))
print("phases", d["config"]["phases_ms"])
PY
exit 0
