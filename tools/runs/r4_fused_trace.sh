#!/bin/bash
# kernel-trace timeline of one gp_fit_predict at the headline size + a plain bench line: r4_fused_trace.sh <tag>
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r4h}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 500 python3 bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
( cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$out/kt" -o kt -- python3 "$GRAFT_REPO_ROOT/tools/fused_once.py" > "$GRAFT_REPO_ROOT/$out/kt.log" 2>&1 )
python3 tools/trace_fused.py $out/kt > $out/fused_timeline.txt 2>&1
find $out/kt -name "*kernel_trace.csv" -delete
head -40 $out/fused_timeline.txt
python3 - <<PY
import json
d=json.load(open("$out/bench.json")); r=d["roofline"]
print("ms_per_step", d["ms_per_step"], "value", d["value"], "frac", r["frac"], "while_running", r["frac_while_running"], "step_frac", r["step_frac"], "traffic", r["traffic"])
print(d["config"]["phases_ms"], d["config"]["cholesky_tflops"], d["config"]["cand_solve_tflops"])
PY
