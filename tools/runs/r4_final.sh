#!/bin/bash
# Round-4 evidence for profiles/: GPU suite, bench line, per-mode kernel stats, counter passes, fit timelines, configuration timings,
# the 2-rank rehearsal.  Usage: gpurun --timeout 1200 -- bash tools/runs/r4_final.sh <tag> [suite|prof|all]
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r4z}; what=${2:-all}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
if [ "$what" = suite ] || [ "$what" = all ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=15 -p no:cacheprovider > $out/pytest.log 2>&1
  rc=$?; echo "pytest rc=$rc"; tail -4 $out/pytest.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
fi
if [ "$what" = prof ] || [ "$what" = all ]; then
  timeout -k 10 500 python3 bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
  ( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/ks_fp64" -o ks -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-emulated-line > "$GRAFT_REPO_ROOT/$out/ks_fp64.log" 2>&1 ); echo "stats fp64 rc=$?"
  ( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/ks_emul" -o ks -- python3 "$GRAFT_REPO_ROOT/tools/emul_bench.py" > "$GRAFT_REPO_ROOT/$out/ks_emul.log" 2>&1 ); echo "stats emulated rc=$?"
  for m in fp64 emul; do
    f=$(find $out/ks_$m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/kernel_stats_$m.csv
    find $out/ks_$m -name "*kernel_trace.csv" -delete
  done
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d "$GRAFT_REPO_ROOT/$out/$c" -o pmc -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-emulated-line > "$GRAFT_REPO_ROOT/$out/$c.log" 2>&1 )
    echo "$c rc=$?"
  done
  python3 tools/pmc_traffic.py $out/FETCH_SIZE $out/WRITE_SIZE $out/kbuild_traffic.json --kernel kbuild_kernel --mode stream > /dev/null; echo "kbuild rc=$?"
  python3 tools/pmc_traffic.py $out/FETCH_SIZE $out/WRITE_SIZE $out/cross_k_traffic.json --kernel cross_k_kernel --mode stream > /dev/null; echo "cross_k rc=$?"
  python3 tools/pmc_traffic.py $out/FETCH_SIZE $out/WRITE_SIZE $out/gemm_traffic.json --min-wgs 1400 > /dev/null; echo "gemm rc=$?"
  find $out -name "*counter_collection.csv" -delete
  for mode in 0 1; do
    ( cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$out/kt$mode" -o kt -- python3 "$GRAFT_REPO_ROOT/tools/fit_once.py" emulate_fp64=$mode > "$GRAFT_REPO_ROOT/$out/kt$mode.log" 2>&1 )
    python3 tools/trace_panels.py $out/kt$mode 6 2,12,19 > $out/fit_panel_timeline_mode$mode.txt 2>&1
    find $out/kt$mode -name "*kernel_trace.csv" -delete
  done
  timeout -k 10 300 python3 tools/configs_timing.py > $out/configs.txt 2>&1; cat $out/configs.txt
  timeout -k 10 200 python3 tools/emul_fit_timing.py > $out/emul.txt 2>&1; grep "^emulate" $out/emul.txt | cut -c1-180
  GPHIP_BENCH_SAME_DEVICE=1 timeout -k 10 300 python3 bench.py --gpus 2 --c4-M 200000 --steps 2 2> $out/bench2.err | grep '^{"metric"' > $out/bench2.json; echo "bench --gpus 2 (same device) rc=$?"
  python3 - <<PY
import json
d=json.load(open("$out/bench.json"))
r=d["roofline"]; e=d.get("emulated_fp64_second_line") or {}
print("ms_per_step", d["ms_per_step"], "value", d["value"], "frac", r["frac"], "step_frac", r["step_frac"], "traffic", r["traffic"])
print("chol", d["config"]["cholesky_tflops"], "cand", d["config"]["cand_solve_tflops"], d["config"].get("cand_solve_executed_tflops"))
print("chain_gemm", d["chain_gemm"]["frac"], "emulated", e.get("ms_per_step"), (e.get("int8_gemm") or {}).get("frac"), "cpu", d.get("cpu_baseline", {}).get("value"))
PY
fi
exit 0
