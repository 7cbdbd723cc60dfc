#!/bin/bash
# Round-5 evidence for profiles/: GPU suite, bench line, kernel stats of the un-overlapped reference, one-location latencies + kernel
# rates + counter traffic, configuration timings, the 2-rank rehearsal.  Usage: gpurun --timeout 1200 -- bash tools/runs/r5_final.sh <tag> suite|prof
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r5z}; what=${2:-prof}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
if [ "$what" = suite ]; then
  timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=12 -p no:cacheprovider > $out/pytest.log 2>&1
  rc=$?; echo "pytest rc=$rc"; tail -4 $out/pytest.log
  exit $rc
fi
timeout -k 10 500 python3 bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$out/ks_fp64" -o ks -- python3 "$R/bench.py" --separate-calls --no-cpu-baseline --no-emulated-line > "$R/$out/bench_separate_calls.json" 2> "$R/$out/ks_fp64.log" ); echo "stats (separate calls) rc=$?"
f=$(find $out/ks_fp64 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/kernel_stats_fp64.csv
find $out/ks_fp64 -name "*kernel_trace.csv" -delete
# one-location calls: wall-clock table, BO iteration, kernel durations, counter traffic of the two streaming kernels
for n in 512 2048 16384; do timeout -k 10 200 python3 tools/gpbench.py rows_trace $n 300; done > $out/rows_trace.txt 2>&1; echo "rows_trace rc=$?"
timeout -k 10 300 python3 tools/gpbench.py bo_iteration_timing > $out/bo_iter.txt 2>&1; echo "bo_iter rc=$?"
timeout -k 10 500 python3 bench.py --small-calls > $out/small_calls.txt 2> $out/small_calls.err; echo "small calls rc=$?"
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$out/ks_rows" -o ks -- python3 "$R/tools/gpbench.py" rows_trace 16384 100 > "$R/$out/ks_rows.log" 2>&1 ); echo "rows stats rc=$?"
f=$(find $out/ks_rows -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "Name|rows_|transpose_tri" "$f" > $out/rows_kernel_stats.csv
find $out/ks_rows -name "*kernel_trace.csv" -delete
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d "$R/$out/rows_$c" -o pmc -- python3 "$R/tools/gpbench.py" rows_trace 16384 20 > "$R/$out/rows_$c.log" 2>&1 )
  echo "rows $c rc=$?"
done
python3 tools/pmc_traffic.py $out/rows_FETCH_SIZE $out/rows_WRITE_SIZE $out/rows_forward_traffic.json --kernel rows_forward_kernel --mode stream > /dev/null; echo "forward traffic rc=$?"
python3 tools/pmc_traffic.py $out/rows_FETCH_SIZE $out/rows_WRITE_SIZE $out/rows_backward_traffic.json --kernel rows_backward_kernel --mode stream > /dev/null; echo "backward traffic rc=$?"
find $out -name "*counter_collection.csv" -delete
timeout -k 10 300 python3 tools/gpbench.py configs_timing > $out/configs.txt 2>&1; cat $out/configs.txt
GPHIP_BENCH_SAME_DEVICE=1 timeout -k 10 300 python3 bench.py --gpus 2 --c4-M 200000 --steps 3 2> $out/bench2.err | grep '^{"metric"' > $out/bench2.json; echo "bench --gpus 2 (same device) rc=$?"
python3 - <<PY
import json
d = json.loads([l for l in open("$out/bench.json") if l.startswith('{"metric"')][0])
r = d["roofline"]
print("bench: %.3f it/s, %.2f ms/step, frac (union) %.3f, per launch %.3f, separate %.3f (%.3f ms), step_frac %.3f, chain %.3f, cpu %.4f" % (
    d["value"], d["ms_per_step"], r["frac"], r["frac_per_launch"], r["separate_calls_reference"]["frac"],
    r["separate_calls_reference"]["avg_launch_ms"], r["step_frac"], d["chain_gemm"]["frac"], d["cpu_baseline"]["value"]))
s = json.loads([l for l in open("$out/bench_separate_calls.json") if l.startswith('{"metric"')][0])
print("separate-calls run under rocprofv3: %.2f ms/step, avg launch %.3f ms (%d launches)" % (s["ms_per_step"], s["roofline"]["avg_launch_ms"], s["roofline"]["launches"]))
d2 = json.loads(open("$out/bench2.json").read())
print("gpus 2 same device: value %.3f scaling %s agree %s winner==N1 %s" % (d2["value"], d2["scaling"], d2["config"]["ranks_agree_on_winner"],
      d2["config"]["best_candidate_global_row"] == d["config"]["best_candidate_global_row"]))
PY
grep -E "gemm_nt_kernel<1, 128, 4" $out/kernel_stats_fp64.csv | head -2
cat $out/rows_kernel_stats.csv
cat $out/rows_trace.txt $out/bo_iter.txt
