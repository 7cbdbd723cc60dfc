#!/bin/bash
# Round-3 evidence for profiles/: bench line, kernel stats per mode, fit timeline, tile-kernel check
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r3p}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 60 tools/micro/potrf_check > $out/potrf_check.txt 2>&1; echo "potrf_check rc=$?"
timeout -k 10 500 python3 bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/ks_fp64" -o ks -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-emulated-line > "$GRAFT_REPO_ROOT/$out/ks_fp64.log" 2>&1 ); echo "stats fp64 rc=$?"
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/ks_emul" -o ks -- python3 "$GRAFT_REPO_ROOT/tools/emul_bench.py" > "$GRAFT_REPO_ROOT/$out/ks_emul.log" 2>&1 ); echo "stats emulated rc=$?"
for m in fp64 emul; do
  f=$(find $out/ks_$m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/kernel_stats_$m.csv
  find $out/ks_$m -name "*kernel_trace.csv" -delete
done
for mode in 0 1; do
  ( cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$out/kt$mode" -o kt -- python3 "$GRAFT_REPO_ROOT/tools/fit_once.py" emulate_fp64=$mode > "$GRAFT_REPO_ROOT/$out/kt$mode.log" 2>&1 )
  python3 tools/trace_panels.py $out/kt$mode 6 2,12,19 > $out/fit_panel_timeline_mode$mode.txt 2>&1
  find $out/kt$mode -name "*kernel_trace.csv" -delete
done
timeout -k 10 300 python3 tools/configs_timing.py > $out/configs.txt 2>&1; cat $out/configs.txt
timeout -k 10 200 python3 tools/emul_fit_timing.py > $out/emul.txt 2>&1; grep "^emulate" $out/emul.txt | cut -c1-180
head -c 300 $out/bench.json; echo
grep -h "potrf\|gemm_nt_kernel<1, 128, 4\|gemm_nt_kernel<1, 64, 2, false, 64\|rns_gemm" $out/kernel_stats_fp64.csv $out/kernel_stats_emul.csv | cut -c1-200
exit 0
