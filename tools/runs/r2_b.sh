#!/bin/bash
# GPU call B of round 2: counter evidence.  FETCH_SIZE calibration, HBM traffic of the dominant GEMM and the K-builds
# (FETCH_SIZE / WRITE_SIZE in separate passes), SQ / GRBM counters of the dominant GEMM in situ, super-tile order A/B,
# the CPU baseline at the quoted size.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/r2b_calib -o c -- $R/tools/micro/fetch_calib > $O/r2b_calib.log 2>&1 || { echo calib failed; tail -5 $O/r2b_calib.log; exit 1; }
echo calib done
for variant in base st8; do
  opt=""; [ $variant = st8 ] && opt="--option supertile=8"
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r2b_${variant}_fetch -o f -- $B $opt > $O/r2b_${variant}_fetch.log 2>&1 || { echo "$variant fetch failed"; tail -5 $O/r2b_${variant}_fetch.log; exit 1; }
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/r2b_${variant}_write -o w -- $B $opt > $O/r2b_${variant}_write.log 2>&1 || { echo "$variant write failed"; exit 1; }
  echo "$variant traffic passes done"
done
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/r2b_sq -o s -- $B > $O/r2b_sq.log 2>&1 || { echo "sq failed"; tail -5 $O/r2b_sq.log; exit 1; }
echo sq done
cd "$R"
# A/B of the super-tile order without a profiler, one process each, interleaved
for i in 1 2; do
  timeout -k 10 200 python bench.py --no-cpu-baseline > $O/r2b_time_base_$i.json 2>/dev/null || exit 1
  timeout -k 10 200 python bench.py --no-cpu-baseline --option supertile=8 > $O/r2b_time_st8_$i.json 2>/dev/null || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2b_time_*.json")):
    r=json.load(open(f)); print(f, round(r["ms_per_step"],2), round(r["roofline"]["frac"],3), round(r["roofline"]["avg_launch_ms"],3))
PY
timeout -k 10 900 python bench.py --steps 3 --warmup 1 --cpu-baseline-full > $O/r2b_cpu_full.json 2> $O/r2b_cpu_full.err || { echo "cpu full failed"; tail -5 $O/r2b_cpu_full.err; exit 1; }
echo cpu full done
# drop the bulky per-dispatch traces that are not needed back home (keep counter_collection and stats)
find $O -name "*kernel_trace.csv" -size +20M -delete
du -sh $O
exit 0
