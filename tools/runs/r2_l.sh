#!/bin/bash
# GPU call L: the whole GPU suite (true fp64), the whole suite once more with GPHIP_EMULATE_FP64=1, the default bench
# line, and the kernel stats of the bench under rocprofv3
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
R="$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=8 -p no:cacheprovider > gpurun_out/r2l_pytest.log 2>&1
rc=$?; echo "pytest (fp64) rc=$rc" | tee gpurun_out/r2l_rc.txt; tail -4 gpurun_out/r2l_pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
GPHIP_EMULATE_FP64=1 timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_emulation.py > gpurun_out/r2l_pytest_emulated.log 2>&1
rc=$?; echo "pytest (GPHIP_EMULATE_FP64=1) rc=$rc" | tee -a gpurun_out/r2l_rc.txt; tail -12 gpurun_out/r2l_pytest_emulated.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 400 python bench.py > gpurun_out/r2l_bench.json 2> gpurun_out/r2l_bench.err || { tail -20 gpurun_out/r2l_bench.err; exit 1; }
python -c "
import json; r=json.load(open('gpurun_out/r2l_bench.json')); print('bench', r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['traffic'], r['emulated_fp64_second_line'], r['cpu_baseline']['value'])"
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2l_kt -o kt -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/r2l_kt.log 2>&1 ); echo "rocprofv3 rc=$?" | tee -a gpurun_out/r2l_rc.txt
grep -c "Aborted at" gpurun_out/r2l_kt.log | sed 's/^/aborts: /'
find gpurun_out/r2l_kt -name "*kernel_trace.csv" -size +30M -delete
exit 0
