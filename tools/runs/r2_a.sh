#!/bin/bash
# GPU call A of round 2: full GPU test suite, default bench, one kernel-trace profile of the bench (exit-code check)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=25 -p no:cacheprovider > gpurun_out/r2a_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc" | tee gpurun_out/r2a_rc.txt
tail -5 gpurun_out/r2a_pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 400 python bench.py > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err
rc2=$?
echo "bench rc=$rc2" | tee -a gpurun_out/r2a_rc.txt
if [ $rc2 -ne 0 ]; then tail -20 gpurun_out/r2a_bench.err; exit $rc2; fi
cut -c1-600 gpurun_out/r2a_bench.json
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r2a_kt" -o kt -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/r2a_kt.log" 2>&1 )
rc3=$?
echo "rocprofv3 kernel-trace rc=$rc3" | tee -a gpurun_out/r2a_rc.txt
grep -c "Aborted at" gpurun_out/r2a_kt.log | sed 's/^/aborts in profiler log: /' | tee -a gpurun_out/r2a_rc.txt
exit 0
