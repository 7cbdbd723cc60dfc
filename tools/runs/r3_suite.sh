#!/bin/bash
# Round-3 GPU call: the GPU test suite (both arithmetic modes inside it), the default bench, the 2-rank rehearsal of the
# self-launching bench on one device.  Usage: gpurun -- bash tools/runs/r3_suite.sh <tag>
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r3a}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --durations=30 -p no:cacheprovider > gpurun_out/${tag}_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc" | tee gpurun_out/${tag}_rc.txt
tail -8 gpurun_out/${tag}_pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 500 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
rc2=$?
echo "bench rc=$rc2" | tee -a gpurun_out/${tag}_rc.txt
if [ $rc2 -ne 0 ]; then tail -20 gpurun_out/${tag}_bench.err; exit $rc2; fi
cut -c1-400 gpurun_out/${tag}_bench.json
GPHIP_BENCH_SAME_DEVICE=1 timeout -k 10 300 python3 bench.py --gpus 2 --M 200000 --steps 2 > gpurun_out/${tag}_bench2.json 2> gpurun_out/${tag}_bench2.err
rc3=$?
echo "bench --gpus 2 (same device) rc=$rc3" | tee -a gpurun_out/${tag}_rc.txt
tail -3 gpurun_out/${tag}_bench2.err
cut -c1-300 gpurun_out/${tag}_bench2.json
exit 0
