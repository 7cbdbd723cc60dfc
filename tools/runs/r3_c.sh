#!/bin/bash
# potrf tile kernel v3 stand-alone check + the f3 tests
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r3c}
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in v3; do
  timeout -k 10 60 tools/micro/potrf_check_$v > gpurun_out/${tag}_potrf_$v.txt 2>&1
  echo "potrf_check_$v rc=$?" | tee -a gpurun_out/${tag}_rc.txt
  cat gpurun_out/${tag}_potrf_$v.txt
done
timeout -k 10 600 python -m pytest tests/test_gpu_round3.py -m gpu -q -p no:cacheprovider -k optimize > gpurun_out/${tag}_pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/${tag}_rc.txt
tail -30 gpurun_out/${tag}_pytest.log
exit 0
