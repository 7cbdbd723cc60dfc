#!/bin/bash
# potrf tile kernel: stand-alone check of both versions, then the whole GPU suite (not -x)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r3b}
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in v1 v2; do
  timeout -k 10 60 tools/micro/potrf_check_$v > gpurun_out/${tag}_potrf_$v.txt 2>&1
  echo "potrf_check_$v rc=$?" | tee -a gpurun_out/${tag}_rc.txt
  cat gpurun_out/${tag}_potrf_$v.txt
done
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=40 -p no:cacheprovider > gpurun_out/${tag}_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a gpurun_out/${tag}_rc.txt
tail -60 gpurun_out/${tag}_pytest.log
exit 0
