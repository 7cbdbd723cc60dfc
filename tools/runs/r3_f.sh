#!/bin/bash
# re-tune panel width / reserved CUs after the faster diagonal-tile kernel
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r3f}
mkdir -p gpurun_out
export TMPDIR=/tmp
for rc in 16 24 32 48; do for pt in 4 5 6 8; do
  GPHIP_RESERVE_CUS=$rc timeout -k 10 120 python3 tools/fused_sweep.py panel_tiles=$pt 2>&1 | tail -1 | sed "s/^/pt=$pt /"
done; done | tee gpurun_out/${tag}_sweep.txt
exit 0
