#!/bin/bash
# gp_fit_predict against the number of CUs kept free of the candidate stream (GPHIP_PRED_RESERVE: fixed per process) x pipelined stages
cd "$GRAFT_REPO_ROOT" || exit 1
for r in 32 64 96 128; do
  for st in 0 5 8; do
    GPHIP_PRED_RESERVE=$r timeout -k 10 120 python3 tools/fused_sweep.py pipe_stages=$st 2>&1 | sed "s/^/pred_reserve=$r stages=$st /" | cut -c1-120
  done
done
