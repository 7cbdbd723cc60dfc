#!/bin/bash
# round 5, VERDICT item 5: gp_fit_predict (C3) against the panel at which the candidate stages are released behind the factorisation
# (pipe_start_pct; 22 panels at N = 16384: 32 % = panel 7 (default), 55 % = panel 12 where the chain is the critical path again) x the
# number of stages released (pipe_stages) x what the bulk stream keeps of the owned-column rule while they run (own_keep_pipe_pct)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5rel
for pct in -1 40 50 55 60 70; do
  for st in 0 2 3; do
    timeout -k 10 120 python3 tools/gpbench.py fused_sweep pipe_start_pct=$pct pipe_stages=$st 2>&1 | sed "s/^/start_pct=$pct stages=$st /" | cut -c1-110
  done
done | tee gpurun_out/r5rel/sweep.txt
for keep in 0 50 100; do
  timeout -k 10 120 python3 tools/gpbench.py fused_sweep pipe_start_pct=55 own_keep_pipe_pct=$keep 2>&1 | sed "s/^/start_pct=55 own_keep_pipe_pct=$keep /" | cut -c1-120
done | tee -a gpurun_out/r5rel/sweep.txt
