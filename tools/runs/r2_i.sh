#!/bin/bash
# rehearsal of the multi-GPU bench path on one GPU: C4 on one rank, then two ranks on the same device (RCCL refuses the
# duplicate device, the exchange falls back to gloo -- the code path around it is what is being rehearsed)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --workload c4 --M 200000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2i_c4_1.json 2> gpurun_out/r2i_c4_1.err || { tail -20 gpurun_out/r2i_c4_1.err; exit 1; }
python -c "
import json; r=json.load(open('gpurun_out/r2i_c4_1.json')); print('c4 x1:', r['value'], r['ms_per_step'], r['scaling'], r['config']['fit_ms_per_iter_not_scaling'], r['config']['predict_ei_ms_per_iter_this_rank'], r['roofline']['frac'])"
GPHIP_BENCH_SAME_DEVICE=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --M 200000 > gpurun_out/r2i_c4_2.json 2> gpurun_out/r2i_c4_2.err || { tail -30 gpurun_out/r2i_c4_2.err; exit 1; }
tail -3 gpurun_out/r2i_c4_2.err
python -c "
import json
for l in open('gpurun_out/r2i_c4_2.json'):
    if l.startswith('{'):
        r=json.loads(l); c=r['config']; print('c4 x2:', r['value'], r['ms_per_step'], r['n_gpus'], c['collective'], c['rccl_comm_ranks'], c['single_gpu_reference'], c['speedup_vs_single_gpu'], c['best_row_matches_single_gpu'], c['candidates_this_rank'])"
