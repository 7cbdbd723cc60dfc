#!/bin/bash
# two-rank rehearsals of the multi-rank bench on the one-GPU box (both ranks on device 0): self-launched and under torch.distributed.run
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r4i}
out=gpurun_out/$tag
mkdir -p $out
GPHIP_BENCH_SAME_DEVICE=1 timeout -k 10 400 python3 bench.py --gpus 2 --c4-M 200000 --steps 3 2> $out/bench2.err > $out/bench2.out; echo "self-launched rc=$?"
grep -c . $out/bench2.out
GPHIP_BENCH_SAME_DEVICE=1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29731 bench.py --gpus 2 --c4-M 200000 --steps 3 2> $out/bench2_torchrun.err > $out/bench2_torchrun.out; echo "torchrun rc=$?"
grep -c . $out/bench2_torchrun.out
timeout -k 10 300 python -m pytest tests/test_gpu_round3.py -q -m gpu -k "bench" -p no:cacheprovider 2>&1 | tail -3
python3 - <<PY
import json
for f in ("$out/bench2.out", "$out/bench2_torchrun.out"):
    d = json.loads([l for l in open(f) if l.startswith('{"metric"')][0])
    c = d["config"]; c4 = d["c4_sharded"]
    print(f, "value", round(d["value"], 3), "ms_per_step", round(d["ms_per_step"], 2), "scaling", d["scaling"], "agree", c["ranks_agree_on_winner"], "collective", c["collective"][:20], "torch", c["torch_imported"], "launcher", c["launcher"][:24])
    print("   c4:", round(c4["ms_per_iter"], 1), "ms/iter", "speedup", round(c4["speedup_vs_single_gpu"], 2), "match", c4["best_row_matches_single_gpu"], "agree", c4["ranks_agree_on_winner"], c4["candidates_total"], c4["candidates_this_rank"])
PY
