#!/bin/bash
# round-5 bench checks: the default line (N = 1), the two-rank same-device rehearsals (strong scaling of the quoted workload)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r5bench}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 500 python3 bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
GPHIP_BENCH_SAME_DEVICE=1 timeout -k 10 400 python3 bench.py --gpus 2 --c4-M 200000 --steps 3 2> $out/bench2.err > $out/bench2.out; echo "self-launched rc=$?"
GPHIP_BENCH_SAME_DEVICE=1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29731 bench.py --gpus 2 --c4-M 200000 --steps 3 2> $out/bench2_torchrun.err > $out/bench2_torchrun.out; echo "torchrun rc=$?"
python3 - <<PY
import json
d1 = json.loads([l for l in open("$out/bench.json") if l.startswith('{"metric"')][0])
r = d1["roofline"]
print("N=1: value %.3f it/s, ms_per_step %.2f, scaling %s, roofline frac (union) %.3f, per launch %.3f, separate calls %.3f (avg launch %.3f ms), step_frac %.3f, chain %.3f" % (
    d1["value"], d1["ms_per_step"], d1["scaling"], r["frac"], r["frac_per_launch"], r["separate_calls_reference"]["frac"],
    r["separate_calls_reference"]["avg_launch_ms"], r["step_frac"], d1["chain_gemm"]["frac"]))
print("   winner", d1["config"]["best_candidate_global_row"], d1["config"]["best_value"], "cpu", d1.get("cpu_baseline", {}).get("value"))
for f in ("$out/bench2.out", "$out/bench2_torchrun.out"):
    d = json.loads([l for l in open(f) if l.startswith('{"metric"')][0])
    c = d["config"]; c4 = d["c4_sharded"]
    print(f, "value", round(d["value"], 3), "ms_per_step", round(d["ms_per_step"], 2), "scaling", d["scaling"], "agree", c["ranks_agree_on_winner"], "collective", c["collective"][:20], "rccl_comm_ranks", c["rccl_comm_ranks"], "torch", c["torch_imported"], "launcher", c["launcher"][:24])
    print("   winner == N=1 winner:", c["best_candidate_global_row"] == d1["config"]["best_candidate_global_row"] and c["best_value"] == d1["config"]["best_value"], c["candidates_total"], c["candidates_this_rank"])
    print("   c4:", round(c4["ms_per_iter"], 1), "ms/iter", "speedup", round(c4["speedup_vs_single_gpu"], 2), "match", c4["best_row_matches_single_gpu"], "agree", c4["ranks_agree_on_winner"], c4["candidates_total"], c4["candidates_this_rank"])
PY
