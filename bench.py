#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json on MI355X: GP fit+predict iterations/s at N=16384, D=8.

A "step" is one pass of the hot path over one batch of synthetic input:
    fit      (K build -> Cholesky -> alpha -> LML)                     SURVEY.md 8a rows A1-A5
    predict  on the rank's 10 000 resident candidates                   rows A6-A7
             (both through ONE call, gp_fit_predict; --separate-calls times gp_fit + gp_predict instead)
    EI scoring of those candidates + device arg-best                    rows A9-A11, A14
    (N > 1) one RCCL all-gather of the per-shard (best, index) pair     SURVEY.md 8e
Inputs are resident in HBM before the timed region; hyper-parameters are fixed.

Multi-GPU (one process per GPU, launched by torch.distributed.run): the candidate table shards
across ranks, the fit is REPLICATED on every rank ("replicas only" for the fit), so
per-GPU work is fixed as N grows ("weak").  `value` counts the fit+predict units all ranks
processed per second; config.job_iters_per_s is the rate of whole sharded iterations.
torch is imported only for N > 1 (rendezvous + barrier over gloo); the data path is libgphip
(ctypes) and its RCCL communicator.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X vendor dense FP64 matrix peak (v_mfma_f64_16x16x4_f64); see DESIGN.md


def synthetic(N, D, M, seed=1234, cand_seed=None):
    """SURVEY.md 8(d) generators (same as oracle.cpu_ref.synthetic_problem, restated so that the timed
    product path never touches oracle/)."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(0, 1, (N, D))
    f = np.sin(2 * np.pi * X).sum(1, keepdims=True) / np.sqrt(D)
    Y = f + 0.05 * np.random.default_rng(seed + 1).standard_normal((N, 1))
    Y = (Y - Y.mean()) / Y.std()
    Xs = np.random.default_rng(seed + 2 if cand_seed is None else cand_seed).uniform(0, 1, (M, D))
    return X, Y, Xs


def cpu_baseline(N, D, M):
    """The oracle (NumPy/SciPy restatement of the reference's path) timed on this host, on a bounded
    sample: N/2, M/2 of the same workload = 1/8 of its flops; the time is scaled by that ratio."""
    from oracle import cpu_ref as O
    try:
        from threadpoolctl import threadpool_info
        blas = [(d.get("internal_api"), d.get("num_threads")) for d in threadpool_info()]
        cores = max([n for _, n in blas if n] or [os.cpu_count()])
    except Exception:  # noqa: BLE001
        blas, cores = [], os.cpu_count()
    Ns, Ms = N // 2, M // 2
    X, Y, Xs = O.synthetic_problem(Ns, D, Ms, seed=1234)
    kern = O.RBF(D, 1.0, O.default_lengthscale(D, False))
    t0 = time.perf_counter()
    lml, mu, var, phases = O.fit_predict_iteration(kern, X, Y, 1e-2, Xs, as_gpy=False)
    t = time.perf_counter() - t0
    flops_full = N ** 3 / 3.0 + float(N) * N * M
    flops_s = Ns ** 3 / 3.0 + float(Ns) * Ns * Ms
    scale = flops_full / flops_s
    return {"value": 1.0 / (t * scale), "unit": "fit+predict iters/s", "cores": int(cores), "kind": "port",
            "sample": "N=%d, M=%d (1/%.0f of the flops of N=%d, M=%d), minimal path (K, dpotrf, dpotrs, K*, dtrtrs); "
                      "measured %.2f s, scaled by %.1f" % (Ns, Ms, scale, N, M, t, scale),
            "phases_s": {k: round(v, 3) for k, v in phases.items()}, "blas": blas}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--N", type=int, default=16384)
    ap.add_argument("--D", type=int, default=8)
    ap.add_argument("--M", type=int, default=10000, help="candidates per GPU")
    ap.add_argument("--kernel", default="rbf", choices=["rbf", "Mat52"])
    ap.add_argument("--panel-tiles", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--separate-calls", action="store_true",
                    help="time gp_fit + gp_predict as two calls instead of the one-call entry point gp_fit_predict")
    ap.add_argument("--extras", action="store_true", help="also time the other of the two call patterns (un-timed region)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    dist = None
    if world > 1:
        import torch.distributed as dist  # plumbing only: rendezvous, barrier, max-over-ranks
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    from gaussian_process_optimization_amd import _lib
    # rehearsal switch for a one-GPU box: every rank on device 0 (RCCL then refuses the duplicate device and the
    # exchange falls back to gloo); never set by the driver
    dev = 0 if os.environ.get("GPHIP_BENCH_SAME_DEVICE") else local_rank
    h = _lib.Handle(dev)
    if args.panel_tiles:
        h.set_option("panel_tiles", args.panel_tiles)
    N, D, M = args.N, args.D, args.M
    X, Y, Xs = synthetic(N, D, M, cand_seed=1236 + rank)   # every rank: same model data, its own candidate shard
    kid = _lib.GP_KERNEL_RBF if args.kernel == "rbf" else _lib.GP_KERNEL_MATERN52
    h.set_data(X, Y)
    h.set_params(kid, 0, 1.0, [0.25 * np.sqrt(D)], 1e-2)
    h.set_candidates(Xs)
    collective = None
    if world > 1:
        import torch
        # the one data-path collective: an all-gather of 16 bytes per rank over RCCL.  If the RCCL
        # communicator cannot be built on this node every rank agrees to fall back to the rendezvous backend (gloo)
        # for that exchange -- said in the JSON line -- so that the scaling run still measures the sharded path.
        ok = 1
        try:
            uid = [h.comm_unique_id() if rank == 0 else None]
        except Exception as e:  # noqa: BLE001
            uid, ok = [None], 0
            sys.stderr.write("rank %d: RCCL unique id failed: %s\n" % (rank, e))
        dist.broadcast_object_list(uid, src=0)
        if uid[0] is None:
            ok = 0
        if ok:
            try:
                h.comm_init(uid[0], rank, world)
            except Exception as e:  # noqa: BLE001
                ok = 0
                sys.stderr.write("rank %d: RCCL comm init failed: %s\n" % (rank, e))
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        collective = "rccl" if int(flag.item()) == 1 else "gloo (RCCL communicator unavailable)"

    def barrier():
        h.synchronize()
        if dist is not None:
            dist.barrier()
        h.synchronize()

    def step(pipelined=False):
        if pipelined:  # gp_fit + gp_predict as one pipelined pass (same results; see include/gphip.h)
            (lml, logdet, jit), mu, var = h.fit_predict(True)
        else:
            lml, logdet, jit = h.fit()
            mu, var = h.predict(True)
        fmin = h.fmin()
        idx, val = h.acq_argbest(_lib.GP_ACQ_EI, 0.01, fmin, -1)
        if world > 1:
            from gaussian_process_optimization_amd.sharded import merge_best
            if collective == "rccl":
                vals, idxs = h.comm_allgather_best(val, rank * M + idx, world)
            else:
                import torch
                mine = torch.tensor([val, float(rank * M + idx)], dtype=torch.float64)
                allp = [torch.empty(2, dtype=torch.float64) for _ in range(world)]
                dist.all_gather(allp, mine)
                vals = np.array([float(t[0]) for t in allp])
                idxs = np.array([int(t[1]) for t in allp], dtype=np.int64)
            idx, val = merge_best(vals, idxs, -1)
        return lml, idx, val

    # the timed step goes through gp_fit_predict (fit + predict as one call: bitwise the results of the two calls,
    # tests/test_gpu_parity.py; the first candidate stages ride behind the factorisation's latency-bound tail)
    fused = not args.separate_calls
    for _ in range(args.warmup):
        out = step(fused)
    h.profile(True)          # HIP events around every launch of the dominant kernel (fp64 MFMA GEMM)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(fused)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    gs = h.gemm_stats()
    busy_ms = h.gemm_busy()
    phases_timed = {p["name"]: round(p["ms"], 3) for p in h.phases()}   # of the last call of the last timed step
    h.profile(False)
    if dist is not None:
        import torch
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # un-timed extras: the same step through the pipelined entry point (gp_fit_predict), and phase breakdowns
    pipelined_ms, phases_pipelined = None, None
    other_roofline = None
    if args.extras:
        step(not fused)
        h.profile(True)
        h.synchronize()
        tp0 = time.perf_counter()
        for _ in range(3):
            step(not fused)
        h.synchronize()
        pipelined_ms = (time.perf_counter() - tp0) / 3 * 1e3
        phases_pipelined = {p["name"]: round(p["ms"], 3) for p in h.phases()}
        gs2 = h.gemm_stats()
        h.profile(False)
        a2 = gs2["flops"] / max(gs2["ms"], 1e-9) / 1e9
        other_roofline = {"achieved": a2, "frac": a2 / FP64_MFMA_PEAK_TFLOPS, "launches": gs2["launches"],
                          "avg_launch_ms": gs2["ms"] / max(gs2["launches"], 1)}
    h.fit()
    ph_fit = h.phases()
    h.predict(True)
    ph_pred = h.phases()
    phases = {p["name"]: round(p["ms"], 3) for p in ph_fit + ph_pred}
    chol = [p for p in ph_fit if p["name"] == "cholesky"][0]
    solve = [p for p in ph_pred if p["name"] == "cand_solve"][0]

    # HBM-side traffic of the dominant kernel: PMC passes of this same command (FETCH_SIZE and WRITE_SIZE in separate
    # rocprofv3 --pmc runs, gfx950 correction applied; tools/pmc_traffic.py), averaged per big launch like `achieved`
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "r01_gemm_traffic.json")
    if os.path.exists(tpath) and (N, D, M) == (16384, 8, 10000):
        with open(tpath) as f:
            tj = json.load(f)
        traffic, traffic_src = tj["traffic_bytes_per_launch"], "profiles/r01_gemm_traffic.json (rocprofv3 --pmc, %d launches)" % tj["launches"]

    result = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        job_rate = args.steps / elapsed
        achieved = gs["flops"] / max(gs["ms"], 1e-9) / 1e9
        result = {
            "metric": "GP fit+predict iters/sec at N=%d D=%d" % (N, D),
            "value": world * job_rate,
            "unit": "fit+predict iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C3: N=%d, D=%d %s iso, fit (K, Cholesky, alpha, LML) + predict mean/var at "
                                   "M=%d candidates per GPU + EI arg-best" % (N, D, args.kernel, M),
                       "noise": 1e-2, "candidates_per_gpu": M, "fit": "replicated on every rank", "collective": collective,
                       "job_iters_per_s": job_rate, "lml": out[0], "best_candidate": int(out[1]),
                       "phases_ms_last_timed_call": phases_timed,
                       "phases_ms": phases, "phases_note": "phases_ms: gp_fit and gp_predict run one after the other "
                                                           "after the timed region (per-phase rates below come from it)",
                       "entry_point": "gp_fit_predict" if fused else "gp_fit + gp_predict",
                       "other_call_pattern": None if pipelined_ms is None else {
                           "entry_point": "gp_fit + gp_predict" if fused else "gp_fit_predict",
                           "ms_per_step": pipelined_ms, "iters_per_s": 1e3 / pipelined_ms, "phases_ms": phases_pipelined,
                           "roofline_same_kernel": other_roofline},
                       "cholesky_tflops": chol["flops"] / chol["ms"] / 1e9,
                       "cholesky_frac_of_fp64_mfma_peak": chol["flops"] / chol["ms"] / 1e9 / FP64_MFMA_PEAK_TFLOPS,
                       "cand_solve_tflops": solve["flops"] / solve["ms"] / 1e9},
            "roofline": {"bound": "mfma", "kernel": "gemm_nt_kernel<1, 128, 4, false, 128> (C -= A B^T on fp64 v_mfma_f64_16x16x4_f64, 8 waves)",
                         "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_unit": "bytes per launch",
                         "traffic_source": traffic_src,
                         "note": ("in gp_fit_predict three candidate-update launches and the trailing updates of the "
                                  "factorisation's tail run CONCURRENTLY (that is where the entry point gains its 5 %): their "
                                  "durations, hence this average, include the time they share the chip; achieved_while_running = the "
                                  "same flops / the union of the launches' intervals; --separate-calls times the same kernel "
                                  "without overlap (0.74 of peak, profiles/)") if fused else None,
                         "launches": gs["launches"], "kernel_ms_total": gs["ms"],
                         "launch_filter": "every launch of that kernel symbol in the timed region (launches of >= 1400 output tiles: trailing updates, candidate updates; > 90 % of the flops)", "avg_launch_ms": gs["ms"] / max(gs["launches"], 1),
                         "busy_ms_total": busy_ms, "achieved_while_running": gs["flops"] / max(busy_ms, 1e-9) / 1e9,
                         "frac_while_running": gs["flops"] / max(busy_ms, 1e-9) / 1e9 / FP64_MFMA_PEAK_TFLOPS,
                         "flops_per_launch_avg": gs["flops"] / max(gs["launches"], 1)},
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(N, D, M)
        print(json.dumps(result))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    h.close()


if __name__ == "__main__":
    main()
