#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json on MI355X: GP fit+predict iterations/s at N=16384, D=8.

A "step" is one pass of the hot path over one batch of synthetic input, inputs resident in HBM before the timed region,
hyper-parameters fixed.

--gpus 1 (default)  workload C3 = BASELINE.json configs[2], the configuration the metric is quoted on:
    fit (K build -> Cholesky -> alpha -> LML, rows A1-A5) + predict on 10 000 resident candidates (A6-A7), both
    through ONE call (gp_fit_predict; --separate-calls times gp_fit + gp_predict instead), + EI scoring + device
    arg-best (A9-A11, A14).  value = iterations/s.

--gpus N > 1        the SAME workload, strong scaling: one model, the same table of 10 000 candidates (seed 1236, the table
    of the N = 1 run) split over the ranks in contiguous row blocks.  Per iteration every rank refits the model (the fit does
    not shard: replicas only; a broadcast of the 2.1 GB factor from one fitting rank would take longer than
    the 28 ms refit it saves -- gp_comm_bcast_fit exists for callers who fit elsewhere), scores its block, reduces a
    device arg-best, and ONE RCCL all-gather of the per-rank (best value, global row) pair over xGMI + the lowest-index
    merge give every rank the job's winner -- the winner of the N = 1 run.  value = job iterations/s (NOT multiplied by
    the rank count); expectation ~ 1 / (fit + solve / N).  BASELINE.json's own sharding
    configuration (C4 = configs[3]: Matern-5/2, ONE table of 10^6 rows split over the ranks) is measured in the same run
    as a second object of the line (c4_sharded); `--workload c4` runs it as the line itself.
    One process per GPU: either launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
    environment), or -- called plainly as `python bench.py --gpus N` -- this process becomes a launcher that makes NO HIP /
    RCCL call itself and starts N fresh rank processes (launch_ranks below; never a re-exec of a process that touched the
    GPU).  No rank imports torch: the 128-byte RCCL unique id and the end-of-run records travel over a local socket between
    the ranks (Rendezvous below), and with the RCCL communicator up the barrier and the max-over-ranks of the elapsed time
    are all-gathers inside libgphip (gp_comm_allgather_best); the JSON line records which librccl the process mapped and
    the version it reports.  The ranks compare their merged winners at the end; a disagreement is exit code 3.
"""
import argparse
import hashlib
import json
import os
import signal
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X dense FP64 matrix peak (32 flop/clk/SIMD x 1024 SIMDs x 2.4 GHz); see DESIGN.md
HBM_PEAK_GBS = 8000.0         # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.3 achievable)
GEMM_SYMBOL = "gemm_nt_kernel<1, 128, 4, false, 128>"


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


def shard_bounds(M, rank, nranks):
    base, rem = divmod(int(M), int(nranks))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def launch_ranks(n, argv, child_cmd=None, poll_s=0.2, grace_s=20.0, timeout_s=None):
    """`python bench.py --gpus N` without a launcher around it: start N rank processes of this script and wait.

    The parent touches neither HIP nor RCCL nor torch (nothing GPU-related is imported before this point), so the
    children are fresh processes, not re-execs of a process that initialised the GPU.  Each child gets RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR=127.0.0.1 and a free MASTER_PORT -- exactly what torch.distributed.run would hand it, so
    the rank code below is the same either way.  Rank 0 inherits stdout (its ONE JSON line is the job's output); the other
    ranks' stdout goes to stderr, like every rank's stderr.  When a rank exits non-zero the others get `grace_s` seconds
    (they may be about to fail the same way), then are terminated by PID; the launcher returns the first non-zero exit
    code.  `timeout_s` (--launch-timeout, GPHIP_BENCH_TIMEOUT) bounds the whole job: a rank stuck in a collective or behind
    a GPU queue that stopped draining ends the job with code 124 instead of polling for ever.  SIGTERM / SIGINT to the
    launcher are passed on to the ranks (a supervisor that signals only this PID does not leave them holding the GPUs)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()   # (the ranks rendezvous over a socket file keyed by this number and the launcher's PID, not over the port)
    cmd = list(child_cmd) if child_cmd is not None else [sys.executable, os.path.abspath(__file__)] + list(argv)
    # the ranks' control channel lives in a directory only this user can enter, under a key nobody can guess (the channel
    # carries pickled objects: a predictable socket name in the shared temp directory would let another local user answer it)
    import secrets
    rdv_dir = tempfile.mkdtemp(prefix="gphip_bench_")
    rdv_key = secrets.token_hex(16)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GPHIP_BENCH_LAUNCHED="1", GPHIP_BENCH_RDV_DIR=rdv_dir,
                   GPHIP_BENCH_RDV_KEY=rdv_key)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this host driver
        procs.append(subprocess.Popen(cmd, env=env, stdout=None if r == 0 else sys.stderr))

    def stop_all(sig_first=signal.SIGTERM):
        for p in procs:
            if p.poll() is None:
                p.send_signal(sig_first)
        for p in procs:
            try:
                p.wait(10)
            except subprocess.TimeoutExpired:
                p.kill()
    caught = []
    previous = {}
    for sg in (signal.SIGTERM, signal.SIGINT):
        try:
            previous[sg] = signal.signal(sg, lambda num, frame: caught.append(num))
        except ValueError:   # not the main thread (a test harness): no handlers, the rest is unchanged
            pass
    rc, t_fail, t0 = 0, None, time.monotonic()
    try:
        while any(p.poll() is None for p in procs):
            if caught:
                stop_all(caught[0])
                return 128 + int(caught[0])
            for p in procs:
                if p.poll() not in (None, 0) and rc == 0:
                    rc, t_fail = p.returncode, time.monotonic()
            if t_fail is not None and time.monotonic() - t_fail > grace_s:
                stop_all()
                break
            if timeout_s is not None and time.monotonic() - t0 > timeout_s:
                sys.stderr.write("bench launcher: the job exceeded %.0f s, stopping the ranks\n" % timeout_s)
                stop_all()
                return 124
            time.sleep(poll_s)
    finally:
        for sg, h in previous.items():
            signal.signal(sg, h)
        import shutil
        shutil.rmtree(rdv_dir, ignore_errors=True)
    for p in procs:
        if p.returncode not in (None, 0) and rc == 0:
            rc = p.returncode
    return rc if rc >= 0 else 128 - rc   # a signal's negative code as the shell reports it


def rendezvous_place():
    """(authkey, socket path) of the ranks' control channel.  Started by bench.py's own launcher: the private directory and
    the random key it made (GPHIP_BENCH_RDV_DIR / GPHIP_BENCH_RDV_KEY).  Started by torch.distributed.run: a per-user
    directory under the temp directory, created 0700 and REFUSED unless it is a real directory owned by this user with no
    access for anybody else; the key is derived from the agent's run id, MASTER_PORT and the agent's PID."""
    port, ppid = os.environ.get("MASTER_PORT", "0"), os.getppid()
    d, k = os.environ.get("GPHIP_BENCH_RDV_DIR"), os.environ.get("GPHIP_BENCH_RDV_KEY")
    if not d:
        d = os.path.join(tempfile.gettempdir(), "gphip_bench_uid%d" % os.getuid())
        try:
            os.mkdir(d, 0o700)
        except FileExistsError:
            pass
        k = hashlib.sha256(("%s|%s|%d" % (os.environ.get("TORCHELASTIC_RUN_ID", ""), port, ppid)).encode()).hexdigest()
    st = os.lstat(d)
    import stat
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RuntimeError("rendezvous directory %s is not a private directory of this user" % d)
    return k.encode(), os.path.join(d, "rdv_%s_%d.sock" % (port, ppid))


class Rendezvous(object):
    """The ranks' control channel on ONE node, without torch: rank 0 listens on a socket file in a private directory
    (rendezvous_place), named by MASTER_PORT and the PID of the process that started the ranks (bench.py's launcher or
    torch.distributed.run: the same parent for every rank), the others connect to it.  It carries small Python objects only -- the 128-byte RCCL unique id,
    flags, the end-of-run records, and the (value, row) pairs when no RCCL communicator can be built; the data path's
    collective is RCCL inside libgphip.  (MASTER_PORT itself is not bound: under torch.distributed.run the agent's own store
    listens there.)"""

    def __init__(self, rank, world, timeout_s=120.0):
        from multiprocessing.connection import Client, Listener
        self.rank, self.world = rank, world
        key, path = rendezvous_place()
        self.path = path
        self.peers = []
        if rank == 0:
            if os.path.exists(path):
                os.unlink(path)          # left behind by a job of the same parent that died
            self.listener = Listener(path, family="AF_UNIX", authkey=key)
            self.listener._listener._socket.settimeout(timeout_s)
            by_rank = {}
            for _ in range(world - 1):
                c = self.listener.accept()
                by_rank[c.recv()] = c
            self.peers = [by_rank[r] for r in range(1, world)]
        else:
            t_end = time.monotonic() + timeout_s
            while True:
                try:
                    self.conn = Client(path, family="AF_UNIX", authkey=key)
                    break
                except (FileNotFoundError, ConnectionRefusedError):
                    if time.monotonic() > t_end:
                        raise RuntimeError("rank %d: rank 0 never opened %s" % (rank, path))
                    time.sleep(0.05)
            self.conn.send(rank)

    def gather(self, obj):
        """Rank 0 gets [obj of rank 0, obj of rank 1, ...], the others None."""
        if self.rank == 0:
            return [obj] + [c.recv() for c in self.peers]
        self.conn.send(obj)
        return None

    def bcast(self, obj):
        """Rank 0's obj on every rank."""
        if self.rank == 0:
            for c in self.peers:
                c.send(obj)
            return obj
        return self.conn.recv()

    def allgather(self, obj):
        return self.bcast(self.gather(obj))

    def barrier(self):
        self.allgather(None)

    def close(self):
        if self.rank == 0:
            for c in self.peers:
                c.close()
            self.listener.close()
            if os.path.exists(self.path):
                os.unlink(self.path)
        else:
            self.conn.close()


def on_stderr(fn):
    """Run fn() with file descriptor 1 pointing at stderr: librccl prints its version banner with printf when a communicator is
    set up, and the job's stdout carries ONE JSON line.  The C-level buffer is flushed before the descriptor is put back."""
    import ctypes
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        return fn()
    finally:
        ctypes.CDLL(None).fflush(None)
        os.dup2(saved, 1)
        os.close(saved)


def rccl_identity():
    """Which librccl this process has mapped (the first mapping of /proc/self/maps whose file name says rccl) -- libgphip is
    linked against /opt/rocm/lib/librccl.so.1; a python package that bundles its own copy could get in first."""
    found = []
    try:
        with open("/proc/self/maps") as f:
            for ln in f:
                path = ln.split()[-1]
                if "rccl" in os.path.basename(path) and path not in found:
                    found.append(path)
    except OSError:
        pass
    return found


def cpu_baseline(N, D, M, sample_only=False):
    """The oracle (NumPy/SciPy restatement of the reference's path, same LAPACK entry points as GPy) timed on this
    host's cores, phase by phase, with every BLAS pool limited to the CPUs this job may use (oracle.usable_cpus: the
    affinity mask cut by the cgroup quota -- BENCH_r02 ran 64 OpenBLAS threads on a 16-CPU share and measured dpotrf at
    30 GFLOP/s).  Default: every phase at the quoted size itself, nothing scaled (about 35 s on a 16-CPU share): the MINIMAL
    path (K, dpotrf, dpotrs, K*, dtrtrs -- what is algebraically needed; `value`) and what GPy does on top (pdinv's dtrtri,
    linalg.py:209; dpotri + two symmetrify, :144,210-212; one get_fmin recomputation, GPyOpt models/gpmodel.py:125-129).
    --cpu-baseline-sample runs a half-size sample with each phase scaled by its own exponent instead (round-2 behaviour)."""
    from oracle import cpu_ref as O
    limiter, cores = O.limit_blas_threads()
    try:
        from threadpoolctl import threadpool_info
        blas = [(d.get("internal_api"), d.get("num_threads")) for d in threadpool_info()]
    except Exception:  # noqa: BLE001
        blas = []
    kern = O.RBF(D, 1.0, O.default_lengthscale(D, False))
    # one small untimed pass first: BLAS thread pool, allocator and page-fault warm-up otherwise land in the first phase
    Xw, Yw, Xsw = O.synthetic_problem(1536, D, 512, seed=1)
    O.fit_predict_iteration(kern, Xw, Yw, 1e-2, Xsw, as_gpy=True)

    def run(Ns, Ms, minimal_part, extras_part):
        rn, rm = N / Ns, M / Ms
        X, Y, Xs = O.synthetic_problem(Ns, D, Ms, seed=1234)
        ph, scale = {}, {}

        def timed(name, factor, fn):
            t0 = time.perf_counter()
            out = fn()
            ph[name] = time.perf_counter() - t0
            scale[name] = factor
            return out
        Ky = timed("K_build", rn ** 2, lambda: kern.K(X))
        O.diag_add(Ky, 1e-2 + 1e-8)
        L = timed("dpotrf", rn ** 3, lambda: O.jitchol(Ky)[0])
        del Ky
        alpha = timed("dpotrs_alpha_lml", rn ** 2, lambda: O.dpotrs(L, Y, lower=1)[0])
        Kx = timed("K_cross", rn * rm, lambda: kern.K(X, Xs))
        tmp = timed("dtrtrs_var", rn ** 2 * rm, lambda: (np.dot(Kx.T, alpha), 1.0 - np.square(O.dtrtrs(L, Kx)[0]).sum(0)))
        del Kx, tmp
        mini = {k: ph[k] for k in ph}
        if not minimal_part:
            ph.clear()
        if extras_part:   # what GPy does on top (pdinv, get_fmin)
            timed("dtrtri_unused_by_inference", rn ** 3, lambda: O.dtrtri(L))
            Wi = timed("dpotri", rn ** 3, lambda: O.dpotri(L, lower=1)[0])
            timed("symmetrify_x2", rn ** 2, lambda: (O.symmetrify(Wi), O.symmetrify(Wi)))
            del Wi
            Kxx = timed("get_fmin_K", rn ** 2, lambda: kern.K(X, X))
            timed("get_fmin_dtrtrs", rn ** 3, lambda: O.dtrtrs(L, Kxx)[0])
        return ph, scale, mini

    half = (N // 2, M // 2)
    mini_keys = ["K_build", "dpotrf", "dpotrs_alpha_lml", "K_cross", "dtrtrs_var"]
    if sample_only:
        ph, scale, _ = run(half[0], half[1], True, True)
        at_size = {k: ph[k] * scale[k] for k in ph}
        sample = ("N=%d, M=%d (half of N=%d, M=%d in both), each phase scaled by its own exponent; measured %.1f s of CPU work"
                  % (half[0], half[1], N, M, sum(ph.values())))
    else:
        # the quoted size itself, every phase, nothing scaled.  (A half-size sample is NOT a fair stand-in on these hosts:
        # at N = 8192 OpenBLAS' dpotrf / dsyrk run 5-8x below their N = 16384 rate -- measured in BENCH_r02 and again in
        # round 3 with the thread count pinned, so it is the size, not oversubscription.)
        ph, scale, _ = run(N, M, True, True)
        at_size = dict(ph)
        sample = "the quoted size itself for every phase: N=%d, M=%d; %.1f s of CPU work, nothing scaled" % (N, M, sum(ph.values()))
    measured = dict(ph)
    minimal = sum(at_size[k] for k in mini_keys)
    as_gpy = sum(at_size.values())
    del limiter
    return {"value": 1.0 / minimal, "unit": "fit+predict iters/s", "cores": int(cores), "kind": "port",
            "sample": sample, "threads_note": "BLAS pools limited to %d threads = the CPUs this job may use "
                                              "(affinity / cgroup quota; os.cpu_count() says %s)" % (cores, os.cpu_count()),
            "total_minimal_s": minimal, "total_as_gpy_does_it_s": as_gpy, "value_as_gpy_does_it": 1.0 / as_gpy,
            "phases_measured_s": {k: round(v, 3) for k, v in measured.items()},
            "phases_at_quoted_size_s": {k: round(v, 3) for k, v in at_size.items()}, "blas": blas,
            "full_size_run": full_size_record()}


def small_calls(sizes=(512, 2048, 16384), rows=(1, 5, 1000), reps=20, cpu=True):
    """One-row call latencies: what the acquisition optimiser hammers between two fits -- scipy L-BFGS-B from each anchor makes
    single-row calls of acquisition_function_withGradients (GPyOpt/GPyOpt/optimization/optimizer.py:28-61 ->
    models/gpmodel.py:131-142 -> GPy/GPy/core/gp.py:407-454), the anchor scoring one call of 1000 rows
    (anchor_points_generator.py:85-98).  Per (N, M): wall time of `gp_set_candidates + gp_predict` and of
    `gp_set_candidates + gp_acq_grad` (EI) through the C ABI on the fitted model, and beside them the CPU oracle's time for the
    same two calls (GPModel.predict / acquisition_function_withGradients restated; the model fitted once, Ky^-1 cached as
    GPy's posterior does) with the BLAS pool limited to this job's CPUs.  D = 8, RBF, noise 1e-2."""
    from gaussian_process_optimization_amd import _lib
    h = _lib.Handle(0)
    h.set_option("emulate_fp64", 0)
    # Steady state: the inverse factor of the one-location path exists (option rows_build = 1).  By default the library builds it
    # at the first call for N <= 4096 and, above that, only after N / 768 calls since the fit, which go through substitutions
    # against L instead (include/gphip.h): the cost of such a call is reported beside the steady-state figure (gpu_acq_grad_rented_ms).
    h.set_option("rows_build", 1)
    rows_out = []
    if cpu:
        from oracle import cpu_ref as O
        limiter, cores = O.limit_blas_threads()
    for N in sizes:
        D = 8
        rng = np.random.default_rng(1)
        X = rng.uniform(0, 1, (N, D))
        Y = np.sin(2 * np.pi * X).sum(1, keepdims=True) / np.sqrt(D) + 0.05 * rng.standard_normal((N, 1))
        ls = 0.25 * np.sqrt(D)
        h.set_data(X, Y)
        h.set_params(_lib.GP_KERNEL_RBF, 0, 1.0, [ls], 1e-2)
        h.fit()
        fmin = h.fmin()
        gm0 = None
        if cpu:
            gp0 = O.OracleGP(X, Y, O.RBF(D, 1.0, ls), 1e-2)
            gm0 = O.OracleGPModel(gp0)
            gp0.posterior                         # the fit (pdinv) happens here, outside the timed calls
            gm0.predict_withGradients(X[:1])      # ... and the lazy Ky^-1 of the posterior
        for M in rows:
            Xs = rng.uniform(0, 1, (M, D))

            # up to 8 locations go down as ONE call taking them by value (gp_predict_rows / gp_acq_rows: what the host layer
            # issues for them); more than that is gp_set_candidates + the batched call
            def dev_predict():
                if M <= 8:
                    return h.predict_rows(Xs, True)
                h.set_candidates(Xs)
                return h.predict(True)

            def dev_grad():
                if M <= 8:
                    return h.acq_rows(Xs, _lib.GP_ACQ_EI, 0.01, fmin, grad=True)
                h.set_candidates(Xs)
                return h.acq_grad(_lib.GP_ACQ_EI, 0.01, fmin)

            def timed(fn, n):
                fn()
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                return (time.perf_counter() - t0) / n * 1e3
            rec = {"N": N, "M": M, "gpu_acq_grad_ms": timed(dev_grad, reps), "gpu_predict_ms": timed(dev_predict, reps)}
            if M <= 8 and N > 4096:
                h.set_option("rows_build", 0)          # the calls before the factor exists: substitutions against L
                h.fit()                                # (a fit drops the factor)
                rec["gpu_acq_grad_rented_ms"] = timed(dev_grad, reps)
                h.set_option("rows_build", 1)
            if cpu:
                ncpu = 3 if N >= 8192 else reps
                rec["cpu_predict_ms"] = timed(lambda: gm0.predict(Xs), ncpu)
                rec["cpu_acq_grad_ms"] = timed(lambda: O.acq_EI_withGradients(gm0, Xs, 0.01, fmin), ncpu)
                rec["cpu_threads"] = int(cores)
            rows_out.append(rec)
            sys.stderr.write(json.dumps(rec) + "\n")
    h.close()
    return rows_out


def full_size_record():
    """The committed one-off run of --cpu-baseline-full on a GPU box's host (static, for orientation beside the sample)."""
    path = os.path.join(ROOT, "profiles", "r02_cpu_baseline_full.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        c = json.load(f)["cpu_baseline"]
    return {"static": True, "file": "profiles/r02_cpu_baseline_full.json", "cores": c["cores"],
            "total_minimal_s": c["total_minimal_s"], "total_as_gpy_does_it_s": c["total_as_gpy_does_it_s"],
            "iters_per_s_minimal": c["value"], "phases_s": c["phases_measured_s"]}


def static_traffic(N, D, M):
    """HBM-side traffic of the dominant kernel from the committed PMC passes of THIS command (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate runs, tools/pmc_traffic.py).  Not measured by this run: carried with its
    provenance and dropped when gemm.hip has changed since the passes were taken."""
    tpath = os.path.join(ROOT, "profiles", "r04_gemm_traffic.json")
    if not os.path.exists(tpath) or (N, D, M) != (16384, 8, 10000):
        return None, None
    with open(tpath) as f:
        tj = json.load(f)
    src = os.path.join(ROOT, "gaussian_process_optimization_amd", "csrc", "gemm.hip")
    sha = hashlib.sha256(open(src, "rb").read()).hexdigest()[:16]
    if tj.get("gemm_hip_sha256_16") != sha:
        return None, {"static": True, "dropped": "gemm.hip changed since the counter passes (%s != %s)"
                                                 % (tj.get("gemm_hip_sha256_16"), sha)}
    return tj["traffic_bytes_per_launch"], {"static": True, "file": "profiles/r04_gemm_traffic.json",
                                            "kernel": GEMM_SYMBOL, "launches": tj["launches"],
                                            "gemm_hip_sha256_16": sha,
                                            "algorithmic_bytes_per_launch": tj.get("algorithmic_bytes_per_launch")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="auto", choices=["auto", "c3", "c4"])
    ap.add_argument("--N", type=int, default=16384)
    ap.add_argument("--D", type=int, default=8)
    ap.add_argument("--M", type=int, default=0, help="candidates: C3 default 10 000 (per GPU), C4 default 10^6 (whole table)")
    ap.add_argument("--panel-tiles", type=int, default=0)
    ap.add_argument("--option", action="append", default=[], help="name=value passed to gp_set_option (tuning runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-sample", action="store_true", help="every phase on the half-size sample, scaled (round 2)")
    ap.add_argument("--separate-calls", action="store_true",
                    help="C3: time gp_fit + gp_predict as two calls instead of the one-call entry point gp_fit_predict")
    ap.add_argument("--no-emulated-line", action="store_true", help="skip the second (int8-emulated) measurement")
    ap.add_argument("--small-calls", action="store_true",
                    help="instead of the bench line: one-row call latencies (M = 1, 5, 1000 at N = 512, 2048, 16384) of the "
                         "device path with the CPU oracle's beside them, as a text table (profiles/r04_small_calls.txt)")
    ap.add_argument("--launch-timeout", type=float, default=float(os.environ.get("GPHIP_BENCH_TIMEOUT", "0")) or None,
                    help="--gpus N without a launcher: stop the ranks and exit 124 after this many seconds")
    ap.add_argument("--no-c4-reference", action="store_true",
                    help="C4, N > 1: skip rank 0's un-timed single-GPU pass over the whole table (the strong-scaling base)")
    ap.add_argument("--no-c4-section", action="store_true",
                    help="N > 1, default workload: skip the second measurement (C4: ONE table of --c4-M candidates split over the ranks)")
    ap.add_argument("--c4-M", type=int, default=1000000, help="N > 1, default workload: size of the C4 section's candidate table")
    ap.add_argument("--c4-steps", type=int, default=3, help="timed iterations of the C4 section")
    args = ap.parse_args()

    if args.small_calls:
        recs = small_calls(cpu=not args.no_cpu_baseline)
        print("%6s %5s | %12s %12s | %12s %12s | %s" % ("N", "M", "GPU predict", "GPU acq_grad", "CPU predict", "CPU acq_grad",
                                                        "ms per call; GPU = set_candidates + call through the C ABI"))
        for r in recs:
            print("%6d %5d | %12.3f %12.3f | %12s %12s |%s" % (r["N"], r["M"], r["gpu_predict_ms"], r["gpu_acq_grad_ms"],
                                                              "%.3f" % r["cpu_predict_ms"] if "cpu_predict_ms" in r else "-",
                                                              "%.3f" % r["cpu_acq_grad_ms"] if "cpu_acq_grad_ms" in r else "-",
                                                              "  (acq_grad before the inverse factor exists -- the first N / 768 calls after a fit: %.3f)"
                                                              % r["gpu_acq_grad_rented_ms"] if "gpu_acq_grad_rented_ms" in r else ""))
        if recs and "cpu_threads" in recs[0]:
            print("CPU: the oracle (GPy-equivalent NumPy/SciPy path) with %d BLAS threads, model fitted and Ky^-1 cached "
                  "before the timed calls" % recs[0]["cpu_threads"])
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: become one (no GPU call in this process), one fresh child per rank
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], timeout_s=args.launch_timeout))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch as `python bench.py --gpus N` (starts its own ranks) or "
                         "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`" % (args.gpus, world))
    # Default workload at every N: the configuration the metric is quoted on (C3), as ONE job.  The fit does not shard (replicas
    # only); the candidates do: on N > 1 ranks the SAME 10^4-row table of the N = 1 run is split over the ranks, every rank refits
    # the model and scores its block, the ranks all-gather their arg-best over RCCL -- strong scaling, value = job iterations/s.
    # BASELINE.json's candidate-sharding configuration (C4: one table of 10^6 candidates split over the ranks) is measured in the
    # same run as a second object of the line (c4_sharded); `--workload c4` runs it as the line itself.
    workload = args.workload if args.workload != "auto" else "c3"
    with_c4 = args.workload == "auto" and world > 1 and not args.no_c4_section
    if world > 1:
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")   # RCCL's log lines belong on stderr: stdout carries the ONE JSON line
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (the host driver's only mode): set before HIP loads,
                                                                  # whoever launched this rank
        # A rank stuck inside a collective (a peer that never arrives, an RCCL bootstrap that finds no route) cannot be reached by a
        # Python-level handler while it sits in C: the default action of SIGALRM ends the process, whatever launched it.
        signal.signal(signal.SIGALRM, signal.SIG_DFL)
        signal.alarm(int(os.environ.get("GPHIP_BENCH_RANK_TIMEOUT", "1500")))
    rdv = Rendezvous(rank, world) if world > 1 else None   # control channel between the ranks (no torch in any rank)

    from gaussian_process_optimization_amd import _lib
    from gaussian_process_optimization_amd.sharded import merge_best
    # rehearsal switch for a one-GPU box: every rank on device 0 (RCCL then refuses the duplicate device and the
    # exchange falls back to the ranks' control channel); never set by the driver
    dev = 0 if os.environ.get("GPHIP_BENCH_SAME_DEVICE") else local_rank
    h = _lib.Handle(dev)
    h.set_option("emulate_fp64", 0)   # the headline is true fp64 whatever GPHIP_EMULATE_FP64 says (recorded in config)
    panel_cols = 768.0                 # inverted diagonal panels: panel_tiles (default 6) x 128 columns
    if args.panel_tiles:
        h.set_option("panel_tiles", args.panel_tiles)
        panel_cols = 128.0 * args.panel_tiles
    for kv in args.option:
        k, v = kv.split("=")
        h.set_option(k, int(v))
        if k == "panel_tiles":
            panel_cols = 128.0 * int(v)
    N, D = args.N, args.D
    if workload == "c3":
        M_total = args.M or 10000                    # ONE table for the whole job: the table of the N = 1 run
        lo, hi = shard_bounds(M_total, rank, world)
        M = hi - lo
        X, Y, table = synthetic(N, D, M_total, cand_seed=1236)
        Xs = table[lo:hi].copy()
        del table
        kid, kname = _lib.GP_KERNEL_RBF, "RBF"
    else:
        M_total = args.M or 1000000
        lo, hi = shard_bounds(M_total, rank, world)
        M = hi - lo
        X, Y, _ = synthetic(N, D, 8)
        # ONE table for the whole job (seed 1236), this rank's contiguous block of it
        Xs = np.random.default_rng(1236).uniform(0, 1, (M_total, D))[lo:hi].copy()
        kid, kname = _lib.GP_KERNEL_MATERN52, "Matern52"
    h.set_data(X, Y)
    h.set_params(kid, 0, 1.0, [0.25 * np.sqrt(D)], 1e-2)
    h.set_candidates(Xs)
    collective, rccl_ranks = None, None
    if world > 1:
        # the one data-path collective: an all-gather of 16 bytes per rank over RCCL.  If the RCCL
        # communicator cannot be built on this node every rank agrees to exchange the pairs over the control channel
        # instead -- said in the JSON line: such a line is not an RCCL measurement.
        ok, uid = 1, None
        if rank == 0:
            try:
                uid = on_stderr(h.comm_unique_id)
            except Exception as e:  # noqa: BLE001
                sys.stderr.write("rank 0: RCCL unique id failed: %s\n" % e)
        uid = rdv.bcast(uid)
        if uid is None:
            ok = 0
        else:
            try:
                on_stderr(lambda: h.comm_init(uid, rank, world))
                rccl_ranks = h.comm_info()[1]
            except Exception as e:  # noqa: BLE001
                ok = 0
                sys.stderr.write("rank %d: RCCL comm init failed: %s\n" % (rank, e))
        collective = "rccl" if min(rdv.allgather(ok)) == 1 else \
            "host sockets between the ranks (RCCL communicator unavailable: NOT an RCCL measurement)"

    def barrier():
        h.synchronize()
        if rdv is not None:
            if collective == "rccl":
                h.comm_allgather_best(0.0, rank, world)     # an all-gather over the communicator is a barrier of its ranks
            else:
                rdv.barrier()
        h.synchronize()

    def max_over_ranks(x):
        if rdv is None:
            return x
        if collective == "rccl":
            return float(np.max(h.comm_allgather_best(float(x), rank, world)[0]))
        return max(rdv.allgather(float(x)))

    def exchange(val, gidx):
        if world == 1:
            return gidx, val
        if collective == "rccl":
            vals, idxs = h.comm_allgather_best(val, gidx, world)
        else:
            pairs = rdv.allgather((float(val), int(gidx)))
            vals = np.array([p[0] for p in pairs])
            idxs = np.array([p[1] for p in pairs], dtype=np.int64)
        return merge_best(vals, idxs, -1)

    fused = workload == "c3" and not args.separate_calls

    def step():
        if fused:  # gp_fit + gp_predict as one pipelined pass (bitwise the same results; include/gphip.h)
            (lml, logdet, jit), mu, var = h.fit_predict(True)
        else:
            lml, logdet, jit = h.fit()
            if workload == "c3":
                mu, var = h.predict(True)
        fmin = h.fmin()
        idx, val = h.acq_argbest(_lib.GP_ACQ_EI, 0.01, fmin, -1)   # C4: posterior + EI over the block, chunked
        gi, gv = exchange(val, lo + idx)
        return lml, gi, gv

    # C4 on N > 1 ranks: rank 0 times one un-profiled pass over the WHOLE table on its own GPU first (the single-GPU
    # base the sharded rate is to be read against); the other ranks wait at the barrier
    c4_ref = None
    if workload == "c4" and world > 1 and not args.no_c4_reference:
        if rank == 0:
            full = np.random.default_rng(1236).uniform(0, 1, (M_total, D))
            h.set_candidates(full)
            h.fit(); h.acq_argbest(_lib.GP_ACQ_EI, 0.01, h.fmin(), -1)     # warm-up (allocations)
            h.synchronize()
            t0 = time.perf_counter()
            h.fit()
            i1, v1 = h.acq_argbest(_lib.GP_ACQ_EI, 0.01, h.fmin(), -1)
            h.synchronize()
            c4_ref = {"single_gpu_iter_s": time.perf_counter() - t0, "best_row": int(i1), "best_value": v1}
            del full
            h.set_candidates(Xs)
        barrier()

    for _ in range(args.warmup):
        out = step()
    h.profile(True)          # HIP events around every launch of the dominant kernel (fp64 MFMA GEMM), on its stream
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    gs = h.gemm_stats()
    busy_ms = h.gemm_busy()
    phases_timed = {p["name"]: round(p["ms"], 3) for p in h.phases()}   # of the last call of the last timed step
    h.profile(False)
    ranks_agree, rank_records = True, None
    if rdv is not None:
        elapsed = max_over_ranks(elapsed)
        # every rank merged the same gathered pairs: the winners must be identical
        rank_records = rdv.allgather({"rank": rank, "best_row": int(out[1]), "best_value": float(out[2]),
                                      "lml": float(out[0])})
        ranks_agree = all(r["best_row"] == rank_records[0]["best_row"] and
                          r["best_value"] == rank_records[0]["best_value"] for r in rank_records)

    # second, clearly labelled measurement (NOT the headline): the same C3 step with the candidate solve's updates on the
    # int8 matrix cores in residue form (option "emulate_fp64", csrc/rns.hip; fp64-equivalent results, parity-tested in
    # tests/test_gpu_emulation.py).  The headline above stays true fp64.
    emulated = None
    if workload == "c3" and world == 1 and not args.no_emulated_line:
        h.set_option("emulate_fp64", 1)
        ref_best = (out[1], out[2])
        step()
        h.profile(True)          # HIP events around every launch of the residue GEMM, on its stream
        h.synchronize()
        te0 = time.perf_counter()
        for _ in range(args.steps):
            oute = step()
        h.synchronize()
        te = (time.perf_counter() - te0) / args.steps
        rs = h.rns_stats()
        h.profile(False)
        phe = {p["name"]: round(p["ms"], 3) for p in h.phases()}
        h.set_option("emulate_fp64", 0)
        emulated = {"label": "NOT the headline: the factorisation's trailing update and the candidate solve's updates emulated on int8 "
                             "MFMA (exact residue arithmetic, 14 moduli; option emulate_fp64); chain, panel products and all "
                             "reductions in true fp64; entry points gp_fit + gp_predict",
                    "ms_per_step": te * 1e3, "iters_per_s": 1.0 / te, "phases_ms_predict": phe,
                    "same_best_candidate": bool(int(oute[1]) == int(ref_best[0])),
                    "best_value_rel_diff": abs(oute[2] - ref_best[1]) / max(abs(ref_best[1]), 1e-300),
                    "lml_equal": bool(oute[0] == out[0]),
                    # the dominant kernel of THIS line, priced like the headline's: int8 operations of the blocks each launch
                    # computes / summed launch durations (HIP events, live) against the dense int8 MFMA peak; what caps it
                    # below that peak (operand staging, 2.7 Pop/s; a bare MFMA loop sustains 3.76) is in DESIGN.md section 9
                    "int8_gemm": {"kernel": "rns_gemm256_kernel (v_mfma_i32_32x32x32_i8, 256 x 256 tiles, 14 moduli)",
                                  "bound": "mfma", "launches": rs["launches"], "kernel_ms_total": rs["ms"],
                                  "avg_launch_ms": rs["ms"] / max(rs["launches"], 1),
                                  "achieved": rs["ops"] / max(rs["ms"], 1e-9) / 1e9, "peak": 5000.0, "unit": "Top/s",
                                  "frac": rs["ops"] / max(rs["ms"], 1e-9) / 1e9 / 5000.0,
                                  "ops_per_launch_avg": rs["ops"] / max(rs["launches"], 1)}}

    # un-timed: per-phase rates from gp_fit and gp_predict run one after the other (the median of three calls each: a single call
    # right after the emulated section has been seen 4 ms off)
    def phases_of(fn):
        runs = []
        for _ in range(3):
            fn()
            runs.append(h.phases())
        return sorted(runs, key=lambda ph: sum(p["ms"] for p in ph))[1]
    ph_fit = phases_of(h.fit)
    fit_ms = sum(p["ms"] for p in ph_fit)
    if workload == "c3":
        ph_pred = phases_of(lambda: h.predict(True))
    else:
        h.acq_argbest(_lib.GP_ACQ_EI, 0.01, h.fmin(), -1)
        ph_pred = h.phases()
    phases = {p["name"]: round(p["ms"], 3) for p in ph_fit}
    for p in ph_pred:   # (a chunked predict reports its LAST chunk's phases: gp_last_phases holds one call's worth)
        phases[p["name"]] = round(p["ms"], 3)
    chol = [p for p in ph_fit if p["name"] == "cholesky"][0]
    kb = [p for p in ph_fit if p["name"] == "kbuild"][0]
    solve = [p for p in ph_pred if p["name"].startswith("cand_solve")][-1]

    # un-timed: the second fp64 symbol (15 % of GPU time in r02's kernel stats, no flop accounting there): events around
    # every launch of gemm_nt_kernel<1, 64, 2, false, 64> in ONE extra step (they stall the latency chain, so not in the
    # timed region)
    h.profile(2)
    step()
    cs = h.gemm_stats()
    h.profile(False)

    # un-timed: the dominant kernel WITHOUT a neighbour of its own kind -- gp_fit then gp_predict as two calls (no candidate update
    # shares the chip with a trailing update), HIP events around every launch of the symbol as in the timed region.  A rocprofv3
    # --stats run of `bench.py --separate-calls` averages exactly these launches (profiles/r05_bench_kernel_stats_fp64.csv).
    sep = None
    if workload == "c3":
        h.fit(); h.predict(True)
        h.profile(True)
        for _ in range(3):
            h.fit(); h.predict(True)
        sep = h.gemm_stats()
        sep["busy_ms"] = h.gemm_busy()
        h.profile(False)

    # N > 1, default workload: the candidate-sharding configuration in the same run (see above).  ONE table for the whole job
    # (seed 1236), this rank's contiguous block of it; rank 0 first times one pass over the WHOLE table on its own GPU (the
    # single-GPU base the sharded rate is read against) while the others wait at the barrier.
    c4_sec = None
    if with_c4:
        M4 = args.c4_M
        lo4, hi4 = shard_bounds(M4, rank, world)
        table = np.random.default_rng(1236).uniform(0, 1, (M4, D))
        h.set_params(_lib.GP_KERNEL_MATERN52, 0, 1.0, [0.25 * np.sqrt(D)], 1e-2)

        def c4_iter(first_row):
            lml4 = h.fit()[0]
            i4, v4 = h.acq_argbest(_lib.GP_ACQ_EI, 0.01, h.fmin(), -1)
            return (lml4,) + tuple(exchange(v4, first_row + i4))
        ref4 = None
        if rank == 0 and not args.no_c4_reference:
            h.set_candidates(table)
            h.fit(); h.acq_argbest(_lib.GP_ACQ_EI, 0.01, h.fmin(), -1)     # warm-up (allocations)
            h.synchronize()
            t4 = time.perf_counter()
            h.fit()
            i1, v1 = h.acq_argbest(_lib.GP_ACQ_EI, 0.01, h.fmin(), -1)
            h.synchronize()
            ref4 = {"single_gpu_iter_s": time.perf_counter() - t4, "best_row": int(i1), "best_value": v1}
        h.set_candidates(table[lo4:hi4].copy())
        del table
        barrier()
        out4 = c4_iter(lo4)                                                  # warm-up
        barrier()
        t4 = time.perf_counter()
        for _ in range(args.c4_steps):
            out4 = c4_iter(lo4)
        barrier()
        el4 = max_over_ranks(time.perf_counter() - t4)
        recs4 = rdv.allgather({"rank": rank, "best_row": int(out4[1]), "best_value": float(out4[2])})
        agree4 = all(r["best_row"] == recs4[0]["best_row"] and r["best_value"] == recs4[0]["best_value"] for r in recs4)
        ranks_agree = ranks_agree and agree4
        h.fit()
        fit4 = sum(ph_["ms"] for ph_ in h.phases())
        ms4 = el4 / args.c4_steps * 1e3
        c4_sec = {"workload": "C4 (BASELINE.json configs[3]): N=%d, D=%d Matern-5/2 iso, fit (replicated per rank) + posterior + EI over "
                              "ONE table of %d candidates split over %d ranks + device arg-best + RCCL all-gather of (best, row) + "
                              "lowest-index merge" % (N, D, M4, world),
                  "scaling": "strong", "candidates_total": M4, "candidates_this_rank": hi4 - lo4, "steps": args.c4_steps,
                  "ms_per_iter": ms4, "iters_per_s": 1e3 / ms4, "collective": collective,
                  "fit_ms_per_iter_not_scaling": round(fit4, 3), "predict_ei_ms_per_iter_this_rank": round(ms4 - fit4, 3),
                  "single_gpu_reference": ref4,
                  "speedup_vs_single_gpu": None if not ref4 else ref4["single_gpu_iter_s"] * 1e3 / ms4,
                  "best_row_matches_single_gpu": None if not ref4 else bool(ref4["best_row"] == int(out4[1])),
                  "ranks_agree_on_winner": agree4, "best_candidate_global_row": int(out4[1]), "best_value": float(out4[2])}

    result = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        job_rate = args.steps / elapsed
        achieved = gs["flops"] / max(gs["ms"], 1e-9) / 1e9              # per launch: flops / SUM of the launch durations
        achieved_busy = gs["flops"] / max(busy_ms, 1e-9) / 1e9         # flops / UNION of the launch intervals
        traffic, traffic_src = static_traffic(N, D, M) if workload == "c3" else (None, None)
        kb_gbs = kb["bytes"] / max(kb["ms"], 1e-9) / 1e6
        if workload == "c3":
            cfg = {"workload": "C3 (BASELINE.json configs[2]): N=%d, D=%d RBF iso, fit (K, Cholesky, alpha, LML) + predict "
                               "mean/var at M=%d candidates + EI arg-best" % (N, D, M_total),
                   "candidates_total": M_total, "candidates_this_rank": M,
                   "entry_point": "gp_fit_predict" if fused else "gp_fit + gp_predict"}
            # one job whatever N is: at N = 1 the work per GPU is the whole job (the contract's 'weak' and 'strong' coincide)
            scaling, value = "strong", job_rate
            if world > 1:
                cfg["note_n_gpus"] = ("ONE model, the N = 1 run's table of %d candidates split over %d ranks (%d here): the fit does "
                                      "not shard (replicas only, SURVEY.md 8e), every rank refits and scores its block, the ranks "
                                      "all-gather their arg-best over RCCL; value = job iterations/s, expected ~ 1 / (fit + "
                                      "solve / N).  C4 (10^6-row table split over the ranks) is measured in the same run: "
                                      "c4_sharded" % (M_total, world, M))
                cfg["fit_ms_per_iter_not_scaling"] = round(fit_ms, 3)
        else:
            cfg = {"workload": "C4 (BASELINE.json configs[3]): N=%d, D=%d Matern-5/2 iso, fit (replicated per rank) + posterior "
                               "+ EI over ONE table of %d candidates split over %d rank(s) + device arg-best + RCCL "
                               "all-gather of (best, row) + lowest-index merge" % (N, D, M_total, world),
                   "candidates_total": M_total, "candidates_this_rank": M,
                   "fit_ms_per_iter_not_scaling": round(fit_ms, 3),
                   "predict_ei_ms_per_iter_this_rank": round(ms_per_step - fit_ms, 3),
                   "single_gpu_reference": c4_ref,
                   "speedup_vs_single_gpu": None if not c4_ref else c4_ref["single_gpu_iter_s"] * 1e3 / ms_per_step,
                   "best_row_matches_single_gpu": None if not c4_ref else bool(c4_ref["best_row"] == int(out[1]))}
            scaling, value = "strong", job_rate
        cfg.update({"kernel": kname, "noise": 1e-2, "fit": "replicated on every rank (does not shard, SURVEY.md 8e)",
                    "collective": collective, "rccl_comm_ranks": rccl_ranks, "job_iters_per_s": job_rate,
                    "rccl": {"librccl_mapped": rccl_identity(), "version": h.comm_version()},
                    "torch_imported": "torch" in sys.modules,
                    "launcher": ("bench.py itself (launch_ranks: %d fresh rank processes, no GPU call in the parent)" % world)
                                if os.environ.get("GPHIP_BENCH_LAUNCHED") else
                                ("torch.distributed.run" if world > 1 else "single process"),
                    "ranks_agree_on_winner": ranks_agree if world > 1 else None, "rank_records": rank_records,
                    "emulate_fp64": 0,
                    "lml": out[0], "best_candidate_global_row": int(out[1]), "best_value": float(out[2]),
                    "phases_ms_last_timed_call": phases_timed, "phases_ms": phases,
                    "phases_note": "phases_ms: gp_fit and the predict pass run one after the other AFTER the timed region, the "
                                   "median of three calls each (the per-phase rates below come from it)",
                    "cholesky_tflops": chol["flops"] / chol["ms"] / 1e9,
                    "cholesky_frac_of_fp64_mfma_peak": chol["flops"] / chol["ms"] / 1e9 / FP64_MFMA_PEAK_TFLOPS,
                    "cand_solve_tflops": solve["flops"] / solve["ms"] / 1e9,
                    # the same phase counting what it executes: N^2 M plus the products with the inverted diagonal panels
                    # (M N x panel width; 768 columns by default) -- the rate to hold against the GEMM kernel's own
                    "cand_solve_executed_tflops": solve["flops"] * (1.0 + panel_cols / N) / solve["ms"] / 1e9})
        result = {
            "metric": "GP fit+predict iters/sec at N=%d D=%d" % (N, D),
            "value": value, "unit": "fit+predict iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": cfg,
            "roofline": {"bound": "mfma", "kernel": GEMM_SYMBOL + " (C -= A B^T on fp64 v_mfma_f64_16x16x4_f64, 8 waves)",
                         # flops of the launches / the time the chip actually spent on them: the UNION of the launches' [start, end]
                         # intervals (HIP events on the stream each launch runs on).  In gp_fit_predict ~25 launches per step overlap
                         # (candidate updates beside the factorisation's trailing updates), so the SUM of their durations counts
                         # shared time twice and exceeds the step itself: that figure is kept as *_per_launch.
                         "achieved": achieved_busy, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_busy / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_unit": "bytes per launch",
                         "traffic_provenance": traffic_src,
                         "busy_ms_total": busy_ms,
                         "achieved_per_launch": achieved, "frac_per_launch": achieved / FP64_MFMA_PEAK_TFLOPS,
                         "launches": gs["launches"], "kernel_ms_total": gs["ms"],
                         "avg_launch_ms": gs["ms"] / max(gs["launches"], 1),
                         "flops_per_launch_avg": gs["flops"] / max(gs["launches"], 1),
                         "launch_filter": "every launch of that kernel symbol in the timed region (launches of >= 1400 output "
                                          "tiles: trailing updates, candidate updates; > 90 %% of the flops), %.0f per step"
                                          % (gs["launches"] / max(args.steps, 1)),
                         # the same symbol with no launch of its own kind beside it: gp_fit then gp_predict, 3 un-timed passes
                         "separate_calls_reference": None if sep is None else {
                             "launches": sep["launches"], "kernel_ms_total": sep["ms"],
                             "avg_launch_ms": sep["ms"] / max(sep["launches"], 1),
                             "flops_per_launch_avg": sep["flops"] / max(sep["launches"], 1),
                             "achieved": sep["flops"] / max(sep["ms"], 1e-9) / 1e9,
                             "frac": sep["flops"] / max(sep["ms"], 1e-9) / 1e9 / FP64_MFMA_PEAK_TFLOPS,
                             "busy_ms_total": sep["busy_ms"],
                             "rocprof": "profiles/r05_bench_kernel_stats_fp64.csv: rocprofv3 --kernel-trace --stats of "
                                        "`bench.py --separate-calls` (every launch of the symbol un-overlapped)"},
                         # the whole step against the same peak: N^3/3 (Cholesky) + N^2 M (candidate solve) algorithmic flops of
                         # ONE rank's step / ms_per_step -- K builds, chain, reductions and launch gaps all inside the time
                         "step_algorithmic_flops": N ** 3 / 3.0 + float(N) * N * M,
                         "step_achieved": (N ** 3 / 3.0 + float(N) * N * M) / (ms_per_step * 1e-3) / 1e12,
                         "step_frac": (N ** 3 / 3.0 + float(N) * N * M) / (ms_per_step * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS},
            "chain_gemm": {"bound": "mfma", "kernel": "gemm_nt_kernel<1, 64, 2, false, 64> (the same C -= A B^T as 64 x 64 work "
                                                      "units: the factorisation chain's in-panel / look-ahead updates)",
                           "launches_per_step": cs["launches"], "kernel_ms_per_step": cs["ms"],
                           "avg_launch_ms": cs["ms"] / max(cs["launches"], 1),
                           "flops_per_launch_avg": cs["flops"] / max(cs["launches"], 1),
                           "achieved": cs["flops"] / max(cs["ms"], 1e-9) / 1e9, "peak": FP64_MFMA_PEAK_TFLOPS,
                           "unit": "TFLOP/s", "frac": cs["flops"] / max(cs["ms"], 1e-9) / 1e9 / FP64_MFMA_PEAK_TFLOPS,
                           "note": "ONE extra un-timed step with HIP events around every launch of this symbol (latency-bound "
                                   "launches of the chain: their rate is set by launch + drain time, not by the matrix pipe)"},
            "kbuild": {"bound": "hbm", "kernel": "kbuild_kernel (+ set_rhs_kernel: the 'kbuild' phase of gp_fit, HIP events)",
                       "algorithmic_bytes": kb["bytes"], "ms": kb["ms"], "achieved": kb_gbs, "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": kb_gbs / HBM_PEAK_GBS,
                       "counter_evidence": "profiles/r04_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"},
        }
        if emulated is not None:
            result["emulated_fp64_second_line"] = emulated
        if c4_sec is not None:
            result["c4_sharded"] = c4_sec
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(N, D, 10000, sample_only=args.cpu_baseline_sample)
        print(json.dumps(result))
        sys.stdout.flush()
    if rdv is not None:
        rdv.barrier()
        rdv.close()
        signal.alarm(0)
    h.close()
    if not ranks_agree:
        sys.stderr.write("rank %d: the ranks disagree on the winner: %s\n" % (rank, rank_records))
        sys.exit(3)


if __name__ == "__main__":
    main()
