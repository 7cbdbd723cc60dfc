#!/usr/bin/env python3
"""gpbench.py -- the measurement scripts of tools/ as sub-commands of one file (test tooling: nothing here is imported by the
package, nothing here touches oracle/).

    python3 tools/gpbench.py <command> [arguments of that command]
    python3 tools/gpbench.py --list

Each command is the former stand-alone script of that name (tools/<command>.py until round 5), body unchanged: it reads its
arguments from sys.argv as before.  Trace post-processors (trace_*.py), pmc_traffic.py, gemm_trace.py, rns_model.py and overlap.py stay
files of their own.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

COMMANDS = {}


def command(fn):
    COMMANDS[fn.__name__] = fn
    return fn

@command
def fit_once():
    """Two fits at the headline size for kernel-trace timelines (test tooling)."""
    import sys, os
    import numpy as np
    from gaussian_process_optimization_amd import _lib
    N, D = 16384, 8
    rng = np.random.default_rng(1234)
    X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1))
    h = _lib.Handle(0)
    for k, v in [a.split("=") for a in sys.argv[1:]]:
        h.set_option(k, int(v))
    h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25*np.sqrt(D)], 1e-2)
    h.fit(); print(h.fit()); print(h.phases())
    h.close()

@command
def fused_once():
    """Two gp_fit_predict calls at the headline size for kernel-trace timelines (test tooling)."""
    import sys, os
    import numpy as np
    from gaussian_process_optimization_amd import _lib
    N, D, M = 16384, 8, 10000
    rng = np.random.default_rng(1234)
    X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1)); Xs = rng.uniform(0, 1, (M, D))
    h = _lib.Handle(0)
    for k, v in [a.split("=") for a in sys.argv[1:]]:
        h.set_option(k, int(v))
    h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25*np.sqrt(D)], 1e-2); h.set_candidates(Xs)
    h.fit_predict(True); h.fit_predict(True); print(h.phases())
    h.close()

@command
def emul_once():
    """Three emulated predicts at C3 (for rocprofv3 --kernel-trace --stats)."""
    import sys, os
    import numpy as np
    from gaussian_process_optimization_amd import _lib
    import bench
    N, D, M = 16384, 8, 10000
    X, Y, Xs = bench.synthetic(N, D, M)
    h = _lib.Handle(0)
    h.set_option("emulate_fp64", 1)
    for kv in sys.argv[1:]:
        k, v = kv.split("="); h.set_option(k, int(v))
    h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25 * np.sqrt(D)], 1e-2); h.set_candidates(Xs)
    h.fit()
    for _ in range(4):
        h.predict(True)
    h.close()

@command
def kbuild_timing():
    """kbuild phase of gp_fit at the headline size, median of several fits (test tooling)."""
    import sys, os
    import numpy as np
    from gaussian_process_optimization_amd import _lib
    h = _lib.Handle(0)
    for N, D, kern in ((16384, 8, 0), (16384, 8, 1), (32768, 16, 0)):
        rng = np.random.default_rng(1)
        X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1))
        h.set_data(X, Y); h.set_params(kern, 0, 1.0, [0.25 * np.sqrt(D)], 1e-2)
        ts = []
        for _ in range(6):
            h.fit(); p = {q["name"]: q for q in h.phases()}["kbuild"]; ts.append(p["ms"])
        ms = float(np.median(ts[1:]))
        print("N=%d D=%d kernel=%d kbuild %.3f ms = %.2f TB/s written (lower triangle)" % (N, D, kern, ms, p["bytes"] / ms / 1e9), flush=True)
    h.close()

@command
def check_fused():
    """gp_fit_predict with a partial pipeline == gp_fit + gp_predict (test tooling)."""
    import sys, os
    import numpy as np
    from gaussian_process_optimization_amd import _lib
    N, D, M = int(os.environ.get("N", 16384)), 8, int(os.environ.get("M", 10000))
    rng = np.random.default_rng(1234)
    X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1)); Xs = rng.uniform(0, 1, (M, D))
    h = _lib.Handle(0)
    h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25*np.sqrt(D)], 1e-2); h.set_candidates(Xs)
    l0 = h.fit(); m0, v0 = h.predict(True)
    for k, v in [a.split("=") for a in sys.argv[1:]]:
        h.set_option(k, int(v))
    (l1), m1, v1 = h.fit_predict(True)
    print("lml equal", l0 == l1, "max|dmu|", np.max(np.abs(m0 - m1)), "max|dvar|", np.max(np.abs(v0 - v1)), "bitwise", np.array_equal(m0, m1) and np.array_equal(v0, v1))
    print({p["name"]: round(p["ms"], 2) for p in h.phases()})
    h.close()

@command
def potri_timing():
    """LML + gradient evaluation timing at C5-like sizes for option settings (test tooling)."""
    import sys, os, time
    import numpy as np
    from gaussian_process_optimization_amd import _lib
    N, D = int(os.environ.get("N", 32768)), 16
    rng = np.random.default_rng(1234)
    X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1))
    h = _lib.Handle(0)
    h.set_data(X, Y); h.set_params(0, 1, 1.0, 0.2 + 0.04 * np.arange(D), 1e-2)
    for k, v in [a.split("=") for a in sys.argv[1:]]:
        h.set_option(k, int(v))
    ref = None
    for rep in range(3):
        t0 = time.perf_counter(); l = h.fit(); t1 = time.perf_counter(); g = h.lml_grad(D); t2 = time.perf_counter()
        ph = {p["name"]: round(p["ms"], 1) for p in h.phases()}
    print(sys.argv[1:], "fit %.1f ms  grad %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), ph, "dv %.10g dn %.10g dl0 %.10g" % (g[0], g[2], g[1][0]), flush=True)
    h.close()

@command
def fused_n():
    """gp_fit_predict vs two calls at another N, over (pipe_stages, pipe_start_pct) pairs given as s:p (test tooling)."""
    import sys, os, time
    import numpy as np
    from gaussian_process_optimization_amd import _lib
    N, D, M = int(os.environ.get("N", 32768)), 8, int(os.environ.get("M", 10000))
    rng = np.random.default_rng(1234)
    X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1)); Xs = rng.uniform(0, 1, (M, D))
    h = _lib.Handle(0)
    h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25*np.sqrt(D)], 1e-2); h.set_candidates(Xs)
    def t(fn, n=5):
        fn(); h.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        h.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    sep = t(lambda: (h.fit(), h.predict(True)))
    for a in ["0:40"] + sys.argv[1:]:
        st, pc = [int(x) for x in a.split(":")]
        h.set_option("pipe_stages", st); h.set_option("pipe_start_pct", pc)
        print("N=%d M=%d pipe_stages=%d start=%d%%: fused %.2f ms  (separate %.2f ms)" % (N, M, st, pc, t(lambda: h.fit_predict(True)), sep), flush=True)
    h.close()

@command
def grad_timing():
    """Predictive-gradient / full-covariance timing at C3-like sizes (test tooling)."""
    import sys, os, time
    import numpy as np
    from gaussian_process_optimization_amd import _lib
    N, D, M = 16384, 8, int(os.environ.get("M", 10000))
    rng = np.random.default_rng(1234)
    X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1)); Xs = rng.uniform(0, 1, (M, D))
    h = _lib.Handle(0)
    for k, v in [a.split("=") for a in sys.argv[1:]]:
        h.set_option(k, int(v))
    h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25*np.sqrt(D)], 1e-2); h.set_candidates(Xs)
    h.fit(); h.predict_grad()
    t0 = time.perf_counter(); h.predict_grad(); t1 = time.perf_counter()
    print("predict_grad M=%d: %.1f ms" % (M, (t1 - t0) * 1e3), {p["name"]: round(p["ms"], 2) for p in h.phases()}, "beta product %.1f TFLOP/s if it were all of it" % (2.0 * M * N * N / ((t1 - t0)) / 1e12))
    Ms = 4096
    h.set_candidates(Xs[:Ms]); h.predict_full_cov(True)
    t0 = time.perf_counter(); h.predict_full_cov(True); t1 = time.perf_counter()
    print("predict_full_cov M=%d: %.1f ms" % (Ms, (t1 - t0) * 1e3), {p["name"]: round(p["ms"], 2) for p in h.phases()})
    h.close()

@command
def supertile_ab():
    """Candidate solve (C3) against the super-tile edge of the long GEMM launches, alternating in one process (test tooling)."""
    import sys, os, time
    import numpy as np
    from gaussian_process_optimization_amd import _lib
    import bench
    N, D, M = 16384, 8, 10000
    X, Y, Xs = bench.synthetic(N, D, M)
    h = _lib.Handle(0)
    h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25 * np.sqrt(D)], 1e-2); h.set_candidates(Xs)
    h.fit()
    vals = [int(v) for v in sys.argv[1:]] or [8, 0, 4, 16, 8, 0, 4, 16]
    for st in vals:
        h.set_option("supertile", st)
        h.predict(True); h.synchronize(); t0 = time.perf_counter()
        for _ in range(4):
            h.predict(True)
        h.synchronize(); dt = (time.perf_counter() - t0) / 4 * 1e3
        print("supertile=%2d  predict %.2f ms (%.1f TFLOP/s)" % (st, dt, float(N) * N * M / dt / 1e9), flush=True)
    h.close()

@command
def panel_width_n():
    """gp_fit / gp_fit_predict / gp_fit_grad against the panel width at several N (test tooling). usage: panel_width_n.py N[,N..] W[,W..]"""
    import sys, os, time
    import numpy as np
    from gaussian_process_optimization_amd import _lib
    Ns = [int(v) for v in sys.argv[1].split(",")]; Ws = [int(v) for v in sys.argv[2].split(",")]
    D, M = 8, 10000
    h = _lib.Handle(0)
    def t(fn, n=4):
        fn(); fn(); h.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        h.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    for N in Ns:
        rng = np.random.default_rng(N)
        X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1)); Xs = rng.uniform(0, 1, (M, D))
        for W in Ws:
            h.set_option("panel_tiles", W)
            h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25 * np.sqrt(D)], 1e-2); h.set_candidates(Xs)
            print("N=%5d W=%2d  fit %.2f  fit_predict %.2f  fit+predict %.2f  fit_grad %.2f ms" % (
                N, W, t(lambda: h.fit()), t(lambda: h.fit_predict(True)), t(lambda: (h.fit(), h.predict(True))), t(lambda: h.fit_grad(1), 2)), flush=True)
    h.close()

@command
def small_fit():
    """Steady-state gp_fit / gp_fit_grad latency at the sizes a BO loop actually has (test tooling)."""
    import sys, os, time
    import numpy as np
    from gaussian_process_optimization_amd import _lib
    h = _lib.Handle(0)
    for k, v in [a.split("=") for a in sys.argv[1:]]:
        h.set_option(k, int(v))
    for N in (64, 128, 256, 512, 1024, 2048, 4096, 8192):
        D = 6
        rng = np.random.default_rng(N)
        X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1))
        h.set_data(X, Y); h.set_params(1, 1, 1.0, 0.5 + 0.05 * np.arange(D), 1e-2)
        h.fit(); h.fit_grad(D)
        t0 = time.perf_counter()
        for _ in range(10): h.fit()
        tf = (time.perf_counter() - t0) / 10 * 1e3
        ph = {p["name"]: round(p["ms"], 3) for p in h.phases()}
        t0 = time.perf_counter()
        for _ in range(10): h.fit_grad(D)
        tg = (time.perf_counter() - t0) / 10 * 1e3
        print("N=%5d: fit %.3f ms  fit_grad %.3f ms" % (N, tf, tg), ph, flush=True)
    h.close()

@command
def small_n_routes():
    """Small N: gp_fit and gp_fit_predict with the look-ahead factorisation vs the single-stream one (option lookahead), M = 2000 (test tooling)."""
    import sys, os, time
    import numpy as np
    from gaussian_process_optimization_amd import _lib
    D, M = 8, int(os.environ.get("M", 2000))
    h = _lib.Handle(0)
    def t(fn, n=8):
        fn(); fn(); h.synchronize(); ts = []
        for _ in range(n):
            t0 = time.perf_counter(); fn(); h.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        return sorted(ts)[len(ts) // 2]
    for N in [int(a) for a in sys.argv[1:]] or [1024, 2048, 3072, 4096, 5120, 6144, 8192]:
        rng = np.random.default_rng(N)
        X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1)); Xs = rng.uniform(0, 1, (M, D))
        h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25 * np.sqrt(D)], 1e-2); h.set_candidates(Xs)
        out = []
        for la in (1, 0):
            h.set_option("lookahead", la)
            out.append((la, t(lambda: h.fit()), t(lambda: h.fit_predict(True)), t(lambda: (h.fit(), h.predict(True)))))
        print("N=%5d  " % N + "   ".join("lookahead=%d: fit %.2f  fit_predict %.2f  two calls %.2f ms" % o for o in out), flush=True)
    h.set_option("lookahead", 1)
    h.close()

@command
def step_breakdown():
    """Wall time of each call of one bench step at C3 (test tooling)."""
    import sys, os, time
    import numpy as np
    from gaussian_process_optimization_amd import _lib
    N, D, M = 16384, 8, 10000
    rng = np.random.default_rng(1234)
    X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1)); Xs = rng.uniform(0, 1, (M, D))
    h = _lib.Handle(0)
    for k, v in [a.split("=") for a in sys.argv[1:]]:
        h.set_option(k, int(v))
    h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25*np.sqrt(D)], 1e-2); h.set_candidates(Xs)
    acc = {}
    def tm(name, fn):
        t0 = time.perf_counter(); r = fn(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; return r
    for it in range(7):
        if it == 2: acc.clear()
        tm("fit", h.fit); tm("predict", lambda: h.predict(True)); f = tm("fmin", h.fmin)
        tm("argbest", lambda: h.acq_argbest(_lib.GP_ACQ_EI, 0.01, f, -1))
    print({k: round(v / 5 * 1e3, 3) for k, v in acc.items()}, "total %.2f ms" % (sum(acc.values()) / 5 * 1e3))
    h.fit(); print({p["name"]: round(p["ms"], 3) for p in h.phases()})
    h.predict(True); print({p["name"]: round(p["ms"], 3) for p in h.phases()})
    h.close()

@command
def fitgrad_sweep():
    """gp_fit_grad vs gp_fit + gp_lml_grad: wall time and equality, for pipeline settings (test tooling)."""
    import sys, os, time
    import numpy as np
    from gaussian_process_optimization_amd import _lib
    N, D = int(os.environ.get("N", 16384)), 8
    rng = np.random.default_rng(1234)
    X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1))
    h = _lib.Handle(0)
    h.set_data(X, Y); h.set_params(0, 1, 1.0, 0.5 + 0.05 * np.arange(D), 1e-2)
    for k, v in [a.split("=") for a in sys.argv[1:]]:
        h.set_option(k, int(v))
    def sep():
        l = h.fit(); return l, h.lml_grad(D)
    def fus():
        return h.fit_grad(D)
    def tm(fn, n=4):
        fn(); h.synchronize(); t0 = time.perf_counter()
        for _ in range(n): r = fn()
        return (time.perf_counter() - t0) / n * 1e3, r
    ts, rs = tm(sep); tf, rf = tm(fus)
    same = rs[0] == rf[0] and rs[1][0] == rf[1][0] and np.array_equal(rs[1][1], rf[1][1]) and rs[1][2] == rf[1][2]
    print(sys.argv[1:], "separate %.2f ms  fused %.2f ms  bitwise-equal %s" % (ts, tf, same), {p["name"]: round(p["ms"], 2) for p in h.phases()}, flush=True)
    h.close()

@command
def fused_sweep():
    """Wall time of gp_fit_predict and of gp_fit + gp_predict (C3) for the GPHIP_RESERVE_CUS given in the environment."""
    import sys, os, time
    import numpy as np
    from gaussian_process_optimization_amd import _lib
    N, D, M = 16384, 8, 10000
    rng = np.random.default_rng(1234)
    X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1)); Xs = rng.uniform(0, 1, (M, D))
    h = _lib.Handle(0)
    for k, v in [a.split("=") for a in sys.argv[1:]]:
        h.set_option(k, int(v))
    h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25*np.sqrt(D)], 1e-2); h.set_candidates(Xs)
    def t(fn, n=5):
        fn(); fn(); h.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): fn()
        h.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    fused = t(lambda: h.fit_predict(True))
    ph = {p["name"]: round(p["ms"], 2) for p in h.phases()}
    sep = t(lambda: (h.fit(), h.predict(True)))
    fit = t(lambda: h.fit())
    print("reserve=%s fused %.2f ms  separate %.2f ms  fit %.2f ms  %s" % (os.environ.get("GPHIP_RESERVE_CUS", "default"), fused, sep, fit, ph), flush=True)
    h.close()

@command
def load_loop():
    """A few seconds of one phase of the bench step in a loop, for sampling clocks / power beside it (test tooling).
    usage: load_loop.py predict|fit|fused|emulated [seconds]"""
    import sys, os, time
    import numpy as np
    from gaussian_process_optimization_amd import _lib
    import bench
    what = sys.argv[1] if len(sys.argv) > 1 else "predict"
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
    N, D, M = 16384, 8, 10000
    X, Y, Xs = bench.synthetic(N, D, M)
    h = _lib.Handle(0)
    h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25 * np.sqrt(D)], 1e-2); h.set_candidates(Xs)
    if what == "emulated": h.set_option("emulate_fp64", 1)
    h.fit(); h.predict(True)
    fn = {"predict": lambda: h.predict(True), "fit": lambda: h.fit(), "fused": lambda: h.fit_predict(True),
          "emulated": lambda: (h.fit(), h.predict(True))}[what]
    print("start %s %.3f" % (what, time.time()), flush=True)
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        fn(); n += 1
    h.synchronize()
    print("end %s %.3f  %d calls, %.2f ms each" % (what, time.time(), n, (time.time() - t0) / n * 1e3), flush=True)
    h.close()

@command
def emul_timing():
    """Candidate solve at C3 (N=16384, D=8, M=10^4): true fp64 vs the int8 residue prototype ("emulate_fp64")."""
    import sys, os, time
    import numpy as np
    from gaussian_process_optimization_amd import _lib
    import bench
    N, D, M = 16384, 8, 10000
    X, Y, Xs = bench.synthetic(N, D, M)
    h = _lib.Handle(0)
    for kv in sys.argv[1:]:
        k, v = kv.split("="); h.set_option(k, int(v))
    h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25 * np.sqrt(D)], 1e-2); h.set_candidates(Xs)
    h.fit()
    for emu, variant, dbg in ((0, 8, 1), (1, 1, 1), (1, 2, 1), (1, 4, 1), (1, 8, 0), (1, 8, 1), (0, 8, 1), (1, 8, 1)):
        h.set_option("emulate_fp64", emu)
        h.set_option("rns_group", variant); h.set_option("rns_interleave", dbg)
        h.predict(True)
        h.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            mu, var = h.predict(True)
        h.synchronize(); dt = (time.perf_counter() - t0) / 3 * 1e3
        ph = {p["name"]: round(p["ms"], 3) for p in h.phases()}
        print("emulate_fp64=%d group %d interleave %d  predict %.2f ms  (%.1f TFLOP/s fp64-equivalent)  phases %s" % (emu, variant, dbg, dt, float(N) * N * M / dt / 1e9, ph))
        if emu == 0: ref = (mu.copy(), var.copy())
        else: print("   max |mean diff| %.2e   max rel var diff %.2e" % (np.max(np.abs(mu - ref[0])), np.max(np.abs(var - ref[1]) / ref[1])))
    h.close()

@command
def fit_sweep():
    """Wall time of gp_fit at C3 for option settings given as k=v[,v2...] (test tooling)."""
    import sys, os, time, itertools
    import numpy as np
    from gaussian_process_optimization_amd import _lib
    N, D = int(os.environ.get("N", 16384)), 8
    rng = np.random.default_rng(1234)
    X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1))
    h = _lib.Handle(0)
    h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25*np.sqrt(D)], 1e-2)
    keys = [a.split("=")[0] for a in sys.argv[1:]]
    vals = [[int(v) for v in a.split("=")[1].split(",")] for a in sys.argv[1:]]
    ref = None
    for combo in itertools.product(*vals):
        for k, v in zip(keys, combo):
            h.set_option(k, v)
        h.fit(); h.fit(); h.synchronize()
        ts = []
        for _ in range(12):
            t0 = time.perf_counter()
            out = h.fit()
            ts.append((time.perf_counter() - t0) * 1e3)
        ms = sorted(ts)[len(ts) // 2]
        chol = [p for p in h.phases() if p["name"] == "cholesky"][0]
        if ref is None: ref = out[0]
        print(dict(zip(keys, combo)), "fit median %.2f ms  chol %.2f ms %.1f TF  lml rel diff %.1e" % (ms, chol["ms"], chol["flops"] / chol["ms"] / 1e9, abs(out[0] - ref) / abs(ref)), flush=True)
    h.close()

@command
def pair_timing():
    """Candidate solve with one vs two panels per update launch ("pair_panels"), C3; bitwise equality of the results."""
    import sys, os, time
    import numpy as np
    from gaussian_process_optimization_amd import _lib
    import bench
    N, D, M = 16384, 8, 10000
    X, Y, Xs = bench.synthetic(N, D, M)
    h = _lib.Handle(0)
    h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25 * np.sqrt(D)], 1e-2); h.set_candidates(Xs)
    h.fit()
    ref = None
    for pair in (0, 1, 0, 1):
        h.set_option("pair_panels", pair)
        h.predict(True)
        h.synchronize(); t0 = time.perf_counter()
        for _ in range(4):
            mu, var = h.predict(True)
        h.synchronize(); dt = (time.perf_counter() - t0) / 4 * 1e3
        if ref is None: ref = (mu.copy(), var.copy())
        t0 = time.perf_counter()
        for _ in range(4):
            h.fit_predict(True)
        h.synchronize(); df = (time.perf_counter() - t0) / 4 * 1e3
        print("pair_panels=%d  predict %.2f ms (%.1f TFLOP/s)  fit_predict %.2f ms   bitwise equal to unpaired: %s"
              % (pair, dt, float(N) * N * M / dt / 1e9, df, np.array_equal(mu, ref[0]) and np.array_equal(var, ref[1])))
    h.close()

@command
def small_calls():
    """Latency of the one-row calls the acquisition optimiser makes (predict, predict_grad, acq_grad at M = 1) (test tooling)."""
    import sys, os, time
    import numpy as np
    from gaussian_process_optimization_amd import _lib
    h = _lib.Handle(0)
    for N in (512, 2048, 16384):
        D = 8
        rng = np.random.default_rng(1)
        X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1))
        h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25 * np.sqrt(D)], 1e-2)
        t0 = time.perf_counter(); h.fit(); tf = (time.perf_counter() - t0) * 1e3
        for M in (1, 5, 1000):
            Xs = rng.uniform(0, 1, (M, D))
            def one():
                h.set_candidates(Xs); return h.predict(True)
            def grad():
                h.set_candidates(Xs); return h.acq_grad(_lib.GP_ACQ_EI, 0.01, 0.0)
            grad(); one()
            t0 = time.perf_counter()
            for _ in range(20): one()
            tp = (time.perf_counter() - t0) / 20 * 1e3
            t0 = time.perf_counter()
            for _ in range(20): grad()
            tg = (time.perf_counter() - t0) / 20 * 1e3
            print("N=%5d M=%4d: fit %.2f ms  set_candidates+predict %.3f ms  set_candidates+acq_grad %.3f ms" % (N, M, tf, tp, tg), {p["name"]: round(p["ms"], 3) for p in h.phases()}, flush=True)
    h.close()

@command
def bo_iteration_timing():
    """One BO iteration's acquisition optimisation (anchor scoring + L-BFGS-B from the 5 best anchors: GPyOpt/GPyOpt/optimization/
    acquisition_optimizer.py:46-77) on a fitted model, with and without the small-M path of the one-row calls (test tooling)."""
    import sys, os, time
    import numpy as np
    import gaussian_process_optimization_amd as gpo

    for N in (500, 4000, 16384):
        D = 8
        rng = np.random.default_rng(3)
        X = rng.uniform(0, 1, (N, D))
        Y = np.sin(2 * np.pi * X).sum(1, keepdims=True) / np.sqrt(D) + 0.05 * rng.standard_normal((N, 1))
        dom = [{'name': 'x', 'type': 'continuous', 'domain': (0, 1), 'dimensionality': D}]
        bo = gpo.methods.BayesianOptimization(f=None, domain=dom, X=X, Y=Y, model_type='GP', acquisition_type='EI', normalize_Y=True,
                                              kernel=gpo.kern.RBF(D, 1.0, 0.25 * np.sqrt(D)), noise_var=1e-2, max_iters=0)
        for small in (8, 0):
            np.random.seed(1)
            bo.suggest_next_locations()                        # fits the model, warms every buffer
            bo.model.model._h.set_option("small_m", small)
            calls = {"n": 0}
            orig = bo.acquisition.acquisition_function_withGradients
            def counted(x, _o=orig):
                calls["n"] += 1
                return _o(x)
            bo.acquisition.acquisition_function_withGradients = counted
            np.random.seed(1)
            t0 = time.perf_counter(); xn = bo.suggest_next_locations(); dt = time.perf_counter() - t0
            bo.acquisition.acquisition_function_withGradients = orig
            print("N=%5d small_m=%d: suggest_next_locations %.1f ms (%d gradient calls of the acquisition optimiser)" % (N, small, dt * 1e3, calls["n"]), flush=True)
        bo.model.model.close()

@command
def bo_iteration_profile():
    """cProfile of one BO iteration's suggest_next_locations at N = 500 (test tooling): where the host time goes."""
    import sys, os, time, cProfile, pstats
    import numpy as np
    import gaussian_process_optimization_amd as gpo
    N, D = 500, 8
    rng = np.random.default_rng(3)
    X = rng.uniform(0, 1, (N, D))
    Y = np.sin(2 * np.pi * X).sum(1, keepdims=True) / np.sqrt(D) + 0.05 * rng.standard_normal((N, 1))
    dom = [{'name': 'x', 'type': 'continuous', 'domain': (0, 1), 'dimensionality': D}]
    bo = gpo.methods.BayesianOptimization(f=None, domain=dom, X=X, Y=Y, model_type='GP', acquisition_type='EI', normalize_Y=True,
                                          kernel=gpo.kern.RBF(D, 1.0, 0.25 * np.sqrt(D)), noise_var=1e-2, max_iters=0)
    np.random.seed(1); bo.suggest_next_locations()
    np.random.seed(1)
    t0 = time.perf_counter(); bo.suggest_next_locations(); print("wall %.2f ms" % ((time.perf_counter() - t0) * 1e3))
    np.random.seed(1)
    pr = cProfile.Profile(); pr.enable(); bo.suggest_next_locations(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)

@command
def one_row_trace():
    """A few one-row gp_acq_grad calls at small N for a kernel trace (test tooling): which launches a call makes."""
    import sys, os, time
    import numpy as np
    from gaussian_process_optimization_amd import _lib
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    D = 8
    rng = np.random.default_rng(1)
    X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1))
    h = _lib.Handle(0)
    h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.7], 1e-2); h.fit(); fmin = h.fmin()
    Xs = rng.uniform(0, 1, (1, D))
    h.set_candidates(Xs); h.acq_grad(0, 0.01, fmin)
    t0 = time.perf_counter()
    for _ in range(200):
        h.set_candidates(Xs); h.acq_grad(0, 0.01, fmin)
    print("N=%d one-row set_candidates + acq_grad: %.1f us per call" % (N, (time.perf_counter() - t0) / 200 * 1e6))
    t0 = time.perf_counter()
    for _ in range(200):
        h.set_candidates(Xs)
    print("   set_candidates alone: %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
    t0 = time.perf_counter()
    for _ in range(200):
        h.predict(True)
    print("   predict alone: %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
    t0 = time.perf_counter()
    for _ in range(200):
        h.predict_grad()
    print("   predict_grad alone: %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
    h.close()

@command
def rows_trace():
    """A burst of one-location gradient calls through gp_acq_rows (csrc/onerow.hip) for a kernel trace or a wall-clock figure
    (test tooling).  usage: rows_trace.py N [calls]"""
    import sys, os, time
    import numpy as np
    from gaussian_process_optimization_amd import _lib
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    D = 8
    rng = np.random.default_rng(1)
    X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1))
    h = _lib.Handle(0)
    h.set_option("emulate_fp64", 0)
    h.set_option("rows_build", 1)      # steady state: the inverse factor at the first call (by default above N = 4096 only after N / 768 calls)
    for kv in sys.argv[3:]:
        k, v = kv.split("=")
        h.set_option(k, int(v))
    h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.7], 1e-2); h.fit(); fmin = h.fmin()
    Xs = rng.uniform(0, 1, (1, D))
    t0 = time.perf_counter(); h.acq_rows(Xs, 0, 0.01, fmin, grad=True); t_first = time.perf_counter() - t0
    for what, fn in (("acq_rows value + gradient", lambda: h.acq_rows(Xs, 0, 0.01, fmin, grad=True)),
                     ("acq_rows value", lambda: h.acq_rows(Xs, 0, 0.01, fmin)),
                     ("predict_rows", lambda: h.predict_rows(Xs, True)),
                     ("predict_rows + gradients", lambda: h.predict_rows(Xs, True, grad=True))):
        fn()
        t0 = time.perf_counter()
        for _ in range(calls):
            fn()
        dt = (time.perf_counter() - t0) / calls
        lower = 8.0 * N * N / 2
        passes = 2 if "grad" in what else 1
        print("N=%d %-28s %8.1f us per call   (%d x %.2f GB of L^-1 -> %.2f TB/s incl. launch + sync)" % (
            N, what, dt * 1e6, passes, lower / 1e9, passes * lower / dt / 1e12), flush=True)
    print("first gradient call after the fit (builds L^-1): %.2f ms" % (t_first * 1e3))
    h.close()

@command
def ab():
    """A/B of option settings inside ONE process, alternating (test tooling):  ab.py name=v1,v2[,v3] [fixed=val ...] [--predict|--fit|--fused]
    Prints the median wall time of gp_predict / gp_fit / gp_fit_predict at C3 per setting, and whether mean / variance are bitwise the first setting's."""
    import sys, os, time
    import numpy as np
    from gaussian_process_optimization_amd import _lib
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    modes = [a[2:] for a in sys.argv[1:] if a.startswith("--")] or ["predict", "fused"]
    sweep = [a for a in args if "," in a][0]
    fixed = [a for a in args if "," not in a]
    name, vals = sweep.split("=")[0], [int(v) for v in sweep.split("=")[1].split(",")]
    N, D, M = 16384, 8, 10000
    rng = np.random.default_rng(1234)
    X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1)); Xs = rng.uniform(0, 1, (M, D))
    h = _lib.Handle(0)
    for k, v in [a.split("=") for a in fixed]:
        h.set_option(k, int(v))
    h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25 * np.sqrt(D)], 1e-2); h.set_candidates(Xs)
    h.fit()
    fn = {"predict": lambda: h.predict(True), "fit": lambda: h.fit(), "fused": lambda: h.fit_predict(True)}
    for mode in modes:
        times = {v: [] for v in vals}
        ref = None
        same = {}
        for rep in range(7):
            for v in vals:
                h.set_option(name, v)
                h.synchronize()
                t0 = time.perf_counter(); r = fn[mode](); times[v].append((time.perf_counter() - t0) * 1e3)
                if mode != "fit":
                    mv = r[-2:] if mode == "fused" else r
                    if ref is None: ref = (mv[0].copy(), mv[1].copy())
                    same[v] = bool(np.array_equal(mv[0], ref[0]) and np.array_equal(mv[1], ref[1]))
        print(mode, " ".join("%s=%d: %.2f ms (min %.2f)%s" % (name, v, np.median(times[v][1:]), min(times[v][1:]), "" if same.get(v, True) else " DIFFERENT RESULT") for v in vals), flush=True)
    h.close()

@command
def emul_bench():
    """The bench step with emulate_fp64 on ONLY (gp_fit + gp_predict + EI arg-best at C3), for a per-mode kernel-stats profile:
    rocprofv3 --kernel-trace --stats -- python3 tools/emul_bench.py   (test tooling; bench.py's second line is the measurement)."""
    import os
    import sys
    import time

    import bench
    from gaussian_process_optimization_amd import _lib

    N, D, M = 16384, 8, 10000
    X, Y, Xs = bench.synthetic(N, D, M)
    h = _lib.Handle(0)
    for kv in sys.argv[1:]:
        k, v = kv.split("=")
        h.set_option(k, int(v))
    h.set_option("emulate_fp64", 1)
    h.set_data(X, Y)
    h.set_params(_lib.GP_KERNEL_RBF, 0, 1.0, [0.25 * D ** 0.5], 1e-2)
    h.set_candidates(Xs)


    def step():
        lml = h.fit()[0]
        h.predict(True)
        return lml, h.acq_argbest(_lib.GP_ACQ_EI, 0.01, h.fmin(), -1)


    for _ in range(2):
        step()
    h.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = step()
    h.synchronize()
    print("emulated step %.2f ms" % ((time.perf_counter() - t0) / 10 * 1e3), out)
    h.close()

@command
def emul_fit_timing():
    """gp_fit and the whole bench step at C3: true fp64 vs emulate_fp64 (trailing update + candidate solve on int8 MFMA)."""
    import sys, os, time
    import numpy as np
    from gaussian_process_optimization_amd import _lib
    import bench
    N, D, M = 16384, 8, 10000
    X, Y, Xs = bench.synthetic(N, D, M)
    h = _lib.Handle(0)
    for kv in sys.argv[1:]:
        k, v = kv.split("="); h.set_option(k, int(v))
    h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25 * np.sqrt(D)], 1e-2); h.set_candidates(Xs)
    ref = None
    for emu, efit in ((0, 0), (1, 1), (1, 8), (0, 0), (1, 8)):
        h.set_option("emulate_fp64", emu); h.set_option("emulate_fit", 1 if efit else 0)
        if efit: h.set_option("rns_group_fit", efit)
        h.fit(); h.predict(True)
        h.synchronize(); t0 = time.perf_counter()
        for _ in range(4):
            lml = h.fit()[0]
        h.synchronize(); tf = (time.perf_counter() - t0) / 4 * 1e3
        phf = {p["name"]: round(p["ms"], 3) for p in h.phases()}
        t0 = time.perf_counter()
        for _ in range(4):
            h.fit(); mu, var = h.predict(True); f = h.fmin(); h.acq_argbest(_lib.GP_ACQ_EI, 0.01, f, -1)
        h.synchronize(); ts = (time.perf_counter() - t0) / 4 * 1e3
        h.fit_predict(True)
        h.synchronize(); t0 = time.perf_counter()
        for _ in range(4):
            (lmlf, _, _), muf, varf = h.fit_predict(True); f = h.fmin(); h.acq_argbest(_lib.GP_ACQ_EI, 0.01, f, -1)
        h.synchronize(); tfu = (time.perf_counter() - t0) / 4 * 1e3
        phu = {p["name"]: round(p["ms"], 3) for p in h.phases()}
        print("   fused gp_fit_predict + EI: %.2f ms = %.2f it/s  same as two calls: %s  phases %s" % (tfu, 1e3 / tfu, bool(lmlf == lml and np.array_equal(muf, mu) and np.array_equal(varf, var)), phu))
        if ref is None: ref = (lml, mu.copy(), var.copy())
        print("emulate_fp64=%d rns_group_fit=%d  fit %.2f ms (cholesky %.2f = %.1f TFLOP/s eq)  step(fit+predict+EI) %.2f ms = %.2f it/s   lml rel diff %.1e  var rel diff %.1e"
              % (emu, efit, tf, phf["cholesky"], N ** 3 / 3.0 / phf["cholesky"] / 1e9, ts, 1e3 / ts, abs(lml - ref[0]) / abs(ref[0]),
                 np.max(np.abs(var - ref[2]) / ref[2])))
    h.close()

@command
def inner_sweep():
    """gp_fit / gp_fit_predict / emulated fit against the in-panel step (inner_tiles, inner_min_rows) at several N (test tooling)."""
    import sys, os, time
    import numpy as np
    from gaussian_process_optimization_amd import _lib

    def data(N, D, M, seed=1234):
        rng = np.random.default_rng(seed)
        X = rng.uniform(0, 1, (N, D))
        Y = np.sin(2 * np.pi * X).sum(1, keepdims=True) / np.sqrt(D) + 0.05 * rng.standard_normal((N, 1))
        return X, (Y - Y.mean()) / Y.std(), np.random.default_rng(seed + 2).uniform(0, 1, (M, D))

    def med(fn, n):
        fn(); ts = []
        for _ in range(n):
            t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
        return float(np.median(ts)) * 1e3, r

    settings = [("inner_tiles", 1, 0)] + [("inner_tiles", 2, m) for m in (0, 8, 16, 24, 32, 48, 64, 96)]
    if len(sys.argv) > 1:
        settings = [("inner_tiles", 2, int(a)) if a != "off" else ("inner_tiles", 1, 0) for a in sys.argv[1:]]
    h = _lib.Handle(0)
    for N, D, M in ((4096, 4, 2000), (8192, 8, 5000), (16384, 8, 10000)):
        X, Y, Xs = data(N, D, M)
        h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25 * np.sqrt(D)], 1e-2); h.set_candidates(Xs)
        ref = None
        for _, it, mr in settings:
            h.set_option("inner_tiles", it); h.set_option("inner_min_rows", mr)
            out = []
            for emu in (0, 1):
                h.set_option("emulate_fp64", emu)
                tf, r = med(h.fit, 5 if N >= 16384 else 15)
                ts, _ = med(lambda: h.fit_predict(True), 3 if N >= 16384 else 8)
                out.append((tf, ts, r[0]))
            if ref is None: ref = out[0][2]
            print("N=%5d inner_tiles=%d min_rows=%3d | fp64: fit %7.3f ms  fit_predict %7.3f ms | emulated: fit %7.3f ms  fit+predict %7.3f ms | lml rel %.1e"
                  % (N, it, mr, out[0][0], out[0][1], out[1][0], out[1][1], abs(out[0][2] - ref) / abs(ref)), flush=True)
    h.close()

@command
def diff_sweep():
    """Randomised differential check of the round-4 routes against the default route (test tooling): the pair step of the in-panel
    factorisation (inner_tiles = 2, any panel width, both schedulers) and the small-M path (small_m = 8 vs 0), random shapes.  Prints the
    largest relative differences seen; exits 1 beyond 1e-9."""
    import sys, os
    import numpy as np
    from gaussian_process_optimization_amd import _lib
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    rng = np.random.default_rng(2024)
    h = _lib.Handle(0)
    worst = {"lml": 0.0, "mean": 0.0, "var": 0.0, "grad": 0.0, "dvdx": 0.0}
    for case in range(n_cases):
        N = int(rng.choice([1, 2, 100, 127, 128, 129, 255, 256, 257, 383, 384, 500, 640, 768, 769, 1000, 1500, 2047, 2048, 2600, 3100]))
        D = int(rng.choice([1, 2, 5, 8, 13])); M = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 40, 200]))
        kern = int(rng.integers(2)); ard = int(rng.integers(2)); noise = float(rng.choice([1e-1, 1e-2, 1e-3]))
        X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1)); Xs = rng.uniform(-0.05, 1.05, (M, D))
        ls = rng.uniform(0.3, 1.2, D if ard else 1) * np.sqrt(D) * 0.5
        opts = dict(panel_tiles=int(rng.integers(1, 9)), lookahead_min_tiles=int(rng.choice([0, 4, 40])), emulate_fp64=int(rng.integers(2)))
        res = {}
        for route in ("default", "new"):
            for k, v in opts.items(): h.set_option(k, v)
            h.set_option("inner_tiles", 2 if route == "new" else 1)
            h.set_option("small_m", 8 if route == "new" else 0)
            h.set_data(X, Y); h.set_params(kern, ard, 1.3, ls, noise); h.set_candidates(Xs)
            if rng.integers(2) and route == "new":
                (lml, _, _), mu, var = h.fit_predict(True)
            else:
                lml = h.fit()[0]; mu, var = h.predict(True)
            g = h.lml_grad(ls.size)
            dm, dv = h.predict_grad()
            res[route] = (lml, mu, var, np.r_[g[0], g[1], g[2]], dv)
        a, b = res["default"], res["new"]
        rel = lambda x, y: float(np.max(np.abs(np.asarray(x) - np.asarray(y))) / max(np.max(np.abs(y)), 1e-300))
        d = {"lml": abs(a[0] - b[0]) / max(abs(a[0]), 1.0), "mean": rel(b[1], a[1]), "var": rel(b[2], a[2]), "grad": rel(b[3], a[3]), "dvdx": rel(b[4], a[4])}
        for k in worst: worst[k] = max(worst[k], d[k])
        if max(d.values()) > 1e-9:
            print("case", case, dict(N=N, D=D, M=M, kern=kern, ard=ard, noise=noise, **opts), d, flush=True)
    print("cases", n_cases, "worst relative differences", {k: "%.1e" % v for k, v in worst.items()}, flush=True)
    h.close()
    sys.exit(1 if max(worst.values()) > 1e-9 else 0)

@command
def configs_timing():
    """Wall time of the BASELINE.json configurations other than the bench line (C2, C4 one shard, C5), for DESIGN.md (test tooling)."""
    import sys, os, time
    import numpy as np
    from gaussian_process_optimization_amd import _lib

    def data(N, D, M, seed=1234):
        rng = np.random.default_rng(seed)
        X = rng.uniform(0, 1, (N, D))
        f = np.sin(2 * np.pi * X).sum(1, keepdims=True) / np.sqrt(D)
        Y = f + 0.05 * np.random.default_rng(seed + 1).standard_normal((N, 1)); Y = (Y - Y.mean()) / Y.std()
        return X, Y, np.random.default_rng(seed + 2).uniform(0, 1, (M, D))

    def tm(fn, n=3):
        fn(); t0 = time.perf_counter()
        for _ in range(n): r = fn()
        return (time.perf_counter() - t0) / n * 1e3, r

    h = _lib.Handle(0)
    # C2: N=4096, D=4 RBF: K-build + Cholesky
    X, Y, Xs = data(4096, 4, 8)
    h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.5], 1e-2)
    ms, _ = tm(h.fit, 10); ph = {p["name"]: round(p["ms"], 3) for p in h.phases()}
    print("C2 fit %.2f ms" % ms, ph, "cholesky %.1f TFLOP/s" % (4096**3 / 3 / ph["cholesky"] / 1e9), flush=True)
    # C4, one rank: N=16384, D=8 Matern-5/2, fit once + EI over 125 000 candidates + argbest
    X, Y, Xs = data(16384, 8, 125000)
    h.set_data(X, Y); h.set_params(1, 0, 1.0, [0.25 * np.sqrt(8)], 1e-2)
    msf, _ = tm(h.fit, 3)
    h.set_candidates(Xs)
    def ei():
        h.predict(True)                     # posterior at the resident shard (8 chunks of mc_max candidates)
        f = h.fmin(); return h.acq_argbest(_lib.GP_ACQ_EI, 0.01, f, -1)
    mse, r = tm(ei, 2)
    print("C4 shard: fit %.1f ms, EI + argbest over 125000 candidates %.1f ms (%.1f TFLOP/s in the candidate solve)" % (msf, mse, 16384.0**2 * 125000 / mse / 1e9), r, flush=True)
    # C5: N=32768, D=16 ARD-RBF: LML + (D+2) gradients per evaluation
    X, Y, Xs = data(32768, 16, 8)
    h.set_data(X, Y); h.set_params(0, 1, 1.0, 0.2 + 0.04 * np.arange(16), 1e-2)
    def ev():
        l = h.fit(); return l, h.lml_grad(16)
    ms, r = tm(ev, 3)
    h.fit(); phf = {p["name"]: round(p["ms"], 2) for p in h.phases()}
    h.lml_grad(16); phg = {p["name"]: round(p["ms"], 2) for p in h.phases()}
    print("C5 evaluation (LML + 18 gradients) %.1f ms" % ms, phf, phg, "cholesky %.1f TFLOP/s, potri %.1f TFLOP/s" % (32768.0**3 / 3 / phf["cholesky"] / 1e9, 2 * 32768.0**3 / 3 / sum(v for k, v in phg.items() if k.startswith("potri")) / 1e9), flush=True)
    h.close()

@command
def emul_sweep():
    """Random sweep: emulate_fp64 against the true-fp64 device path on problems far from the bench line (test tooling).

    Kernel, ARD, D, N, panel width, variance over six decades, lengthscales over 2.5 decades, noise from 1e-8 to 1 -- for each
    draw: gp_fit + gp_predict both ways, differences of LML / mean / variance, whether the jitter ladder took the same rung,
    and how often the residue path had to fall back (non-finite data are not drawn here, so that count must stay 0)."""
    import sys, os, json
    import numpy as np
    from gaussian_process_optimization_amd import _lib

    n_draws = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(20260)
    h = _lib.Handle(0)
    rows, fails = [], []
    for it in range(n_draws):
        N = int(rng.integers(130, 3000)); D = int(rng.integers(1, 11)); M = int(rng.integers(1, 400))
        kernel = int(rng.integers(0, 2)); ard = int(rng.integers(0, 2)); pt = int(rng.integers(1, 9))
        variance = float(10 ** rng.uniform(-3, 3)); noise = float(variance * 10 ** rng.uniform(-8, 0))
        ls = (10 ** rng.uniform(-1.5, 1.0, D if ard else 1)).tolist()
        X = rng.uniform(0, 1, (N, D)); Y = np.sin(3 * X.sum(1, keepdims=True)) * np.sqrt(variance) + np.sqrt(noise) * rng.standard_normal((N, 1))
        Xs = rng.uniform(-0.1, 1.1, (M, D))
        h.set_option("panel_tiles", pt)
        out = []
        try:
            for emu in (0, 1):
                h.set_option("emulate_fp64", emu)
                h.set_data(X, Y); h.set_params(kernel, ard, variance, ls, noise); h.set_candidates(Xs)
                f = h.fit(); mu, var = h.predict(True)
                out.append((f, mu.copy(), var.copy()))
        except Exception as e:  # noqa: BLE001
            fails.append(dict(draw=it, N=N, D=D, kernel=kernel, ard=ard, pt=pt, variance=variance, noise=noise, emu=emu, error=str(e)[:200]))
            continue
        (f0, m0, v0), (f1, m1, v1) = out
        rows.append(dict(draw=it, N=N, D=D, M=M, kernel=kernel, ard=ard, pt=pt, variance=variance, noise_over_variance=noise / variance,
                         jitter_fp64=f0[2], jitter_emulated=f1[2],
                         lml_rel=abs(f1[0] - f0[0]) / max(abs(f0[0]), 1e-300),
                         mean_rel=float(np.max(np.abs(m1 - m0)) / max(np.max(np.abs(m0)), 1e-300)),
                         var_rel_to_prior=float(np.max(np.abs(v1 - v0)) / (variance + noise))))
    h.set_option("emulate_fp64", 0)
    h.close()
    def pct(k):
        a = np.array([r[k] for r in rows]); return dict(median=float(np.median(a)), p90=float(np.percentile(a, 90)), max=float(a.max()))
    summary = dict(draws=n_draws, completed=len(rows), errors=fails, same_jitter_rung=int(sum(r["jitter_fp64"] == r["jitter_emulated"] for r in rows)),
                   lml_rel=pct("lml_rel"), mean_rel=pct("mean_rel"), var_rel_to_prior=pct("var_rel_to_prior"),
                   worst=sorted(rows, key=lambda r: -max(r["lml_rel"], r["mean_rel"], r["var_rel_to_prior"]))[:5])
    print(json.dumps(summary, indent=1))
    json.dump(dict(summary=summary, rows=rows), open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "emul_sweep.json"), "w"))

@command
def opt_sweep():
    """Option A/B at several N, alternating inside one process (test tooling):
       opt_sweep.py name=v1,v2[,...] [fixed=val ...] [--sizes 8192,16384,32768]
       opt_sweep.py nameA:nameB=a1:b1,a2:b2 ...          (several options per setting)
    Prints per N the median gp_fit / gp_fit_predict / gp_fit_grad wall time per setting and whether LML, mean and variance are bitwise the
    first setting's."""
    import sys, os, time
    import numpy as np
    from gaussian_process_optimization_amd import _lib

    argv = sys.argv[1:]
    sizes = [8192, 16384, 32768]
    if "--sizes" in argv:
        i = argv.index("--sizes"); sizes = [int(x) for x in argv[i + 1].split(",")]; del argv[i:i + 2]
    sweep = [a for a in argv if "," in a][0]
    fixed = [a.split("=") for a in argv if "," not in a]
    names = sweep.split("=")[0].split(":")
    vals = [tuple(int(x) for x in v.split(":")) for v in sweep.split("=")[1].split(",")]
    name = ":".join(names)
    def apply(v):
        for k, x in zip(names, v): h.set_option(k, x)
    lab = lambda v: ":".join(str(x) for x in v)
    h = _lib.Handle(0)
    for k, v in fixed: h.set_option(k, int(v))
    for N in sizes:
        D, M = 8, 10000 if N >= 16384 else 5000
        rng = np.random.default_rng(1234)
        X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1)); Xs = rng.uniform(0, 1, (M, D))
        h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25 * np.sqrt(D)], 1e-2); h.set_candidates(Xs)
        h.fit()
        reps = 4 if N >= 32768 else 7
        for mode, fn in (("fit", h.fit), ("fit_predict", lambda: h.fit_predict(True)), ("fit_grad", lambda: h.fit_grad(1))):
            times = {v: [] for v in vals}; ref = None; same = {}
            for rep in range(reps):
                for v in vals:
                    apply(v); h.synchronize()
                    t0 = time.perf_counter(); r = fn(); times[v].append((time.perf_counter() - t0) * 1e3)
                    flat = []
                    def walk(x):
                        if isinstance(x, tuple):
                            for y in x: walk(y)
                        else: flat.append(np.asarray(x))
                    walk(r)
                    if ref is None: ref = [x.copy() for x in flat]
                    same[v] = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(flat, ref))
            print("N=%5d %-11s " % (N, mode) + "  ".join("%s=%s: %.2f (min %.2f)%s" % (name, lab(v), np.median(times[v][1:]), min(times[v][1:]),
                  "" if same[v] else " DIFFERENT") for v in vals), flush=True)
    h.close()


@command
def lp_driver_timing():
    """The thesis driver's call (run.py:1206-1258) end to end at fixed hyper-parameters: BayesianOptimization(Gower, local penalisation,
    batch of 5).suggest_next_locations() and the candidate-table loop over 20 000 rows, N = 300 and 4000 (test tooling)."""
    import time
    import numpy as np
    import gaussian_process_optimization_amd as gpo
    for N in (300, 4000):
        D = 6
        rng = np.random.default_rng(3)
        dom = [{'name': 'm', 'type': 'discrete', 'domain': tuple(range(6))}, {'name': 'p', 'type': 'discrete', 'domain': tuple(range(9))},
               {'name': 'q', 'type': 'discrete', 'domain': tuple(range(4))}, {'name': 'r', 'type': 'discrete', 'domain': tuple(range(3))},
               {'name': 'c', 'type': 'continuous', 'domain': (12.0, 48.0)}, {'name': 'l', 'type': 'continuous', 'domain': (25.4, 100.0)}]
        sp = gpo.Design_space(dom)
        np.random.seed(N)                       # the design, the table and estimate_L draw from numpy's global generator
        X = sp.samples_uniform(N)
        Y = (np.sin(X[:, 4] / 7) + 0.3 * np.cos(X[:, 0]) + 0.2 * (X[:, 2] == 1) + 0.01 * (X[:, 5] - 60) ** 2 / 100)[:, None] + 0.02 * rng.standard_normal((N, 1))
        table = sp.samples_uniform(20000)
        for gower in (True, False):
            bo = gpo.methods.BayesianOptimization(f=None, domain=dom, X=X, Y=Y, acquisition_type='EI', normalize_Y=True, exact_feval=True,
                                                  acquisition_optimizer_type='lbfgs', evaluator_type='local_penalization', batch_size=5,
                                                  Gower=gower, noise_var=0, max_iters=0)
            np.random.seed(1); bo.suggest_next_locations()
            np.random.seed(1)
            t0 = time.perf_counter(); bo.suggest_next_locations(); t1 = time.perf_counter() - t0
            np.random.seed(1)
            t0 = time.perf_counter(); rows = bo.evaluator.compute_batch_from_table(table, sense=+1); t2 = time.perf_counter() - t0
            print("N=%d Gower=%s: suggest_next_locations (LP batch of 5, L-BFGS) %.1f ms; table loop over 20000 rows %.1f ms" % (N, gower, t1 * 1e3, t2 * 1e3), flush=True)
            bo.model.model.close()


@command
def rows_fuzz():
    """Random sweep of the one-location entry points against the batched ones: N, D, M, kernel, ARD, Gower, noise, variance drawn at
    random (test tooling).  usage: rows_fuzz [draws]"""
    import json
    import numpy as np
    from gaussian_process_optimization_amd import _lib
    draws = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(2025)
    h = _lib.Handle(0)
    h.set_option("emulate_fp64", 0)
    h.set_option("rows_build", 1)
    worst = {"mean": 0.0, "var": 0.0, "dm": 0.0, "dv": 0.0, "acq": 0.0, "dacq": 0.0}
    bad = []
    for it in range(draws):
        N = int(rng.choice([1, 2, 5, 31, 32, 33, 64, 127, 128, 129, 255, 257, 500, 1023, 1024, 1025, 1500, 2047, 2048, 2049, 2600, 3100]))
        D = int(rng.choice([1, 2, 3, 6, 8, 16, 31, 64]))
        M = int(rng.integers(1, 9))
        kern = int(rng.integers(2)); ard = int(rng.integers(2)); gower = int(rng.integers(3) == 0)
        noise = float(rng.choice([1e-1, 1e-2, 1e-3])); var = float(rng.uniform(0.4, 2.0))
        X = rng.uniform(0, 1, (N, D))
        if gower:
            disc = rng.integers(0, 2, D)
            X[:, disc == 1] = np.round(X[:, disc == 1] * 3)
        Y = rng.standard_normal((N, 1))
        Xs = rng.uniform(-0.1, 1.1, (M, D))
        if gower:
            Xs[:, disc == 1] = np.round(np.clip(Xs[:, disc == 1], 0, 1) * 3)
        if N >= M and rng.integers(2):
            Xs[0] = X[int(rng.integers(N))]
        ls = rng.uniform(0.3, 1.5, D) * np.sqrt(D) * 0.5 if ard else np.array([0.35 * np.sqrt(D)])
        h.set_data(X, Y)
        h.set_params(kern, ard, var, ls, noise)
        if gower:
            h.set_gower(disc.astype(np.int32), np.where(disc == 1, 1.0, 1.2))
        else:
            h.set_gower()
        h.fit()
        fmin = h.fmin()
        h.set_candidates(Xs)
        mu_b, v_b = h.predict(True)
        dm_b, dv_b = h.predict_grad()
        a_b, da_b = h.acq_grad(_lib.GP_ACQ_EI, 0.01, fmin)
        mu, v, dm, dv = h.predict_rows(Xs, True, grad=True)
        a, da = h.acq_rows(Xs, _lib.GP_ACQ_EI, 0.01, fmin, grad=True)
        jm = h.mean_grad_rows(Xs)

        def rel(x, y, floor=1e-300):
            return float(np.max(np.abs(np.asarray(x) - np.asarray(y))) / max(float(np.max(np.abs(y))), floor))
        errs = {"mean": rel(mu, mu_b, 1e-6), "var": float(np.max(np.abs(v - v_b) / np.maximum(np.abs(v_b), 1e-12))),
                "dm": max(rel(dm, dm_b, 1e-6), rel(jm, dm_b, 1e-6)), "dv": rel(dv, dv_b, 1e-6), "acq": rel(a, a_b, 1e-12),
                "dacq": rel(da, da_b, 1e-12)}
        finite = all(np.all(np.isfinite(z)) == np.all(np.isfinite(zb)) for z, zb in ((mu, mu_b), (v, v_b), (dm, dm_b), (dv, dv_b)))
        for k, e in errs.items():
            if np.isfinite(e):
                worst[k] = max(worst[k], e)
        if not finite or errs["mean"] > 1e-8 or errs["var"] > 1e-7 or errs["dm"] > 1e-8 or errs["dv"] > 1e-6 or errs["acq"] > 1e-6 or errs["dacq"] > 1e-5:
            bad.append({"draw": it, "N": N, "D": D, "M": M, "kernel": kern, "ard": ard, "gower": gower, "noise": noise, **errs})
    st = h.rows_stats()
    print(json.dumps({"draws": draws, "worst_relative_difference_to_the_batched_calls": worst, "outside_tolerance": bad,
                      "fused_calls": st["fused"], "fallback_calls": st["fallback"]}, indent=1))
    h.close()


def main():
    if len(sys.argv) < 2 or sys.argv[1] in ("-h", "--help", "--list"):
        print(__doc__)
        for name, fn in COMMANDS.items():
            print("  %-22s %s" % (name, (fn.__doc__ or "").strip().split("\n")[0][:150]))
        return 0
    name = sys.argv[1]
    if name not in COMMANDS:
        print("unknown command %r (try --list)" % name, file=sys.stderr)
        return 2
    sys.argv = ["gpbench.py " + name] + sys.argv[2:]
    COMMANDS[name]()
    return 0


if __name__ == "__main__":
    sys.exit(main())
