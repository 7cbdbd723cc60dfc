"""Steady-state gp_fit / gp_fit_grad latency at the sizes a BO loop actually has (test tooling)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
