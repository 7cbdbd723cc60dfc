"""Latency of the one-row calls the acquisition optimiser makes (predict, predict_grad, acq_grad at M = 1) (test tooling)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
