"""A few one-row gp_acq_grad calls at small N for a kernel trace (test tooling): which launches a call makes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
