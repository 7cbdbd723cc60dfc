"""Predictive-gradient / full-covariance timing at C3-like sizes (test tooling)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
