"""Timing sweep over schedule options at the headline size (test tooling)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussian_process_optimization_amd import _lib

N, D, M = 16384, 8, 10000
rng = np.random.default_rng(1234)
X = rng.uniform(0, 1, (N, D))
Y = np.sin(2*np.pi*X).sum(1, keepdims=True)/np.sqrt(D) + 0.05*rng.standard_normal((N, 1))
Y = (Y - Y.mean())/Y.std()
Xs = rng.uniform(0, 1, (M, D))
h = _lib.Handle(0)
h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25*np.sqrt(D)], 1e-2); h.set_candidates(Xs)
ref = None
for (pt, la, rc) in [(8, 0, 8), (8, 1, 8), (8, 1, 16), (8, 1, 0), (4, 1, 8), (6, 1, 8), (12, 1, 8), (16, 1, 8)]:
    h.set_option("panel_tiles", pt); h.set_option("lookahead", la); h.set_option("reserve_cus", rc)
    h.fit()
    ts = []
    for r in range(3):
        t0 = time.perf_counter(); lml, _, _ = h.fit(); ts.append(time.perf_counter() - t0)
    ph = {p["name"]: p["ms"] for p in h.phases()}
    if ref is None: ref = lml
    print("panel=%2d lookahead=%d reserve=%2d: fit %.2f ms (chol %.2f ms = %.1f TF, alpha %.2f) lml diff %.2e" % (pt, la, rc, min(ts)*1e3, ph["cholesky"], N**3/3/ph["cholesky"]/1e9, ph["alpha_lml"], abs(lml-ref)/abs(ref)))
h.close()
