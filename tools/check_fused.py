"""gp_fit_predict with a partial pipeline == gp_fit + gp_predict (test tooling)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
