"""Two gp_fit_predict calls at the headline size for kernel-trace timelines (test tooling)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
