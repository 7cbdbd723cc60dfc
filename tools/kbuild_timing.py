"""kbuild phase of gp_fit at the headline size, median of several fits (test tooling)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
