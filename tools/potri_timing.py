"""LML + gradient evaluation timing at C5-like sizes for option settings (test tooling)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
