"""Wall time of gp_fit_predict and of gp_fit + gp_predict (C3) for the GPHIP_RESERVE_CUS given in the environment."""
import sys, os, time
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
