"""gp_fit_predict vs two calls at another N (test tooling)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussian_process_optimization_amd import _lib
N, D, M = int(os.environ.get("N", 32768)), 8, int(os.environ.get("M", 10000))
rng = np.random.default_rng(1234)
X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1)); Xs = rng.uniform(0, 1, (M, D))
h = _lib.Handle(0)
h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25*np.sqrt(D)], 1e-2); h.set_candidates(Xs)
def t(fn, n=3):
    fn(); h.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    h.synchronize(); return (time.perf_counter() - t0) / n * 1e3
sep = t(lambda: (h.fit(), h.predict(True)))
for st in [0] + [int(a) for a in sys.argv[1:]]:
    h.set_option("pipe_stages", st)
    print("N=%d M=%d pipe_stages=%d: fused %.1f ms  (separate %.1f ms)" % (N, M, st, t(lambda: h.fit_predict(True)), sep), flush=True)
h.close()
