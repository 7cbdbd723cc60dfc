"""gp_fit_predict vs two calls at another N, over (pipe_stages, pipe_start_pct) pairs given as s:p (test tooling)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussian_process_optimization_amd import _lib
N, D, M = int(os.environ.get("N", 32768)), 8, int(os.environ.get("M", 10000))
rng = np.random.default_rng(1234)
X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1)); Xs = rng.uniform(0, 1, (M, D))
h = _lib.Handle(0)
h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25*np.sqrt(D)], 1e-2); h.set_candidates(Xs)
def t(fn, n=5):
    fn(); h.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    h.synchronize(); return (time.perf_counter() - t0) / n * 1e3
sep = t(lambda: (h.fit(), h.predict(True)))
for a in ["0:40"] + sys.argv[1:]:
    st, pc = [int(x) for x in a.split(":")]
    h.set_option("pipe_stages", st); h.set_option("pipe_start_pct", pc)
    print("N=%d M=%d pipe_stages=%d start=%d%%: fused %.2f ms  (separate %.2f ms)" % (N, M, st, pc, t(lambda: h.fit_predict(True)), sep), flush=True)
h.close()
