"""Small N: gp_fit and gp_fit_predict with the look-ahead factorisation vs the single-stream one (option lookahead), M = 2000 (test tooling)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussian_process_optimization_amd import _lib
D, M = 8, int(os.environ.get("M", 2000))
h = _lib.Handle(0)
def t(fn, n=8):
    fn(); fn(); h.synchronize(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); h.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]
for N in [int(a) for a in sys.argv[1:]] or [1024, 2048, 3072, 4096, 5120, 6144, 8192]:
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1)); Xs = rng.uniform(0, 1, (M, D))
    h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25 * np.sqrt(D)], 1e-2); h.set_candidates(Xs)
    out = []
    for la in (1, 0):
        h.set_option("lookahead", la)
        out.append((la, t(lambda: h.fit()), t(lambda: h.fit_predict(True)), t(lambda: (h.fit(), h.predict(True)))))
    print("N=%5d  " % N + "   ".join("lookahead=%d: fit %.2f  fit_predict %.2f  two calls %.2f ms" % o for o in out), flush=True)
h.set_option("lookahead", 1)
h.close()
