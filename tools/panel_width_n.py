"""gp_fit / gp_fit_predict / gp_fit_grad against the panel width at several N (test tooling). usage: panel_width_n.py N[,N..] W[,W..]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussian_process_optimization_amd import _lib
Ns = [int(v) for v in sys.argv[1].split(",")]; Ws = [int(v) for v in sys.argv[2].split(",")]
D, M = 8, 10000
h = _lib.Handle(0)
def t(fn, n=4):
    fn(); fn(); h.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    h.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for N in Ns:
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1)); Xs = rng.uniform(0, 1, (M, D))
    for W in Ws:
        h.set_option("panel_tiles", W)
        h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25 * np.sqrt(D)], 1e-2); h.set_candidates(Xs)
        print("N=%5d W=%2d  fit %.2f  fit_predict %.2f  fit+predict %.2f  fit_grad %.2f ms" % (
            N, W, t(lambda: h.fit()), t(lambda: h.fit_predict(True)), t(lambda: (h.fit(), h.predict(True))), t(lambda: h.fit_grad(1), 2)), flush=True)
h.close()
