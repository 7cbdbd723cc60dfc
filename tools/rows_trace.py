"""A burst of one-location gradient calls through gp_acq_rows (csrc/onerow.hip) for a kernel trace or a wall-clock figure
(test tooling).  usage: rows_trace.py N [calls]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussian_process_optimization_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
D = 8
rng = np.random.default_rng(1)
X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1))
h = _lib.Handle(0)
h.set_option("emulate_fp64", 0)
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    h.set_option(k, int(v))
h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.7], 1e-2); h.fit(); fmin = h.fmin()
Xs = rng.uniform(0, 1, (1, D))
t0 = time.perf_counter(); h.acq_rows(Xs, 0, 0.01, fmin, grad=True); t_first = time.perf_counter() - t0
for what, fn in (("acq_rows value + gradient", lambda: h.acq_rows(Xs, 0, 0.01, fmin, grad=True)),
                 ("acq_rows value", lambda: h.acq_rows(Xs, 0, 0.01, fmin)),
                 ("predict_rows", lambda: h.predict_rows(Xs, True)),
                 ("predict_rows + gradients", lambda: h.predict_rows(Xs, True, grad=True))):
    fn()
    t0 = time.perf_counter()
    for _ in range(calls):
        fn()
    dt = (time.perf_counter() - t0) / calls
    lower = 8.0 * N * N / 2
    passes = 2 if "grad" in what else 1
    print("N=%d %-28s %8.1f us per call   (%d x %.2f GB of L^-1 -> %.2f TB/s incl. launch + sync)" % (
        N, what, dt * 1e6, passes, lower / 1e9, passes * lower / dt / 1e12), flush=True)
print("first gradient call after the fit (builds L^-1): %.2f ms" % (t_first * 1e3))
h.close()
