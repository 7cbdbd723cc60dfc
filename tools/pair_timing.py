"""Candidate solve with one vs two panels per update launch ("pair_panels"), C3; bitwise equality of the results."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_process_optimization_amd import _lib
import bench
N, D, M = 16384, 8, 10000
X, Y, Xs = bench.synthetic(N, D, M)
h = _lib.Handle(0)
h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25 * np.sqrt(D)], 1e-2); h.set_candidates(Xs)
h.fit()
ref = None
for pair in (0, 1, 0, 1):
    h.set_option("pair_panels", pair)
    h.predict(True)
    h.synchronize(); t0 = time.perf_counter()
    for _ in range(4):
        mu, var = h.predict(True)
    h.synchronize(); dt = (time.perf_counter() - t0) / 4 * 1e3
    if ref is None: ref = (mu.copy(), var.copy())
    t0 = time.perf_counter()
    for _ in range(4):
        h.fit_predict(True)
    h.synchronize(); df = (time.perf_counter() - t0) / 4 * 1e3
    print("pair_panels=%d  predict %.2f ms (%.1f TFLOP/s)  fit_predict %.2f ms   bitwise equal to unpaired: %s"
          % (pair, dt, float(N) * N * M / dt / 1e9, df, np.array_equal(mu, ref[0]) and np.array_equal(var, ref[1])))
h.close()
