"""Candidate solve (C3) against the super-tile edge of the long GEMM launches, alternating in one process (test tooling)."""
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
vals = [int(v) for v in sys.argv[1:]] or [8, 0, 4, 16, 8, 0, 4, 16]
for st in vals:
    h.set_option("supertile", st)
    h.predict(True); h.synchronize(); t0 = time.perf_counter()
    for _ in range(4):
        h.predict(True)
    h.synchronize(); dt = (time.perf_counter() - t0) / 4 * 1e3
    print("supertile=%2d  predict %.2f ms (%.1f TFLOP/s)" % (st, dt, float(N) * N * M / dt / 1e9), flush=True)
h.close()
