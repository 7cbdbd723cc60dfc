"""gp_fit time with the cooperative tail kernel ("tail_tiles") against the stream version, at C2 and C3."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_process_optimization_amd import _lib
import bench
h = _lib.Handle(0)
for N, D, tails in ((4096, 4, (0, 64, 32, 16)), (2048, 4, (0, 64)), (8192, 8, (0, 24, 36, 48, 64)), (16384, 8, (0, 18, 24, 36, 48, 60))):
    X, Y, _ = bench.synthetic(N, D, 8)
    h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25 * np.sqrt(D)], 1e-2)
    ref = None
    for tail in tails:
        for wgs in ((0,) if tail == 0 else (0, 128, 64)):
            h.set_option("tail_tiles", tail); h.set_option("tail_wgs", wgs)
            h.fit(); h.fit()
            h.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                lml = h.fit()[0]
            h.synchronize(); dt = (time.perf_counter() - t0) / 5 * 1e3
            ph = {p["name"]: round(p["ms"], 3) for p in h.phases()}
            if ref is None: ref = lml
            print("N=%5d tail_tiles=%3d wgs=%3d  fit %.3f ms  cholesky %.3f ms (%.1f TFLOP/s)  lml rel diff %.1e"
                  % (N, tail, wgs, dt, ph["cholesky"], N ** 3 / 3.0 / ph["cholesky"] / 1e9, abs(lml - ref) / abs(ref)))
h.close()
