"""Three emulated predicts at C3 (for rocprofv3 --kernel-trace --stats)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_process_optimization_amd import _lib
import bench
N, D, M = 16384, 8, 10000
X, Y, Xs = bench.synthetic(N, D, M)
h = _lib.Handle(0)
h.set_option("emulate_fp64", 1)
for kv in sys.argv[1:]:
    k, v = kv.split("="); h.set_option(k, int(v))
h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25 * np.sqrt(D)], 1e-2); h.set_candidates(Xs)
h.fit()
for _ in range(4):
    h.predict(True)
h.close()
