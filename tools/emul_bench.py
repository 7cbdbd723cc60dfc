"""The bench step with emulate_fp64 on ONLY (gp_fit + gp_predict + EI arg-best at C3), for a per-mode kernel-stats profile:
rocprofv3 --kernel-trace --stats -- python3 tools/emul_bench.py   (test tooling; bench.py's second line is the measurement)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gaussian_process_optimization_amd import _lib

N, D, M = 16384, 8, 10000
X, Y, Xs = bench.synthetic(N, D, M)
h = _lib.Handle(0)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    h.set_option(k, int(v))
h.set_option("emulate_fp64", 1)
h.set_data(X, Y)
h.set_params(_lib.GP_KERNEL_RBF, 0, 1.0, [0.25 * D ** 0.5], 1e-2)
h.set_candidates(Xs)


def step():
    lml = h.fit()[0]
    h.predict(True)
    return lml, h.acq_argbest(_lib.GP_ACQ_EI, 0.01, h.fmin(), -1)


for _ in range(2):
    step()
h.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    out = step()
h.synchronize()
print("emulated step %.2f ms" % ((time.perf_counter() - t0) / 10 * 1e3), out)
h.close()
