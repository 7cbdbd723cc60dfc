"""A few seconds of one phase of the bench step in a loop, for sampling clocks / power beside it (test tooling).
usage: load_loop.py predict|fit|fused|emulated [seconds]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_process_optimization_amd import _lib
import bench
what = sys.argv[1] if len(sys.argv) > 1 else "predict"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
N, D, M = 16384, 8, 10000
X, Y, Xs = bench.synthetic(N, D, M)
h = _lib.Handle(0)
h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25 * np.sqrt(D)], 1e-2); h.set_candidates(Xs)
if what == "emulated": h.set_option("emulate_fp64", 1)
h.fit(); h.predict(True)
fn = {"predict": lambda: h.predict(True), "fit": lambda: h.fit(), "fused": lambda: h.fit_predict(True),
      "emulated": lambda: (h.fit(), h.predict(True))}[what]
print("start %s %.3f" % (what, time.time()), flush=True)
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    fn(); n += 1
h.synchronize()
print("end %s %.3f  %d calls, %.2f ms each" % (what, time.time(), n, (time.time() - t0) / n * 1e3), flush=True)
h.close()
