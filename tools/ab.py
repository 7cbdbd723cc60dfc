"""A/B of option settings inside ONE process, alternating (test tooling):  ab.py name=v1,v2[,v3] [fixed=val ...] [--predict|--fit|--fused]
Prints the median wall time of gp_predict / gp_fit / gp_fit_predict at C3 per setting, and whether mean / variance are bitwise the first setting's."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussian_process_optimization_amd import _lib
args = [a for a in sys.argv[1:] if not a.startswith("--")]
modes = [a[2:] for a in sys.argv[1:] if a.startswith("--")] or ["predict", "fused"]
sweep = [a for a in args if "," in a][0]
fixed = [a for a in args if "," not in a]
name, vals = sweep.split("=")[0], [int(v) for v in sweep.split("=")[1].split(",")]
N, D, M = 16384, 8, 10000
rng = np.random.default_rng(1234)
X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1)); Xs = rng.uniform(0, 1, (M, D))
h = _lib.Handle(0)
for k, v in [a.split("=") for a in fixed]:
    h.set_option(k, int(v))
h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25 * np.sqrt(D)], 1e-2); h.set_candidates(Xs)
h.fit()
fn = {"predict": lambda: h.predict(True), "fit": lambda: h.fit(), "fused": lambda: h.fit_predict(True)}
for mode in modes:
    times = {v: [] for v in vals}
    ref = None
    same = {}
    for rep in range(7):
        for v in vals:
            h.set_option(name, v)
            h.synchronize()
            t0 = time.perf_counter(); r = fn[mode](); times[v].append((time.perf_counter() - t0) * 1e3)
            if mode != "fit":
                mv = r[-2:] if mode == "fused" else r
                if ref is None: ref = (mv[0].copy(), mv[1].copy())
                same[v] = bool(np.array_equal(mv[0], ref[0]) and np.array_equal(mv[1], ref[1]))
    print(mode, " ".join("%s=%d: %.2f ms (min %.2f)%s" % (name, v, np.median(times[v][1:]), min(times[v][1:]), "" if same.get(v, True) else " DIFFERENT RESULT") for v in vals), flush=True)
h.close()
