"""Option A/B at several N, alternating inside one process (test tooling):
   opt_sweep.py name=v1,v2[,...] [fixed=val ...] [--sizes 8192,16384,32768]
   opt_sweep.py nameA:nameB=a1:b1,a2:b2 ...          (several options per setting)
Prints per N the median gp_fit / gp_fit_predict / gp_fit_grad wall time per setting and whether LML, mean and variance are bitwise the
first setting's."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussian_process_optimization_amd import _lib

argv = sys.argv[1:]
sizes = [8192, 16384, 32768]
if "--sizes" in argv:
    i = argv.index("--sizes"); sizes = [int(x) for x in argv[i + 1].split(",")]; del argv[i:i + 2]
sweep = [a for a in argv if "," in a][0]
fixed = [a.split("=") for a in argv if "," not in a]
names = sweep.split("=")[0].split(":")
vals = [tuple(int(x) for x in v.split(":")) for v in sweep.split("=")[1].split(",")]
name = ":".join(names)
def apply(v):
    for k, x in zip(names, v): h.set_option(k, x)
lab = lambda v: ":".join(str(x) for x in v)
h = _lib.Handle(0)
for k, v in fixed: h.set_option(k, int(v))
for N in sizes:
    D, M = 8, 10000 if N >= 16384 else 5000
    rng = np.random.default_rng(1234)
    X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1)); Xs = rng.uniform(0, 1, (M, D))
    h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25 * np.sqrt(D)], 1e-2); h.set_candidates(Xs)
    h.fit()
    reps = 4 if N >= 32768 else 7
    for mode, fn in (("fit", h.fit), ("fit_predict", lambda: h.fit_predict(True)), ("fit_grad", lambda: h.fit_grad(1))):
        times = {v: [] for v in vals}; ref = None; same = {}
        for rep in range(reps):
            for v in vals:
                apply(v); h.synchronize()
                t0 = time.perf_counter(); r = fn(); times[v].append((time.perf_counter() - t0) * 1e3)
                flat = []
                def walk(x):
                    if isinstance(x, tuple):
                        for y in x: walk(y)
                    else: flat.append(np.asarray(x))
                walk(r)
                if ref is None: ref = [x.copy() for x in flat]
                same[v] = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(flat, ref))
        print("N=%5d %-11s " % (N, mode) + "  ".join("%s=%s: %.2f (min %.2f)%s" % (name, lab(v), np.median(times[v][1:]), min(times[v][1:]),
              "" if same[v] else " DIFFERENT") for v in vals), flush=True)
h.close()
