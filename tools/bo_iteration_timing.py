"""One BO iteration's acquisition optimisation (anchor scoring + L-BFGS-B from the 5 best anchors: GPyOpt/GPyOpt/optimization/
acquisition_optimizer.py:46-77) on a fitted model, with and without the small-M path of the one-row calls (test tooling)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gaussian_process_optimization_amd as gpo

for N in (500, 4000, 16384):
    D = 8
    rng = np.random.default_rng(3)
    X = rng.uniform(0, 1, (N, D))
    Y = np.sin(2 * np.pi * X).sum(1, keepdims=True) / np.sqrt(D) + 0.05 * rng.standard_normal((N, 1))
    dom = [{'name': 'x', 'type': 'continuous', 'domain': (0, 1), 'dimensionality': D}]
    bo = gpo.methods.BayesianOptimization(f=None, domain=dom, X=X, Y=Y, model_type='GP', acquisition_type='EI', normalize_Y=True,
                                          kernel=gpo.kern.RBF(D, 1.0, 0.25 * np.sqrt(D)), noise_var=1e-2, max_iters=0)
    for small in (8, 0):
        np.random.seed(1)
        bo.suggest_next_locations()                        # fits the model, warms every buffer
        bo.model.model._h.set_option("small_m", small)
        calls = {"n": 0}
        orig = bo.acquisition.acquisition_function_withGradients
        def counted(x, _o=orig):
            calls["n"] += 1
            return _o(x)
        bo.acquisition.acquisition_function_withGradients = counted
        np.random.seed(1)
        t0 = time.perf_counter(); xn = bo.suggest_next_locations(); dt = time.perf_counter() - t0
        bo.acquisition.acquisition_function_withGradients = orig
        print("N=%5d small_m=%d: suggest_next_locations %.1f ms (%d gradient calls of the acquisition optimiser)" % (N, small, dt * 1e3, calls["n"]), flush=True)
    bo.model.model.close()
