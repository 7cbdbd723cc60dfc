"""cProfile of one BO iteration's suggest_next_locations at N = 500 (test tooling): where the host time goes."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gaussian_process_optimization_amd as gpo
N, D = 500, 8
rng = np.random.default_rng(3)
X = rng.uniform(0, 1, (N, D))
Y = np.sin(2 * np.pi * X).sum(1, keepdims=True) / np.sqrt(D) + 0.05 * rng.standard_normal((N, 1))
dom = [{'name': 'x', 'type': 'continuous', 'domain': (0, 1), 'dimensionality': D}]
bo = gpo.methods.BayesianOptimization(f=None, domain=dom, X=X, Y=Y, model_type='GP', acquisition_type='EI', normalize_Y=True,
                                      kernel=gpo.kern.RBF(D, 1.0, 0.25 * np.sqrt(D)), noise_var=1e-2, max_iters=0)
np.random.seed(1); bo.suggest_next_locations()
np.random.seed(1)
t0 = time.perf_counter(); bo.suggest_next_locations(); print("wall %.2f ms" % ((time.perf_counter() - t0) * 1e3))
np.random.seed(1)
pr = cProfile.Profile(); pr.enable(); bo.suggest_next_locations(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
