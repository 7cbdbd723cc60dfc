"""gaussian_process_optimization_amd -- MI355X-native exact-GP regression hot path.

Drop-in surface for the path GPyOpt/GPy sit on:

    import gaussian_process_optimization_amd as gpo
    m = gpo.models.GPRegression(X, Y, gpo.kern.RBF(D), noise_var=1e-2)   # GPy.models.GPRegression
    m.log_likelihood(); m.predict(Xs); m.predictive_gradients(Xs); m.optimize()
    bo = gpo.methods.BayesianOptimization(f=None, domain=..., X=X, Y=Y)  # GPyOpt.methods
    acq = gpo.acquisitions.AcquisitionEI(gpo.GPModel(...), ...)           # GPyOpt.acquisitions

Host code is plain Python + ctypes over the C-ABI in include/gphip.h; every
numeric step runs in hand-written HIP kernels for gfx950 (csrc/).  There is no
CPU fallback: importing the package works anywhere, using it needs an MI355X.
"""
import types as _types

from . import _lib
from . import kern
from .gp_regression import GPRegression, Gaussian, Standardize
from .gpmodel import GPModel, BOModel
from . import acquisitions
from .acquisitions import (AcquisitionEI, AcquisitionLCB, AcquisitionMPI, AcquisitionBase, AcquisitionLP,
                           LocalPenalization, estimate_L)
from .bayesian_optimization import BayesianOptimization, Design_space, AcquisitionOptimizer
from .sharded import ShardedCandidates, merge_best

# namespaces named like the reference packages
models = _types.SimpleNamespace(GPRegression=GPRegression, GPModel=GPModel)
methods = _types.SimpleNamespace(BayesianOptimization=BayesianOptimization)
likelihoods = _types.SimpleNamespace(Gaussian=Gaussian)

__all__ = ["kern", "models", "methods", "likelihoods", "acquisitions", "GPRegression", "GPModel", "BOModel",
           "AcquisitionEI", "AcquisitionLCB", "AcquisitionMPI", "AcquisitionBase", "AcquisitionLP",
           "LocalPenalization", "estimate_L", "BayesianOptimization",
           "Design_space", "AcquisitionOptimizer", "ShardedCandidates", "merge_best", "Standardize"]
