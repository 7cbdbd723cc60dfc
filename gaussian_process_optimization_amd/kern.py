"""Host-side kernel descriptors mirroring ``GPy.kern.RBF`` / ``GPy.kern.Matern52``.

Reference: GPy/GPy/kern/src/stationary.py:23-82 (Stationary.__init__: variance,
lengthscale, ARD), rbf.py:12-57 (RBF), stationary.py:546-579 (Matern52).  These
objects only *hold* hyper-parameters; every covariance evaluation happens on
the GPU inside libgphip (kbuild.hip).  ``K``/``Kdiag`` are provided for API
parity and run through the device as well.
"""
import numpy as np

from . import _lib
from .parameterization import Param, Parameterized


class Stationary(Parameterized):
    _kernel_id = None

    def __init__(self, input_dim, variance=1., lengthscale=None, ARD=False, active_dims=None, name="stationary",
                 Gower=False, space=None):
        super(Stationary, self).__init__(name)
        # the fork's mixed-variable option (stationary.py:61-65): product of 1-D kernels over the variables of `space`
        self.Gower = bool(Gower)
        self.space = space
        self.input_dim = int(input_dim)
        if active_dims is not None and list(active_dims) != list(range(self.input_dim)):
            raise NotImplementedError("active_dims slicing is outside the accelerated path")
        self.ARD = bool(ARD)
        # stationary.py:66-79
        if not ARD:
            if lengthscale is None:
                lengthscale = np.ones(1)
            else:
                lengthscale = np.asarray(lengthscale, dtype=float).reshape(-1)
                assert lengthscale.size == 1, "Only 1 lengthscale needed for non-ARD kernel"
        else:
            if lengthscale is not None:
                lengthscale = np.asarray(lengthscale, dtype=float).reshape(-1)
                assert lengthscale.size in [1, input_dim], "Bad number of lengthscales"
                if lengthscale.size != input_dim:
                    lengthscale = np.ones(input_dim) * lengthscale
            else:
                lengthscale = np.ones(self.input_dim)
        self.variance = Param("variance", np.atleast_1d(float(variance)))
        self.lengthscale = Param("lengthscale", lengthscale)
        assert self.variance.size == 1
        self.link_parameters(self.variance, self.lengthscale)

    # -- device-backed evaluations (API parity; not used by the fit/predict path) ------
    def K(self, X, X2=None):
        """kern.K(X[, X2]) -- stationary.py:107-140, evaluated by the K-build kernels (kern.py:119 is the contract:
        X2 None -> K(X, X) with the diagonal forced to the variance, X2 given -> the [N, M2] cross covariance)."""
        # one scratch context per DEVICE for the whole process (not per kernel object: a kernel holds hyper-parameters only,
        # so it deep-copies and pickles like the reference's; a loop over kern.K still pays no context creation per call)
        h = _scratch_handle(int(getattr(self, "_device", 0)))
        X = _lib.as_f64(X, 2)
        h.set_data(X, np.zeros((X.shape[0], 1)))
        h.set_params(self._kernel_id, self.ARD, float(self.variance), self.lengthscale.values, 0.0)
        if self.Gower and self.space is not None:
            h.set_gower(*gower_config(self.space, self.input_dim))
        else:
            h.set_gower()
        out = h.kernel_matrix() if X2 is None else h.cross_kernel_matrix(X2)
        if X.shape[0] > _SCRATCH_KEEP_N:      # an N x N device buffer of gigabytes is not kept alive behind the caller's back
            release_scratch(int(getattr(self, "_device", 0)))
        return out

    def Kdiag(self, X):
        ret = np.empty(X.shape[0])  # stationary.py:195-198
        ret[:] = float(self.variance)
        return ret

    def copy(self):
        return self.__class__(self.input_dim, float(self.variance), self.lengthscale.values.copy(), self.ARD,
                              Gower=self.Gower, space=self.space)


_SCRATCH = {}
_SCRATCH_KEEP_N = 4096        # scratch contexts that served more rows than this are closed right after the call (128 MB at 4096)


def _scratch_handle(device=0):
    h = _SCRATCH.get(device)
    if h is None or h.h is None:
        h = _SCRATCH[device] = _lib.Handle(device)
    return h


def release_scratch(device=None):
    """Close the scratch context(s) ``kern.K`` evaluates on (all devices, or one)."""
    for d in ([device] if device is not None else list(_SCRATCH)):
        h = _SCRATCH.pop(d, None)
        if h is not None:
            h.close()


def gower_config(space, input_dim):
    """(is_discrete[D], range[D]) from a design space, as the fork's Stationary.K reads it
    (stationary.py:117-119: space.get_continuous_dims / get_discrete_dims / lengthscales)."""
    cont = list(space.get_continuous_dims())
    disc = list(space.get_discrete_dims())
    ranges = list(space.lengthscales())
    is_disc = np.zeros(input_dim, dtype=np.int32)
    rng = np.ones(input_dim)
    for idx, d in enumerate(cont):
        rng[d] = ranges[idx]
    for d in disc:
        is_disc[d] = 1
    if sorted(cont + disc) != list(range(input_dim)):
        raise ValueError("the Gower kernel needs every input dimension to be continuous or discrete in `space`")
    return is_disc, rng


class RBF(Stationary):
    """GPy.kern.RBF -- k(r) = variance * exp(-r^2 / 2)  (rbf.py:50-51)."""
    _kernel_id = _lib.GP_KERNEL_RBF

    def __init__(self, input_dim, variance=1., lengthscale=None, ARD=False, active_dims=None, name="rbf",
                 Gower=False, space=None):
        super(RBF, self).__init__(input_dim, variance, lengthscale, ARD, active_dims, name, Gower, space)


class Matern52(Stationary):
    """GPy.kern.Matern52 -- k(r) = variance (1 + sqrt5 r + 5/3 r^2) exp(-sqrt5 r)  (stationary.py:575-576)."""
    _kernel_id = _lib.GP_KERNEL_MATERN52

    def __init__(self, input_dim, variance=1., lengthscale=None, ARD=False, active_dims=None, name="Mat52",
                 Gower=False, space=None):
        super(Matern52, self).__init__(input_dim, variance, lengthscale, ARD, active_dims, name, Gower, space)
