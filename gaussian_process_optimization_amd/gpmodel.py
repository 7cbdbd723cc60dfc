"""``GPModel`` -- host mirror of ``GPyOpt.models.GPModel`` on top of the HIP ``GPRegression``.

Reference: GPyOpt/GPyOpt/models/base.py:7-33 (BOModel contract),
GPyOpt/GPyOpt/models/gpmodel.py:9-177 (GPModel).  Same constructor keywords, same
return conventions: ``predict`` returns (mean, **std**) with the variance clipped at
1e-10 (:95-112), ``get_fmin`` is the minimum posterior mean over the training inputs
(:125-129; cached per fit on the device -- the reference recomputes the identical
value on every acquisition call), ``predict_withGradients`` (:131-142).
"""
import numpy as np

from . import kern as _kern
from .gp_regression import GPRegression


class BOModel(object):
    """GPyOpt/GPyOpt/models/base.py:7-33."""
    MCMC_sampler = False
    analytical_gradient_prediction = False

    def updateModel(self, X_all, Y_all, X_new, Y_new):
        return

    def predict(self, X):
        return

    def predict_withGradients(self, X):
        return

    def get_fmin(self):
        return


class GPModel(BOModel):
    analytical_gradient_prediction = True  # gpmodel.py:29

    def _gower(self):
        return bool(self.Gower and self.space is not None)

    def __init__(self, kernel=None, noise_var=None, exact_feval=False, optimizer='bfgs', max_iters=1000,
                 optimize_restarts=5, sparse=False, num_inducing=10, verbose=True, ARD=False, Gower=False,
                 space=None, device=0):
        if sparse:
            raise NotImplementedError("sparse GP is a different model family (out of scope)")
        self.kernel = kernel
        self.noise_var = noise_var
        self.exact_feval = exact_feval
        self.optimize_restarts = optimize_restarts
        self.optimizer = optimizer
        self.max_iters = max_iters
        self.verbose = verbose
        self.sparse = sparse
        self.num_inducing = num_inducing
        self.model = None
        self.ARD = ARD
        self.Gower = Gower
        self.space = space
        self.device = device

    @staticmethod
    def fromConfig(config):
        return GPModel(**config)

    def _create_model(self, X, Y):
        """gpmodel.py:50-76."""
        self.input_dim = X.shape[1]
        if self.kernel is None:
            kern = _kern.Matern52(self.input_dim, variance=1., ARD=self.ARD, Gower=self.Gower, space=self.space)
        else:
            kern = self.kernel
            self.kernel = None
        noise_var = Y.var() * 0.01 if self.noise_var is None else self.noise_var
        self.model = GPRegression(X, Y, kernel=kern, noise_var=noise_var, device=self.device)
        if self.exact_feval:
            self.model.Gaussian_noise.constrain_fixed(1e-6, warning=False)
        else:
            self.model.Gaussian_noise.constrain_bounded(1e-9, 1e6, warning=False)

    def updateModel(self, X_all, Y_all, X_new, Y_new):
        """gpmodel.py:78-93."""
        if self.model is None:
            self._create_model(X_all, Y_all)
        else:
            self.model.set_XY(X_all, Y_all)
        if self.max_iters > 0:
            if self.optimize_restarts == 1:
                self.model.optimize(optimizer=self.optimizer, max_iters=self.max_iters, messages=False,
                                    ipython_notebook=False)
            else:
                self.model.optimize_restarts(num_restarts=self.optimize_restarts, optimizer=self.optimizer,
                                             max_iters=self.max_iters, verbose=self.verbose)

    def _predict(self, X, full_cov, include_likelihood):
        if X.ndim == 1:
            X = X[None, :]
        m, v = self.model.predict(X, full_cov=full_cov, include_likelihood=include_likelihood)
        v = np.clip(v, 1e-10, np.inf)
        return m, v

    def predict(self, X, with_noise=True):
        m, v = self._predict(X, False, with_noise)
        return m, np.sqrt(v)

    def predict_covariance(self, X, with_noise=True):
        _, v = self._predict(X, True, with_noise)
        return v

    def get_fmin(self):
        """gpmodel.py:125-129 -- ``self.model.predict(self.model.X)[0].min()`` evaluated on the device."""
        m = self.model
        m._ensure_fit()
        f = m._h.fmin()
        if m.normalizer is not None:
            f = float(m.normalizer.inverse_mean(np.array([[f]]))[0, 0])
        return f

    def predict_withGradients(self, X):
        """gpmodel.py:131-142."""
        if X.ndim == 1:
            X = X[None, :]
        m, v = self.model.predict(X)
        v = np.clip(v, 1e-10, np.inf)
        dmdx, dvdx = self.model.predictive_gradients(X)
        dmdx = dmdx[:, :, 0]
        if self.model.normalizer is not None:  # gradients of the un-normalised mean / variance
            dmdx = dmdx * self.model.normalizer.std
            dvdx = dvdx * self.model.normalizer.std ** 2
        dsdx = dvdx / (2 * np.sqrt(v))
        return m, np.sqrt(v), dmdx, dsdx

    def copy(self):
        copied = GPModel(kernel=self.model.kern.copy(), noise_var=self.noise_var, exact_feval=self.exact_feval,
                         optimizer=self.optimizer, max_iters=self.max_iters,
                         optimize_restarts=self.optimize_restarts, verbose=self.verbose, ARD=self.ARD,
                         device=self.device)
        copied._create_model(self.model.X, self.model.Y)
        copied.updateModel(self.model.X, self.model.Y, None, None)
        return copied

    def get_model_parameters(self):
        return np.atleast_2d(self.model[:])

    def get_covariance_between_points(self, x1, x2):
        """gpmodel.py:173-177."""
        return self.model.posterior_covariance_between_points(x1, x2)

    def get_model_parameters_names(self):
        return self.model.parameter_names_flat().tolist()
