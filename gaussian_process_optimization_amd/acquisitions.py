"""Acquisition functions mirroring GPyOpt's (EI / LCB / MPI) with device-side batched scoring.

Reference: GPyOpt/GPyOpt/acquisitions/base.py:7-68 (AcquisitionBase: ``acquisition_function``
returns the NEGATED value weighted by constraints and cost), EI.py:7-51, LCB.py:6-46,
MPI.py:7-51, GPyOpt/GPyOpt/util/general.py:113-129 (get_quantiles).

``_compute_acq`` / ``_compute_acq_withGradients`` keep the reference's host formulas on top of
``model.predict`` (so any BOModel works); when the model is the HIP ``GPModel`` and there are no
constraints or cost, ``acquisition_function`` and ``argbest`` score the whole candidate table
inside libgphip (gp_acq / gp_acq_argbest) -- the batched call pattern of
GPyOpt/GPyOpt/optimization/anchor_points_generator.py:59,96-98 and run.py:1240-1241.
"""
import numpy as np
from scipy.special import erfc

from . import _lib
from .gpmodel import GPModel


def get_quantiles(acquisition_par, fmin, m, s):
    """GPyOpt/GPyOpt/util/general.py:113-129."""
    if isinstance(s, np.ndarray):
        s[s < 1e-10] = 1e-10
    elif s < 1e-10:
        s = 1e-10
    u = (fmin - m - acquisition_par) / s
    phi = np.exp(-0.5 * u ** 2) / np.sqrt(2 * np.pi)
    Phi = 0.5 * erfc(-u / np.sqrt(2))
    return (phi, Phi, u)


def constant_cost_withGradients(x):
    """GPyOpt/GPyOpt/core/task/cost.py:76."""
    return np.ones(x.shape[0])[:, None], np.zeros(x.shape)


class _NoConstraints(object):
    def indicator_constraints(self, x):
        return np.ones((np.atleast_2d(x).shape[0], 1))

    def has_constraints(self):
        return False


class AcquisitionBase(object):
    """acquisitions/base.py:7-68."""
    analytical_gradient_prediction = False
    _acq_id = None

    def __init__(self, model, space=None, optimizer=None, cost_withGradients=None):
        self.model = model
        self.space = space if space is not None else _NoConstraints()
        self.optimizer = optimizer
        self.analytical_gradient_acq = self.analytical_gradient_prediction and self.model.analytical_gradient_prediction
        self.cost_withGradients = constant_cost_withGradients if cost_withGradients is None else cost_withGradients

    # -- device fast path ---------------------------------------------------------------
    def _device_ok(self):
        if not isinstance(self.model, GPModel) or self._acq_id is None or self.model.model is None:
            return False
        if self.cost_withGradients is not constant_cost_withGradients:
            return False
        has = getattr(self.space, "has_constraints", None)
        if has is not None and has():
            return False
        return self.model.model.output_dim == 1

    def _par(self):
        raise NotImplementedError

    def _device_stage(self, x):
        gp = self.model.model
        x = np.atleast_2d(np.asarray(x, dtype=float))
        gp._stage(x)
        nz = gp.normalizer
        y_mean = float(nz.mean[0]) if nz is not None else 0.0
        y_std = float(nz.std[0]) if nz is not None else 1.0
        fmin = self.model.get_fmin() if self._acq_id != _lib.GP_ACQ_LCB else 0.0
        return gp, fmin, y_mean, y_std

    def acquisition_function(self, x):
        """base.py:33-39: -(acq * indicator_constraints) / cost."""
        if self._device_ok():
            gp, fmin, y_mean, y_std = self._device_stage(x)
            return gp._h.acq(self._acq_id, self._par(), fmin, y_mean, y_std)
        f_acqu = self._compute_acq(x)
        cost_x, _ = self.cost_withGradients(x)
        return -(f_acqu * self.space.indicator_constraints(x)) / cost_x

    def acquisition_function_withGradients(self, x):
        """base.py:42-50."""
        f_acqu, df_acqu = self._compute_acq_withGradients(x)
        cost_x, cost_grad_x = self.cost_withGradients(x)
        f_acq_cost = f_acqu / cost_x
        df_acq_cost = (df_acqu * cost_x - f_acqu * cost_grad_x) / (cost_x ** 2)
        ind = self.space.indicator_constraints(x)
        return -f_acq_cost * ind, -df_acq_cost * ind

    def argbest(self, x, sense=-1):
        """Index and value of the best row of ``acquisition_function(x)``.

        sense=-1: the smallest (GPyOpt's convention, anchor_points_generator.py:61);
        sense=+1: the largest (run.py:1241 takes ``np.argmax``).  Ties -> lowest index.
        """
        if self._device_ok():
            gp, fmin, y_mean, y_std = self._device_stage(x)
            return gp._h.acq_argbest(self._acq_id, self._par(), fmin, sense, y_mean, y_std)
        a = self.acquisition_function(x)[:, 0]
        i = int(np.argmin(a) if sense < 0 else np.argmax(a))
        return i, float(a[i])

    def optimize(self, duplicate_manager=None):
        """base.py:52-60."""
        if not self.analytical_gradient_acq:
            return self.optimizer.optimize(f=self.acquisition_function, duplicate_manager=duplicate_manager)
        return self.optimizer.optimize(f=self.acquisition_function, f_df=self.acquisition_function_withGradients,
                                       duplicate_manager=duplicate_manager)

    def _compute_acq(self, x):
        raise NotImplementedError('')

    def _compute_acq_withGradients(self, x):
        raise NotImplementedError('')


class AcquisitionEI(AcquisitionBase):
    """Expected improvement, EI.py:7-51."""
    analytical_gradient_prediction = True
    _acq_id = _lib.GP_ACQ_EI

    def __init__(self, model, space=None, optimizer=None, cost_withGradients=None, jitter=0.01):
        self.optimizer = optimizer
        super(AcquisitionEI, self).__init__(model, space, optimizer, cost_withGradients=cost_withGradients)
        self.jitter = jitter

    @staticmethod
    def fromConfig(model, space, optimizer, cost_withGradients, config):
        return AcquisitionEI(model, space, optimizer, cost_withGradients, jitter=config['jitter'])

    def _par(self):
        return self.jitter

    def _compute_acq(self, x):
        m, s = self.model.predict(x)
        fmin = self.model.get_fmin()
        phi, Phi, u = get_quantiles(self.jitter, fmin, m, s)
        return s * (u * Phi + phi)

    def _compute_acq_withGradients(self, x):
        fmin = self.model.get_fmin()
        m, s, dmdx, dsdx = self.model.predict_withGradients(x)
        phi, Phi, u = get_quantiles(self.jitter, fmin, m, s)
        f_acqu = s * (u * Phi + phi)
        df_acqu = dsdx * phi - Phi * dmdx
        return f_acqu, df_acqu


class AcquisitionLCB(AcquisitionBase):
    """GP lower confidence bound, LCB.py:6-46."""
    analytical_gradient_prediction = True
    _acq_id = _lib.GP_ACQ_LCB

    def __init__(self, model, space=None, optimizer=None, cost_withGradients=None, exploration_weight=2):
        self.optimizer = optimizer
        super(AcquisitionLCB, self).__init__(model, space, optimizer)
        self.exploration_weight = exploration_weight
        if cost_withGradients is not None:
            print('The set cost function is ignored! LCB acquisition does not make sense with cost.')

    @staticmethod
    def fromConfig(model, space, optimizer, cost_withGradients, config):
        return AcquisitionLCB(model, space, optimizer, cost_withGradients, exploration_weight=config['weight'])

    def _par(self):
        return self.exploration_weight

    def _compute_acq(self, x):
        m, s = self.model.predict(x)
        return -m + self.exploration_weight * s

    def _compute_acq_withGradients(self, x):
        m, s, dmdx, dsdx = self.model.predict_withGradients(x)
        return -m + self.exploration_weight * s, -dmdx + self.exploration_weight * dsdx


class AcquisitionMPI(AcquisitionBase):
    """Maximum probability of improvement, MPI.py:7-51."""
    analytical_gradient_prediction = True
    _acq_id = _lib.GP_ACQ_MPI

    def __init__(self, model, space=None, optimizer=None, cost_withGradients=None, jitter=0.01):
        self.optimizer = optimizer
        super(AcquisitionMPI, self).__init__(model, space, optimizer, cost_withGradients=cost_withGradients)
        self.jitter = jitter

    @staticmethod
    def fromConfig(model, space, optimizer, cost_withGradients, config):
        return AcquisitionMPI(model, space, optimizer, cost_withGradients, jitter=config['jitter'])

    def _par(self):
        return self.jitter

    def _compute_acq(self, x):
        m, s = self.model.predict(x)
        fmin = self.model.get_fmin()
        _, Phi, _ = get_quantiles(self.jitter, fmin, m, s)
        return Phi

    def _compute_acq_withGradients(self, x):
        fmin = self.model.get_fmin()
        m, s, dmdx, dsdx = self.model.predict_withGradients(x)
        phi, Phi, u = get_quantiles(self.jitter, fmin, m, s)
        return Phi, -(phi / s) * (dmdx + dsdx * u)
