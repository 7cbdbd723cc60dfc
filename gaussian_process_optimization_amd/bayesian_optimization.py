"""``BayesianOptimization`` facade mirroring GPyOpt's entry point for the GP path.

Reference: GPyOpt/GPyOpt/methods/bayesian_optimization.py:76-170 (constructor keywords),
GPyOpt/GPyOpt/util/arguments_manager.py:42-147 (kwargs -> objects: kernel, ARD, noise_var,
model_optimizer_type, max_iters, optimize_restarts, acquisition_jitter=0.01,
acquisition_weight=2), GPyOpt/GPyOpt/core/bo.py:55-71 (suggest_next_locations), :73-168
(run_optimization), :216-254 (_compute_next_evaluations / _update_model, Y normalised with
util/general.py:203-234), GPyOpt/GPyOpt/optimization/acquisition_optimizer.py:46-77 and
anchor_points_generator.py:19-98 (1000 random candidates scored in ONE batched call, best 5
anchors, L-BFGS-B from each), optimizer.py:28-61 (scipy fmin_l_bfgs_b).

Only the GP model type and the EI / LCB / MPI acquisitions with the sequential evaluator are
on the accelerated path; the design space here covers continuous and discrete variables
(bounds, rounding).  Everything numeric -- fit, batched scoring, arg-best -- runs on the GPU.
"""
import time

import numpy as np
from scipy import optimize as _sopt

from . import kern as _kern
from .acquisitions import AcquisitionEI, AcquisitionLCB, AcquisitionMPI, AcquisitionLP, LocalPenalization
from .gpmodel import GPModel


def normalize(Y, normalization_type='stats'):
    """GPyOpt/GPyOpt/util/general.py:203-234."""
    Y = np.asarray(Y, dtype=float)
    if np.max(Y.shape) != Y.size:
        raise NotImplementedError('Only 1-dimensional arrays are supported.')
    if normalization_type == 'stats':
        Y_norm = Y - Y.mean()
        std = Y.std()
        if std > 0:
            Y_norm /= std
    elif normalization_type == 'maxmin':
        Y_norm = Y - Y.min()
        y_range = np.ptp(Y)
        if y_range > 0:
            Y_norm /= y_range
            Y_norm = 2 * (Y_norm - 0.5)
    else:
        raise ValueError('Unknown normalization type: {}'.format(normalization_type))
    return Y_norm


class Design_space(object):
    """Continuous / discrete box domain (subset of GPyOpt/GPyOpt/core/task/space.py:13-532)."""

    def __init__(self, space, constraints=None):
        # space.py:303-318: every constraint is the body of ``lambda x: ...`` over the 2-D array of locations; a
        # location is feasible where the expression is < 0.  Compiled once here (the reference re-execs per call).
        self.constraints = constraints
        self._constraint_fns = []
        for d in (constraints or []):
            try:
                self._constraint_fns.append(eval('lambda x: ' + d['constraint'], {'np': np, 'numpy': np}))
            except Exception:
                print('Fail to compile the constraint: ' + str(d))
                raise
        self.config_space = space
        self.names, self.types, self.domains = [], [], []
        for i, v in enumerate(space):
            n = int(v.get('dimensionality', 1))
            for k in range(n):
                self.names.append(v.get('name', 'var_%d' % i) + ('_%d' % k if n > 1 else ''))
                self.types.append(v.get('type', 'continuous'))
                self.domains.append(tuple(v['domain']))
        for t in self.types:
            if t not in ('continuous', 'discrete'):
                raise NotImplementedError("variable type %r is outside the accelerated path" % t)
        self.dimensionality = len(self.names)
        self.model_dimensionality = self.dimensionality

    def has_constraints(self):
        """space.py:226-230."""
        return self.constraints is not None

    def indicator_constraints(self, x):
        """space.py:303-318: ones / zeros [M, 1]."""
        x = np.atleast_2d(x)
        I_x = np.ones((x.shape[0], 1))
        for fn, d in zip(self._constraint_fns, self.constraints or []):
            try:
                ind_x = (np.asarray(fn(x)) < 0) * 1
                I_x *= ind_x.reshape(x.shape[0], 1)
            except Exception:
                print('Fail to compile the constraint: ' + str(d))
                raise
        return I_x

    def get_bounds(self):
        return [(min(d), max(d)) for d in self.domains]

    def round_optimum(self, x):
        """space.py:328-349: discrete variables snap to the closest admissible value."""
        x = np.array(x, dtype=float).reshape(-1)
        for i, (t, d) in enumerate(zip(self.types, self.domains)):
            if t == 'discrete':
                dom = np.asarray(d, dtype=float)
                x[i] = dom[np.argmin(np.abs(dom - x[i]))]
            else:
                x[i] = min(max(x[i], d[0]), d[1])
        return x[None, :]

    # -- the fork's additions for the Gower kernel (space.py:351-362,436-445,483-492) -------------------
    def get_continuous_dims(self):
        return [i for i, t in enumerate(self.types) if t == 'continuous']

    def get_discrete_dims(self):
        return [i for i, t in enumerate(self.types) if t == 'discrete']

    def lengthscales(self):
        """Ranges of the continuous variables, in order (space.py:351-362)."""
        return [d[-1] - d[0] for t, d in zip(self.types, self.domains) if t == 'continuous']

    def samples_uniform(self, n, rng=np.random):
        """experiment_design/random_design.py:15-35: rejection sampling when the space has constraints."""
        if not self.has_constraints():
            return self._samples_box(n, rng)
        samples = np.empty((0, self.dimensionality))
        while samples.shape[0] < n:
            Z = self._samples_box(n, rng)
            ok = (self.indicator_constraints(Z) == 1).flatten()
            if ok.sum() > 0:
                samples = np.vstack((samples, Z[ok, :]))
        return samples[0:n, :]

    def _samples_box(self, n, rng=np.random):
        """experiment_design/random_design.py:37-65: every discrete variable first (one ``choice`` call each, in order), then
        every continuous one (one ``uniform`` call each, in order) -- the reference's consumption of the generator, so that a
        seeded run draws the reference's design whatever the order of the variables."""
        Z = np.empty((n, self.dimensionality))
        for i, (t, d) in enumerate(zip(self.types, self.domains)):
            if t == 'discrete':
                Z[:, i] = rng.choice(np.asarray(d, dtype=float), n)
        for i, (t, d) in enumerate(zip(self.types, self.domains)):
            if t != 'discrete':
                Z[:, i] = rng.uniform(d[0], d[1], n)
        return Z


class AcquisitionOptimizer(object):
    """optimization/acquisition_optimizer.py:46-77 with ObjectiveAnchorPointsGenerator (:85-98)."""

    def __init__(self, space, optimizer='lbfgs', num_samples=1000, num_anchor=5, maxiter=1000):
        self.space = space
        self.optimizer_name = optimizer
        self.num_samples = num_samples
        self.num_anchor = num_anchor
        self.maxiter = maxiter

    def optimize(self, f=None, df=None, f_df=None, duplicate_manager=None):
        X = self.space.samples_uniform(self.num_samples)
        scores = f(X).flatten()                       # ONE batched call (anchor_points_generator.py:59)
        anchors = X[np.argsort(scores)[:min(len(scores), self.num_anchor)], :]
        bounds = self.space.get_bounds()
        best_x, best_fx = None, np.inf
        for a in anchors:
            if f_df is not None and self.optimizer_name == 'lbfgs':
                def _f_df(x):                        # optimizer.py:45-47
                    fx, dfx = f_df(np.atleast_2d(x))
                    return float(np.asarray(fx).ravel()[0]), np.asarray(dfx, dtype=float).ravel()
                res = _sopt.fmin_l_bfgs_b(_f_df, x0=a, bounds=bounds, maxiter=self.maxiter)
            else:
                res = _sopt.fmin_l_bfgs_b(lambda x: float(f(np.atleast_2d(x)).ravel()[0]), x0=a, bounds=bounds,
                                          approx_grad=True, maxiter=self.maxiter)
            x = np.atleast_2d(res[0])
            if res[2].get('task', b'') in (b'ABNORMAL_TERMINATION_IN_LNSRCH', 'ABNORMAL_TERMINATION_IN_LNSRCH'):
                x = np.atleast_2d(a)                  # optimizer.py:53-59
            x = self.space.round_optimum(x)           # optimizer.py:130-168 (apply_optimizer)
            fx = float(f(x).ravel()[0])
            if fx < best_fx:
                best_x, best_fx = x, fx
        return best_x, best_fx


def _lbfgs_from_anchor(space, a, f, f_df, maxiter=1000):
    """optimizer.py:36-59 + apply_optimizer :130-168: one bounded L-BFGS run from anchor ``a``."""
    def _f_df(x):
        fx, dfx = f_df(np.atleast_2d(x))
        return float(np.asarray(fx).ravel()[0]), np.asarray(dfx, dtype=float).ravel()
    res = _sopt.fmin_l_bfgs_b(_f_df, x0=a, bounds=space.get_bounds(), maxiter=maxiter)
    x = np.atleast_2d(res[0])
    if res[2].get('task', b'') in (b'ABNORMAL_TERMINATION_IN_LNSRCH', 'ABNORMAL_TERMINATION_IN_LNSRCH'):
        x = np.atleast_2d(a)
    return space.round_optimum(x)


class ThompsonSamplingAnchorPointsGenerator(object):
    """optimization/anchor_points_generator.py:66-82: marginal Thompson sampling.  ONE batched device
    predict over ``num_samples`` random locations, one normal draw per location from its own (mean, sd),
    the ``num_anchor`` lowest draws are the anchors (:19-63)."""

    def __init__(self, space, design_type, model, num_samples=25000):
        self.space, self.design_type, self.model, self.num_samples = space, design_type, model, num_samples

    def get_anchor_point_scores(self, X):
        m, s = self.model.predict(X)
        return np.random.normal(m.ravel(), s.ravel())

    def get(self, num_anchor=5, duplicate_manager=None, unique=False, context_manager=None):
        X = self.space.samples_uniform(self.num_samples)
        scores = self.get_anchor_point_scores(X)
        return X[np.argsort(scores)[:min(len(scores), num_anchor)], :]


class RandomAnchorPointsGenerator(object):
    """anchor_points_generator.py:100-113: every sample scores the same, i.e. the first ``num_anchor`` rows."""

    def __init__(self, space, design_type, num_samples=10000):
        self.space, self.num_samples = space, num_samples

    def get(self, num_anchor=5, duplicate_manager=None, unique=False, context_manager=None):
        X = self.space.samples_uniform(self.num_samples)
        return X[np.argsort(np.arange(X.shape[0]), kind='stable')[:num_anchor], :]


class SamplingBasedBatchEvaluator(object):
    """core/evaluators/base.py:21-92 without the duplicate manager (de_duplication is host bookkeeping)."""

    def __init__(self, acquisition, batch_size):
        self.acquisition, self.batch_size, self.space = acquisition, batch_size, acquisition.space
        self.num_anchor = 5 * batch_size

    def compute_batch(self, duplicate_manager=None, context_manager=None):
        if duplicate_manager is not None or context_manager is not None:
            raise NotImplementedError("duplicate / context managers are outside the accelerated path")
        return self.compute_batch_without_duplicate_logic()


class ThompsonBatch(SamplingBasedBatchEvaluator):
    """core/evaluators/batch_thompson.py:9-55: anchors by marginal Thompson sampling on the model, each of
    the first ``batch_size`` anchors refined by L-BFGS on the acquisition."""

    def __init__(self, acquisition, batch_size):
        super(ThompsonBatch, self).__init__(acquisition, batch_size)
        self.model = acquisition.model
        self.f, self.f_df = acquisition.acquisition_function, acquisition.acquisition_function_withGradients

    def get_anchor_points(self):
        return ThompsonSamplingAnchorPointsGenerator(self.space, "random", self.model).get(num_anchor=self.num_anchor)

    def optimize_anchor_point(self, a):
        return _lbfgs_from_anchor(self.space, a, self.f, self.f_df)

    def compute_batch_without_duplicate_logic(self):
        anchors = self.get_anchor_points()
        return np.vstack([self.optimize_anchor_point(a) for a, _ in zip(anchors, range(self.batch_size))])


class RandomBatch(SamplingBasedBatchEvaluator):
    """core/evaluators/batch_random.py:9-48: first element = the optimised acquisition, the rest uniform."""

    def compute_batch_without_duplicate_logic(self):
        x, _ = self.acquisition.optimize()
        k = self.batch_size - 1
        anchors = RandomAnchorPointsGenerator(self.space, "random").get(num_anchor=self.num_anchor)
        return np.vstack([x] + [a for a, _ in zip(anchors, range(k))])


class BayesianOptimization(object):
    """GPyOpt.methods.BayesianOptimization for model_type='GP' (bayesian_optimization.py:76-170)."""

    def __init__(self, f, domain=None, constraints=None, cost_withGradients=None, model_type='GP', X=None, Y=None,
                 initial_design_numdata=5, initial_design_type='random', acquisition_type='EI', normalize_Y=True,
                 exact_feval=False, acquisition_optimizer_type='lbfgs', model_update_interval=1,
                 evaluator_type='sequential', batch_size=1, num_cores=1, verbosity=False, verbosity_model=False,
                 maximize=False, de_duplication=False, model=None, acquisition=None, device=0, **kwargs):
        if model_type not in ('GP',) and model is None:
            raise NotImplementedError("model_type %r is outside the accelerated path" % model_type)
        if evaluator_type not in ('sequential', 'local_penalization', 'thompson_sampling', 'random', None):
            raise NotImplementedError("evaluator %r is outside the accelerated path" % evaluator_type)
        self.evaluator_type = evaluator_type
        self.batch_size = batch_size
        self.f = f
        self.maximize = maximize
        self.space = Design_space(domain, constraints)
        self.normalize_Y = normalize_Y
        self.model_update_interval = model_update_interval
        self.verbosity = verbosity
        self.kwargs = kwargs
        # arguments_manager.py:78-147
        kernel = kwargs.get('kernel', None)
        if isinstance(kernel, str):
            kernel = {'RBF': _kern.RBF, 'Matern52': _kern.Matern52}[kernel](self.space.dimensionality,
                                                                            ARD=kwargs.get('ARD', False))
        self.model = model if model is not None else GPModel(
            kernel=kernel, noise_var=kwargs.get('noise_var', None), exact_feval=exact_feval,
            optimizer=kwargs.get('model_optimizer_type', 'lbfgs'), max_iters=kwargs.get('max_iters', 1000),
            optimize_restarts=kwargs.get('optimize_restarts', 5), verbose=verbosity_model,
            ARD=kwargs.get('ARD', False), Gower=kwargs.get('Gower', False), space=self.space, device=device)
        self.acquisition_optimizer = AcquisitionOptimizer(self.space, acquisition_optimizer_type)
        # arguments_manager.py:42-75
        jitter = kwargs.get('acquisition_jitter', 0.01)
        weight = kwargs.get('acquisition_weight', 2)
        if acquisition is not None:
            self.acquisition = acquisition
        elif acquisition_type in (None, 'EI'):
            self.acquisition = AcquisitionEI(self.model, self.space, self.acquisition_optimizer, cost_withGradients, jitter)
        elif acquisition_type == 'LCB':
            self.acquisition = AcquisitionLCB(self.model, self.space, self.acquisition_optimizer, None, weight)
        elif acquisition_type == 'MPI':
            self.acquisition = AcquisitionMPI(self.model, self.space, self.acquisition_optimizer, cost_withGradients, jitter)
        else:
            raise NotImplementedError("acquisition %r is outside the accelerated path" % acquisition_type)
        # arguments_manager.py:17-38 (evaluator_creator): local penalisation wraps the acquisition (LP.py)
        if batch_size == 1 or evaluator_type == 'sequential':
            self.evaluator = None                      # Sequential: one optimised point (arguments_manager.py:23-24)
        elif evaluator_type in ('random', None):
            self.evaluator = RandomBatch(self.acquisition, batch_size)
        elif evaluator_type == 'thompson_sampling':
            self.evaluator = ThompsonBatch(self.acquisition, batch_size)
        elif evaluator_type == 'local_penalization':
            if not isinstance(self.acquisition, AcquisitionLP):
                self.acquisition = AcquisitionLP(self.model, self.space, self.acquisition_optimizer, self.acquisition,
                                                 transform=kwargs.get('acquisition_transformation', 'none'))
            self.evaluator = LocalPenalization(self.acquisition, batch_size)
        else:
            self.evaluator = None
        # initial data
        if X is None:
            X = self.space.samples_uniform(initial_design_numdata)
        self.X = np.asarray(X, dtype=float)
        if Y is None:
            if f is None:
                raise ValueError("f=None requires X and Y")
            Y = self._evaluate(self.X)
        self.Y = np.asarray(Y, dtype=float).reshape(-1, 1)
        self.num_acquisitions = 0
        self.suggested_sample = None
        self.cum_time = 0.0

    def _evaluate(self, X):
        Y = np.asarray(self.f(X), dtype=float).reshape(-1, 1)
        return -Y if self.maximize else Y

    # -- core/bo.py:236-254 ----------------------------------------------------------------
    def _update_model(self, normalization_type='stats'):
        if self.num_acquisitions % self.model_update_interval == 0:
            Y_inmodel = normalize(self.Y, normalization_type) if self.normalize_Y else self.Y
            self.model.updateModel(self.X, Y_inmodel, None, None)

    # -- core/bo.py:216-234 ----------------------------------------------------------------
    def _compute_next_evaluations(self, pending_zipped_X=None, ignored_zipped_X=None):
        if self.evaluator is not None:
            return self.evaluator.compute_batch()
        x, _ = self.acquisition.optimize()
        return x

    def suggest_next_locations(self, context=None, pending_X=None, ignored_X=None):
        """core/bo.py:55-71."""
        if context is not None:
            raise NotImplementedError("context variables are host-side bookkeeping outside the accelerated path")
        self._update_model()
        self.suggested_sample = self._compute_next_evaluations(pending_X, ignored_X)
        return self.suggested_sample

    def run_optimization(self, max_iter=0, max_time=np.inf, eps=1e-8, context=None, verbosity=False, **kw):
        """core/bo.py:73-168 (sequential evaluator)."""
        if self.f is None:
            raise ValueError("Cannot run the optimization loop without the objective function")
        t0 = time.time()
        it = 0
        while (time.time() - t0) < max_time:
            try:
                self._update_model()
            except np.linalg.LinAlgError:
                break                                  # bo.py:134-137
            # bo.py:139-141: the model is refreshed first, then the budget and the distance between the LAST TWO
            # evaluated points decide (the point that triggers the rule has been evaluated and is kept)
            if it >= max_iter or (self.X.shape[0] > 1 and
                                  np.sqrt(np.sum((self.X[-1, :] - self.X[-2, :]) ** 2)) <= eps):
                break
            x = self._compute_next_evaluations()
            y = self._evaluate(x)
            self.X = np.vstack((self.X, x))
            self.Y = np.vstack((self.Y, y))
            self.num_acquisitions += 1
            it += 1
            if verbosity or self.verbosity:
                print("num acquisition: %d, time elapsed: %.2fs" % (self.num_acquisitions, time.time() - t0))
        self.cum_time = time.time() - t0
        i = int(np.argmin(self.Y))
        self.x_opt, self.fx_opt = self.X[i], float(self.Y[i, 0])
        return self.x_opt, self.fx_opt
