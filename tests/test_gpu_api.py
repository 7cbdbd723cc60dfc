"""GPU: the host mirror of the reference API (GPRegression / GPModel / Acquisition* / BayesianOptimization).

These follow the reference's own tests for the path: pinv closed form
(GPy/GPy/testing/model_tests.py:63-82), var >= 0 stress (:25-61), normaliser equivalence (:84-119),
checkgrad-style finite differences (:664-723, kernel_tests.py:414-422), set_XY round trip
(gp_tests.py:50-60).
"""
import numpy as np
import pytest

import gaussian_process_optimization_amd as gpo
from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu


def test_raw_predict_pinv_closed_form():
    rng = np.random.RandomState(0)
    N, M = 20, 50
    X = rng.randn(N, 1); Y = np.sin(X) + rng.randn(N, 1) * 0.05; Xn = rng.randn(M, 1)
    k = gpo.kern.RBF(1)
    m = gpo.models.GPRegression(X, Y, kernel=k, noise_var=0.5)
    ko = O.RBF(1, 1.0, 1.0)
    Kinv = np.linalg.pinv(ko.K(X) + np.eye(N) * 0.5)
    mu_hat = ko.K(Xn, X).dot(Kinv).dot(Y)
    cov_hat = ko.K(Xn) - ko.K(Xn, X).dot(Kinv).dot(ko.K(X, Xn))
    mu, cov = m.predict_noiseless(Xn, full_cov=True)
    assert mu.shape == (M, 1) and cov.shape == (M, M)
    np.testing.assert_almost_equal(mu_hat, mu)
    np.testing.assert_almost_equal(cov_hat, cov)
    mu, var = m.predict_noiseless(Xn)
    np.testing.assert_almost_equal(np.diag(cov_hat)[:, None], var)
    m.close()


def test_posterior_covariance_between_points():
    """GP.posterior_covariance_between_points (gp.py:714-721) / GPModel.get_covariance_between_points
    (gpmodel.py:173-177) against the oracle's restatement of posterior.py:109-128."""
    rng = np.random.RandomState(3)
    N, D = 150, 3
    X = rng.rand(N, D); Y = np.sin(3 * X.sum(1, keepdims=True)) + 0.1 * rng.randn(N, 1)
    X1, X2 = rng.rand(7, D), rng.rand(11, D)
    for kname, cls in (("rbf", gpo.kern.RBF), ("Mat52", gpo.kern.Matern52)):
        m = gpo.models.GPRegression(X, Y, kernel=cls(D, variance=1.4, lengthscale=0.6), noise_var=0.05)
        gp = O.OracleGP(X, Y, O.make_kernel(kname, D, 1.4, np.array([0.6]), ARD=False), 0.05)
        C = m.posterior_covariance_between_points(X1, X2)
        C0 = gp.posterior_covariance_between_points(X1, X2)
        assert C.shape == (7, 11)
        assert np.max(np.abs(C - C0)) <= 1e-6 * max(1.0, np.max(np.abs(C0)))
        # a point with itself: the off-diagonal block's diagonal is the noiseless predictive variance
        Cs = m.posterior_covariance_between_points(X1, X1)
        _, v = m.predict_noiseless(X1)
        assert np.max(np.abs(np.diag(Cs)[:, None] - v)) <= 1e-9
        m.close()


def test_raw_predict_numerical_stability():
    rs = np.random.RandomState(3)
    x1, x2 = np.meshgrid(np.linspace(-5, 10, 5), np.linspace(0, 15, 5))
    X = np.c_[x1.ravel(), x2.ravel()]
    Y = ((X[:, 1] - 5.1 / (4 * np.pi ** 2) * X[:, 0] ** 2 + 5 * X[:, 0] / np.pi - 6) ** 2
         + 10 * (1 - 1 / (8 * np.pi)) * np.cos(X[:, 0]) + 10)[:, None]
    m = gpo.models.GPRegression(X, Y, gpo.kern.RBF(2, 2.0, [5.0, 5.0], ARD=True), noise_var=1e-5)
    Xt = np.c_[rs.uniform(-5, 10, 100000), rs.uniform(0, 15, 100000)]
    _, v = m.predict(Xt)
    assert (v >= 0).all()
    m.close()


def test_normalizer_equivalence():
    rng = np.random.RandomState(1)
    X = rng.rand(30, 2); Y = rng.randn(30, 1) * 4 + 7
    m1 = gpo.models.GPRegression(X, Y, gpo.kern.RBF(2, 1.0, 0.5), normalizer=True, noise_var=0.1)
    Ys = (Y - Y.mean(0)) / Y.std(0)
    m2 = gpo.models.GPRegression(X, Ys, gpo.kern.RBF(2, 1.0, 0.5), noise_var=0.1)
    a, va = m1.predict(X[:7]); b, vb = m2.predict(X[:7])
    np.testing.assert_allclose(a, b * Y.std(0) + Y.mean(0), rtol=1e-10)
    np.testing.assert_allclose(va, vb * Y.std(0) ** 2, rtol=1e-10)
    assert m1.log_likelihood() == m2.log_likelihood()
    m1.close(); m2.close()


@pytest.mark.parametrize("cls", ["RBF", "Matern52"])
@pytest.mark.parametrize("ard", [False, True])
def test_checkgrad(cls, ard):
    rng = np.random.RandomState(2)
    D = 2
    X = rng.rand(40, D); Y = rng.randn(40, 1)
    k = getattr(gpo.kern, cls)(D, 1.3, [0.4, 0.7] if ard else 0.5, ARD=ard)
    m = gpo.models.GPRegression(X, Y, k, noise_var=0.2)
    x0 = m.optimizer_array.copy()
    g = m.objective_function_gradients()
    for i in range(x0.size):
        e = np.zeros_like(x0); e[i] = 1e-6
        m.optimizer_array = x0 + e; fp = m.objective_function()
        m.optimizer_array = x0 - e; fm = m.objective_function()
        assert abs((fp - fm) / 2e-6 - g[i]) <= 1e-5 * max(1.0, abs(g[i]))
    m.optimizer_array = x0
    xs = rng.rand(3, D)
    dm, dv = m.predictive_gradients(xs)
    assert dm.shape == (3, D, 1) and dv.shape == (3, D)
    for q in range(D):
        e = np.zeros((1, D)); e[0, q] = 1e-6
        mp, vp = m.predict(xs + e); mm, vm = m.predict(xs - e)
        np.testing.assert_allclose(dm[:, q, 0], ((mp - mm) / 2e-6)[:, 0], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(dv[:, q], ((vp - vm) / 2e-6)[:, 0], rtol=1e-4, atol=1e-6)
    m.close()


def test_setxy_roundtrip_and_optimize():
    np.random.seed(12345)
    X = np.random.rand(60, 2); Y = np.sin(3 * X[:, :1]) + 0.1 * np.random.randn(60, 1)
    m = gpo.models.GPRegression(X, Y, gpo.kern.Matern52(2), noise_var=0.5)
    l0 = m.log_likelihood()
    m.set_XY(X[:30], Y[:30])
    assert m.log_likelihood() != l0
    m.set_XY(X, Y)
    assert m.log_likelihood() == l0
    m.optimize(max_iters=200)
    assert m.log_likelihood() > l0 + 1.0
    p = m.param_array.copy()
    assert (p > 0).all()
    runs = m.optimize_restarts(num_restarts=2, verbose=False, max_iters=50)
    assert len(runs) == 2 and m.objective_function() <= min(r[0] for r in runs) + 1e-9
    m.close()


def _oracle_twin(X, Y, kname, ls, var, noise):
    gp = O.OracleGP(X, Y, O.make_kernel(kname, X.shape[1], var, ls, ARD=np.size(ls) > 1), noise)
    return gp, O.OracleGPModel(gp)


def test_gpmodel_and_acquisitions_match_reference_formulas():
    X, Y, Xs = O.synthetic_problem(150, 3, 400, seed=21)
    gm = gpo.GPModel(kernel=gpo.kern.Matern52(3, 1.2, 0.6), noise_var=0.03, max_iters=0, verbose=False)
    gm.updateModel(X, Y, None, None)
    gp0, gm0 = _oracle_twin(X, Y, "Mat52", [0.6], 1.2, 0.03)
    # GPModel defaults put a bounded constraint on the noise: value unchanged
    assert float(gm.model.Gaussian_noise.variance) == pytest.approx(0.03, rel=1e-12)
    m, s = gm.predict(Xs); m0, s0 = gm0.predict(Xs)
    np.testing.assert_allclose(m, m0, rtol=1e-6, atol=1e-9); np.testing.assert_allclose(s, s0, rtol=1e-6)
    assert gm.get_fmin() == pytest.approx(gm0.get_fmin(), rel=1e-6)
    mm, ss, dm, ds = gm.predict_withGradients(Xs[:20]); r = gm0.predict_withGradients(Xs[:20])
    for a, b in zip((mm, ss, dm, ds), r):
        np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-8)
    for cls, fn, fng, kw in ((gpo.AcquisitionEI, O.acq_EI, O.acq_EI_withGradients, dict(jitter=0.01)),
                             (gpo.AcquisitionLCB, O.acq_LCB, O.acq_LCB_withGradients, dict(exploration_weight=2)),
                             (gpo.AcquisitionMPI, O.acq_MPI, O.acq_MPI_withGradients, dict(jitter=0.01))):
        acq = cls(gm, **kw)
        par = list(kw.values())[0]
        ref = -fn(gm0, Xs, par)
        a = acq.acquisition_function(Xs)                    # device batched path
        assert a.shape == (400, 1)
        np.testing.assert_allclose(a, ref, rtol=1e-6, atol=1e-9 * np.max(np.abs(ref)))
        host = -acq._compute_acq(Xs)                         # reference host formula on device predictions
        np.testing.assert_allclose(host, ref, rtol=1e-6, atol=1e-9 * np.max(np.abs(ref)))
        f, df = acq.acquisition_function_withGradients(Xs[:20])
        f0, df0 = fng(gm0, Xs[:20], par)
        np.testing.assert_allclose(f, -f0, rtol=1e-6, atol=1e-9 * np.max(np.abs(ref)))
        np.testing.assert_allclose(df, -df0, rtol=1e-5, atol=1e-8 * np.max(np.abs(df0)))
        i, v = acq.argbest(Xs, -1)
        assert i == int(np.argmin(a)) and v == a[i, 0]
        i, v = acq.argbest(Xs, +1)                           # run.py:1241 takes argmax of the same vector
        assert i == int(np.argmax(a))
    gm.model.close()


def test_acquisition_with_normalizer_on_device():
    X, Y, Xs = O.synthetic_problem(120, 2, 300, seed=5, standardize=False)
    Y = 10 * Y + 3
    gm = gpo.GPModel(kernel=gpo.kern.RBF(2, 1.0, 0.4), noise_var=0.05, max_iters=0, verbose=False)
    gm.model = gpo.models.GPRegression(X, Y, gpo.kern.RBF(2, 1.0, 0.4), normalizer=True, noise_var=0.05)
    gp0 = O.OracleGP(X, Y, O.RBF(2, 1.0, 0.4), 0.05, normalizer=True)
    gm0 = O.OracleGPModel(gp0)
    acq = gpo.AcquisitionEI(gm, jitter=0.01)
    ref = -O.acq_EI(gm0, Xs, 0.01)
    np.testing.assert_allclose(acq.acquisition_function(Xs), ref, rtol=1e-6, atol=1e-9 * np.max(np.abs(ref)))
    gm.model.close()


def test_bayesian_optimization_loop_improves():
    np.random.seed(0)
    f = lambda x: np.sum((x - 0.3) ** 2, axis=1, keepdims=True)  # noqa: E731
    dom = [{'name': 'x', 'type': 'continuous', 'domain': (0, 1), 'dimensionality': 2}]
    X0 = np.random.rand(6, 2)
    bo = gpo.methods.BayesianOptimization(f=f, domain=dom, X=X0, Y=f(X0), acquisition_type='EI', exact_feval=True,
                                          optimize_restarts=1, max_iters=100)
    x1 = bo.suggest_next_locations()
    assert x1.shape == (1, 2) and (x1 >= 0).all() and (x1 <= 1).all()
    xo, fo = bo.run_optimization(max_iter=6)
    assert fo <= f(X0).min() + 1e-12 and fo < 0.02
    # user-table pattern of run.py:1240-1241: score a fixed candidate table, take the arg-best
    table = np.random.rand(5000, 2)
    vals = bo.acquisition.acquisition_function(table)
    i, v = bo.acquisition.argbest(table, -1)
    assert i == int(np.argmin(vals))
    bo.model.model.close()


def test_rccl_comm_single_rank_and_sharded_wrapper():
    """The RCCL entry points with a one-rank communicator (what a 1-GPU box can exercise): unique id, init,
    all-gather of the (value, index) pair, broadcast of the fit, and the ShardedCandidates wrapper on top."""
    from gaussian_process_optimization_amd import _lib
    from gaussian_process_optimization_amd.sharded import RcclCollective, ShardedCandidates
    X, Y, Xs = O.synthetic_problem(300, 3, 1000, seed=2)
    h = _lib.Handle(0)
    h.set_data(X, Y)
    h.set_params(_lib.GP_KERNEL_MATERN52, 0, 1.0, [0.5], 1e-2)
    h.fit()
    uid = h.comm_unique_id()
    assert len(uid) == 128
    h.comm_init(uid, 0, 1)
    vals, idxs = h.comm_allgather_best(-1.25, 77, 1)
    assert vals.tolist() == [-1.25] and idxs.tolist() == [77]
    lml0 = h.fit()[0]
    h.comm_bcast_fit(0)           # root == self: state must survive unchanged
    h.set_candidates(Xs)
    m1, v1 = h.predict(True)
    gp = O.OracleGP(X, Y, O.Matern52(3, 1.0, 0.5), 1e-2)
    m0, v0 = gp.predict(Xs)
    np.testing.assert_allclose(m1, m0, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(v1, v0, rtol=1e-6)

    def score_local(Xb, sense):
        h.set_candidates(Xb)
        return h.acq_argbest(_lib.GP_ACQ_LCB, 2.0, 0.0, sense)
    sc = ShardedCandidates(0, 1, RcclCollective(h, 1))
    gi, gv = sc.argbest(Xs, score_local, -1)
    a = h.acq(_lib.GP_ACQ_LCB, 2.0, 0.0)[:, 0]
    assert gi == int(np.argmin(a)) and gv == a[gi]
    h.lib.gp_comm_destroy(h.h)
    h.close()


@pytest.mark.parametrize("base", ["EI", "LCB", "MPI"])
def test_local_penalization_matches_reference_formulas(base):
    """AcquisitionLP on the device (gp_acq_lp) vs the restated LP.py formulas on oracle predictions."""
    X, Y, table = O.synthetic_problem(180, 3, 1500, seed=33)
    gm = gpo.GPModel(kernel=gpo.kern.Matern52(3, 1.1, 0.5), noise_var=0.02, max_iters=0, verbose=False)
    gm.updateModel(X, Y, None, None)
    gp0, gm0 = _oracle_twin(X, Y, "Mat52", [0.5], 1.1, 0.02)
    cls = {"EI": gpo.AcquisitionEI, "LCB": gpo.AcquisitionLCB, "MPI": gpo.AcquisitionMPI}[base]
    fn0 = {"EI": lambda x: O.acq_EI(gm0, x, 0.01), "LCB": lambda x: O.acq_LCB(gm0, x, 2.0),
           "MPI": lambda x: O.acq_MPI(gm0, x, 0.01)}[base]
    space = gpo.Design_space([{'name': 'x', 'type': 'continuous', 'domain': (0, 1), 'dimensionality': 3}])
    lp = gpo.AcquisitionLP(gm, space, None, cls(gm, space))
    tr = "softplus" if base == "LCB" else "none"
    assert lp.transform == tr
    # un-penalised
    v = lp.acquisition_function(table)
    ref = O.lp_penalized_acquisition(-fn0(table), table, None, None, None, tr)
    np.testing.assert_allclose(v, ref, rtol=1e-6, atol=1e-9)
    # penalised with a 3-point batch; L and Min as compute_batch would supply them
    Xb = table[[5, 200, 900]]
    L, Min = 3.7, float(Y.min())
    lp.update_batches(Xb, L, Min)
    r0, s0 = O.lp_hammer_precompute(gm0, Xb, L, Min)
    np.testing.assert_allclose(lp.r_x0, r0, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(lp.s_x0, s0, rtol=1e-6)
    v = lp.acquisition_function(table)
    ref = O.lp_penalized_acquisition(-fn0(table), table, Xb, r0, s0, tr)
    np.testing.assert_allclose(v, ref, rtol=1e-6, atol=1e-8)
    host = lp._score_on_host(table)      # reference formulas on device predictions
    np.testing.assert_allclose(v, host, rtol=1e-9, atol=1e-9)
    # arg-best with exclusion == masked argmax of the reference vector (run.py:1249-1252)
    taken = [int(np.argmax(v)), 17]
    i, val = lp.argbest(table, +1, exclude=taken)
    mv = np.ma.array(v, mask=False); mv.mask[taken] = True
    assert i == int(np.argmax(mv)) and val == v[i]
    # gradients: the reference's own formula (LP.py:91-133), which drops the direction factor of the penaliser
    # gradient -- kept bug-compatible, so compared with its restatement rather than with finite differences
    xq = table[:4] * 0.8 + 0.1
    f, df = lp.acquisition_function_withGradients(xq)
    fng = {"EI": lambda x: O.acq_EI_withGradients(gm0, x, 0.01), "LCB": lambda x: O.acq_LCB_withGradients(gm0, x, 2.0),
           "MPI": lambda x: O.acq_MPI_withGradients(gm0, x, 0.01)}[base]
    a0, da0 = fng(xq)
    ref_d = O.lp_d_acquisition(-a0, -da0, xq, Xb, r0, s0, tr)
    np.testing.assert_allclose(df, ref_d, rtol=1e-5, atol=1e-7 * np.max(np.abs(ref_d)))
    gm.model.close()


@pytest.mark.parametrize("base", ["EI", "LCB"])
def test_local_penalization_batch_from_table(base):
    """The thesis driver's loop (run.py:1234-1258) on a fixed table: the rows the device loop picks == the rows the oracle's
    restatement of that loop picks (O.lp_table_batch over O.OracleLP, pinned to the reference's verbatim AcquisitionLP in
    oracle/pin_against_reference.py), same numpy seed for estimate_L; EI (plain log) and LCB (softplus)."""
    X, Y, table = O.synthetic_problem(120, 2, 4000, seed=12)
    gm = gpo.GPModel(kernel=gpo.kern.Matern52(2, 1.0, 0.3), noise_var=0.01, max_iters=0, verbose=False)
    gm.updateModel(X, Y, None, None)
    gp0, gm0 = _oracle_twin(X, Y, "Mat52", [0.3], 1.0, 0.01)
    space = gpo.Design_space([{'name': 'x', 'type': 'continuous', 'domain': (0, 1), 'dimensionality': 2}])
    cls = {"EI": gpo.AcquisitionEI, "LCB": gpo.AcquisitionLCB}[base]
    lp = gpo.AcquisitionLP(gm, space, None, cls(gm, space))
    lp0 = O.OracleLP(gm0, space, base)
    np.random.seed(3)
    want, L0, Min0 = O.lp_table_batch(lp0, table, 5)
    np.random.seed(3)
    L = gpo.estimate_L(gm.model, space.get_bounds())
    assert abs(L - L0) <= 1e-3 * L0       # polished by forward differences of step 1e-8: see test_gpu_gower.py
    np.random.seed(3)
    chosen = gpo.LocalPenalization(lp, 5).compute_batch_from_table(table, sense=+1)
    assert len(set(chosen)) == 5
    final = lp0.acquisition_function(table)               # the oracle's last penalised score vector
    for got, ref in zip(chosen, want):
        assert got == ref or abs(final[got] - final[ref]) <= 1e-6 * max(1.0, abs(final[ref])), (chosen, want)
    # with the oracle's L handed in the rows are the oracle's whatever the polish did
    assert gpo.LocalPenalization(lp, 5).compute_batch_from_table(table, sense=+1, lipschitz=L0) == want
    gm.model.close()


def test_bayesian_optimization_local_penalization_batch():
    """The constructor arguments of run.py:1207-1224 (evaluator_type='local_penalization', batch of suggestions)."""
    np.random.seed(1)
    f = lambda x: np.sum((x - 0.6) ** 2, axis=1, keepdims=True)  # noqa: E731
    dom = [{'name': 'x', 'type': 'continuous', 'domain': (0, 1), 'dimensionality': 2}]
    X0 = np.random.rand(8, 2)
    bo = gpo.methods.BayesianOptimization(f=None, domain=dom, X=X0, Y=f(X0), model_type='GP', acquisition_type='EI',
                                          normalize_Y=True, exact_feval=True, evaluator_type='local_penalization',
                                          batch_size=3, optimize_restarts=1, max_iters=50)
    Xn = bo.suggest_next_locations()
    assert Xn.shape == (3, 2) and (Xn >= 0).all() and (Xn <= 1).all()
    assert len({tuple(np.round(r, 6)) for r in Xn}) == 3
    # the candidate-table loop of run.py:1234-1258 on the same acquisition
    table = np.random.rand(3000, 2)
    chosen = bo.evaluator.compute_batch_from_table(table, sense=+1)
    assert len(set(chosen)) == 3
    bo.model.model.close()


def test_bayesian_optimization_thompson_and_random_batches():
    """evaluator_type='thompson_sampling' / 'random' (arguments_manager.py:26-30; batch_thompson.py, batch_random.py):
    marginal Thompson anchors come from ONE batched device predict over 25 000 samples."""
    np.random.seed(2)
    f = lambda x: np.sum((x - 0.3) ** 2, axis=1, keepdims=True)  # noqa: E731
    dom = [{'name': 'x', 'type': 'continuous', 'domain': (0, 1), 'dimensionality': 2}]
    X0 = np.random.rand(10, 2)
    for ev in ('thompson_sampling', 'random'):
        bo = gpo.methods.BayesianOptimization(f=None, domain=dom, X=X0, Y=f(X0), model_type='GP', acquisition_type='EI',
                                              normalize_Y=True, exact_feval=True, evaluator_type=ev, batch_size=4,
                                              optimize_restarts=1, max_iters=50)
        Xn = bo.suggest_next_locations()
        assert Xn.shape == (4, 2) and (Xn >= 0).all() and (Xn <= 1).all()
        if ev == 'thompson_sampling':
            gen = gpo.bayesian_optimization.ThompsonSamplingAnchorPointsGenerator(bo.space, 'random', bo.model, 5000)
            Xs = bo.space.samples_uniform(5000)
            m, s = bo.model.predict(Xs)
            np.random.seed(7)
            draws = gen.get_anchor_point_scores(Xs)
            np.random.seed(7)
            # anchor_points_generator.py:79-82: one normal(m_i, s_i) draw per location, in order
            ref = np.array([np.random.normal(mi, si) for mi, si in zip(m, s)]).flatten()
            assert np.allclose(draws, ref, rtol=0, atol=1e-12)
        bo.model.model.close()
    # batch_size == 1 falls back to the sequential evaluator whatever evaluator_type says (arguments_manager.py:23)
    bo = gpo.methods.BayesianOptimization(f=None, domain=dom, X=X0, Y=f(X0), evaluator_type='thompson_sampling',
                                          batch_size=1, optimize_restarts=1, max_iters=20)
    assert bo.evaluator is None and bo.suggest_next_locations().shape == (1, 2)
    bo.model.model.close()


def test_bayesian_optimization_with_string_constraints():
    """constraints=[{'name', 'constraint'}] (space.py:303-318): the acquisition is zeroed outside the feasible set
    (base.py:33-39, host epilogue over the device posterior) and the suggestion is feasible."""
    np.random.seed(5)
    f = lambda x: np.sum((x - 0.8) ** 2, axis=1, keepdims=True)  # noqa: E731  (unconstrained optimum is infeasible)
    dom = [{'name': 'x', 'type': 'continuous', 'domain': (0, 1), 'dimensionality': 2}]
    cons = [{'name': 'c', 'constraint': 'x[:,0] + x[:,1] - 1'}]
    bo = gpo.methods.BayesianOptimization(f=f, domain=dom, constraints=cons, initial_design_numdata=6,
                                          acquisition_type='EI', exact_feval=True, optimize_restarts=1, max_iters=50)
    assert (bo.X.sum(1) < 1).all()                       # the initial design is drawn by rejection
    xn = bo.suggest_next_locations()
    assert bo.space.indicator_constraints(xn)[0, 0] == 1.0
    Z = np.random.rand(400, 2)
    a = bo.acquisition.acquisition_function(Z)
    assert (a[Z.sum(1) >= 1] == 0).all() and (a[Z.sum(1) < 1] <= 0).all()
    bo.model.model.close()


def test_gower_mixed_variable_kernel():
    """The fork's only numerical change (stationary.py:116-135): product of 1-D kernels, |dx|/range on continuous
    variables, Hamming on discrete ones; Kdiag stays `variance` as in the fork."""
    rng = np.random.default_rng(4)
    dom = [{'name': 'a', 'type': 'discrete', 'domain': (0, 1, 2, 3)},
           {'name': 'x', 'type': 'continuous', 'domain': (-2.0, 5.0)},
           {'name': 'b', 'type': 'discrete', 'domain': (10, 20)},
           {'name': 'y', 'type': 'continuous', 'domain': (0.0, 0.5)}]
    space = gpo.Design_space(dom)
    N, M = 260, 90
    def draw(n):
        return np.c_[rng.integers(0, 4, n), rng.uniform(-2, 5, n), rng.choice([10, 20], n), rng.uniform(0, 0.5, n)].astype(float)
    X, Xs = draw(N), draw(M)
    Xs[:5] = X[:5]                                   # exact matches exercise r = 0 on every factor
    Y = (np.sin(X[:, 1]) + 0.3 * X[:, 0] - 0.1 * (X[:, 2] == 20) + 2 * X[:, 3])[:, None] + 0.05 * rng.standard_normal((N, 1))
    for cls, name, var in ((gpo.kern.Matern52, "Mat52", 1.0), (gpo.kern.RBF, "rbf", 1.3)):
        k = cls(4, variance=var, Gower=True, space=space)
        ko = O.make_kernel(name, 4, var, None, Gower=True, space=space)
        K = k.K(X)
        K0 = ko.K(X)
        np.testing.assert_allclose(K, K0, rtol=1e-13, atol=1e-15)
        assert abs(K[0, 0] - var ** 4) < 1e-13             # product of four factors at r = 0
        m = gpo.models.GPRegression(X, Y, k, noise_var=0.05)
        gp = O.OracleGP(X, Y, ko, 0.05)
        assert abs(m.log_likelihood() - gp.log_likelihood()) <= 1e-8 * abs(gp.log_likelihood())
        mu, v = m.predict(Xs)
        mu0, v0 = gp.predict(Xs)
        np.testing.assert_allclose(mu, mu0, rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(v, v0, rtol=1e-6, atol=1e-9)
        # the fork's pairing: Gower K inside Euclidean gradients_X on the kernel's own lengthscale (gp.py:407-454 over
        # stationary.py:336-364); rows 0..4 sit on training points (inverse distance 0 there, stationary.py:251-258)
        dm, dv = m.predictive_gradients(Xs[:12])
        dm0, dv0 = gp.predictive_gradients(Xs[:12])
        np.testing.assert_allclose(dm, dm0, rtol=1e-6, atol=1e-6 * np.max(np.abs(dm0)))
        np.testing.assert_allclose(dv, dv0, rtol=1e-6, atol=1e-6 * np.max(np.abs(dv0)))
        m.close()
    # through GPyOpt's front door, as run.py:1207-1224 builds it (Gower=True, exact_feval=True, no optimisation here)
    gm = gpo.GPModel(exact_feval=True, max_iters=0, Gower=True, space=space, verbose=False)
    gm.updateModel(X, Y, None, None)
    acq = gpo.AcquisitionEI(gm, space)
    a = acq.acquisition_function(Xs)
    gp = O.OracleGP(X, Y, O.make_kernel("Mat52", 4, 1.0, None, Gower=True, space=space), 1e-6)
    ref = -O.acq_EI(O.OracleGPModel(gp), Xs, 0.01)
    np.testing.assert_allclose(a, ref, rtol=1e-4, atol=1e-6 * np.max(np.abs(ref)))
    # hyper-parameter optimisation runs on finite differences of the device LML and must not lower it
    l0 = gm.model.log_likelihood()
    gm.model.optimize(max_iters=15)
    assert gm.model.log_likelihood() >= l0 - 1e-6
    gm.model.close()


@pytest.mark.parametrize("kname,ard", [("RBF", False), ("RBF", True), ("Matern52", False), ("Matern52", True)])
def test_checkgrad_as_the_reference_tests_use_it(kname, ard):
    """The reference pins its gradients by `assert m.checkgrad()` (GPy/GPy/testing/model_tests.py:684-723 on GPRegression with
    RBF / Matern-5/2, iso and ARD; kernel_tests.py:414-422): the same call on the HIP model, at the initial and at randomised
    hyper-parameters, silent and verbose; with a fixed noise (GPyOpt's exact_feval) too; and a wrong gradient fails it."""
    np.random.seed(7)
    X = np.random.uniform(-3., 3., (60, 3))
    Y = np.sin(X[:, :1]) + 0.5 * np.cos(2 * X[:, 1:2]) + 0.05 * np.random.randn(60, 1)
    k = getattr(gpo.kern, kname)(3, ARD=ard)
    m = gpo.models.GPRegression(X, Y, k)
    assert m.checkgrad()
    m.randomize()
    assert m.checkgrad() and m.checkgrad(verbose=True)
    m.Gaussian_noise.constrain_fixed(1e-3, warning=False)
    assert m.checkgrad()
    x0 = m.optimizer_array.copy()
    m.checkgrad()
    assert np.allclose(m.optimizer_array, x0, rtol=1e-13, atol=0)    # the model is left where it was (up to the transform round trip)
    real = m._obj_grad
    m._obj_grad = lambda x: (real(x)[0], 1.5 * real(x)[1])   # a gradient off by a factor
    assert not m.checkgrad()
    m._obj_grad = real
    m.close()


def test_mean_gradients_alone_equal_the_first_output_of_predictive_gradients():
    """gp_predict_grad with dvdx = NULL (GPRegression.mean_gradients; what estimate_L asks for on 500 + N points) == the first
    output of the full call, for a batch and for a handful of rows, and needs no Ky^-1."""
    X, Y, Xs = O.synthetic_problem(700, 3, 300, seed=8)
    m = gpo.models.GPRegression(X, Y, gpo.kern.Matern52(3, 1.2, [0.4, 0.5, 0.6], ARD=True), noise_var=0.02)
    m._h.profile(True)
    jm = m.mean_gradients(Xs)
    assert "potri_lauum" not in [p["name"] for p in m._h.phases()]
    m._h.profile(False)
    dm, dv = m.predictive_gradients(Xs)
    assert jm.shape == dm.shape == (300, 3, 1) and np.array_equal(jm, dm)
    np.testing.assert_allclose(m.mean_gradients(Xs[:3]), dm[:3], rtol=0, atol=1e-10 * np.max(np.abs(dm)))
    assert m.mean_gradients(np.empty((0, 3))).shape == (0, 3, 1)
    gp0 = O.OracleGP(X, Y, O.make_kernel("Mat52", 3, 1.2, [0.4, 0.5, 0.6], ARD=True), 0.02)
    np.testing.assert_allclose(jm, gp0.predictive_gradients(Xs)[0], rtol=0, atol=1e-6 * np.max(np.abs(dm)))
    m.close()


def test_empty_candidate_set_and_shape_errors():
    """Zero prediction rows give the empty arrays NumPy gives the reference; wrong column counts raise."""
    rng = np.random.RandomState(0)
    X = rng.rand(40, 3); Y = rng.randn(40, 1)
    m = gpo.models.GPRegression(X, Y, kernel=gpo.kern.RBF(3), noise_var=0.1)
    mu, var = m.predict(np.empty((0, 3)))
    assert mu.shape == (0, 1) and var.shape == (0, 1)
    mu, cov = m.predict(np.empty((0, 3)), full_cov=True)
    assert mu.shape == (0, 1) and cov.shape == (0, 0)
    dm, dv = m.predictive_gradients(np.empty((0, 3)))
    assert dm.shape == (0, 3, 1) and dv.shape == (0, 3)
    with pytest.raises(ValueError):
        m.predict(rng.rand(5, 2))
    # a single 1-D location is promoted to one row (gpmodel.py:96-97)
    mu, var = m.predict(rng.rand(3))
    assert mu.shape == (1, 1) and var.shape == (1, 1)
    m.close()


def test_one_call_fit_predict_path_in_the_host_mirror():
    """A prediction or an acquisition on a model with a pending refit goes through gp_fit_predict; the numbers are
    bitwise those of an explicit fit followed by the same prediction."""
    rng = np.random.RandomState(1)
    N, D, M = 900, 4, 300
    X = rng.rand(N, D); Y = np.sin(4 * X.sum(1, keepdims=True)) + 0.05 * rng.randn(N, 1); Xs = rng.rand(M, D)
    m = gpo.models.GPRegression(X, Y, kernel=gpo.kern.Matern52(D, lengthscale=0.5), noise_var=0.01)
    m.kern.lengthscale[:] = 0.45            # -> dirty
    assert m._dirty
    mu1, v1 = m.predict(Xs)                 # one call
    names = [p["name"] for p in m._h.phases()]
    # the pipelined entry point (true fp64: one phase "cholesky+cand_solve") or, with GPHIP_EMULATE_FP64=1, the
    # factorisation followed by the emulated solve (the phases of the last call, gp_predict, are what is left)
    assert any("cand_solve" in n for n in names) and not m._dirty
    m.kern.lengthscale[:] = 0.45
    m._dirty = True
    lml = m.log_likelihood()                # explicit fit ...
    mu2, v2 = m.predict(Xs)                 # ... then predict
    assert np.array_equal(mu1, mu2) and np.array_equal(v1, v2) and np.isfinite(lml)
    # acquisition through the device fast path, refit pending
    gm = gpo.GPModel(exact_feval=False, optimize_restarts=1, max_iters=5, verbose=False)
    gm.updateModel(X, Y, None, None)
    acq = gpo.acquisitions.AcquisitionEI(gm, None, None, None, 0.01)
    a1 = acq.acquisition_function(Xs)
    gm.model._dirty = True                  # force the pending-refit state with the same hyper-parameters
    a2 = acq.acquisition_function(Xs)
    assert np.array_equal(a1, a2)
    m.close(); gm.model.close()
