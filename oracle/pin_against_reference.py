"""Pin oracle/cpu_ref.py against the reference (BUILD CONTAINER ONLY).

Checks, with /root/reference mounted:
  1. linalg restatements == verbatim GPy.util.linalg (jitchol ladder incl. the
     linalg_test.py:18-37 case, pdinv, dpotrs, dpotri, dtrtrs, dtrtri, tdot,
     symmetrify), diag.add == GPy.util.diag.add, Standardize == normalizer,
     get_quantiles/normalize == GPyOpt.util.general.
  2. kernel gradient reductions == the reference's compiled stationary_utils.c
     (the cython_tests.py:39-67 setup: X 300x10, Z 20x10, random dL_dK).
  3. an end-to-end LML through the *reference's* tdot/diag.add/pdinv/dpotrs
     equals exact_gaussian_inference().
  4. the reference tests' invariants: pinv closed form (model_tests.py:63-82),
     var >= 0 stress (model_tests.py:25-61), normaliser equivalence
     (model_tests.py:84-119), finite-difference gradients
     (model_tests.py:664-723 / kernel_tests.py:414-422).
Exit status 0 iff everything holds.  tests/test_oracle_pin.py runs the same
checks under pytest (skipped where the reference tree is absent).
"""
import ctypes
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpu_ref as O  # noqa: E402
from oracle import ref_leaf  # noqa: E402


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def check_linalg(R):
    linalg, diag, normalizer, general = R
    rng = np.random.default_rng(0)
    A = rng.standard_normal((60, 60))
    A = A @ A.T + 60 * np.eye(60)
    Ai, L, Li, logdet = linalg.pdinv(A)
    oAi, oL, oLi, ologdet, jit = O.pdinv(A)
    assert jit == 0.0
    for a, b in ((Ai, oAi), (L, oL), (Li, oLi)):
        assert np.array_equal(a, b), "pdinv mismatch"
    assert logdet == ologdet
    B = rng.standard_normal((60, 3))
    assert np.array_equal(linalg.dpotrs(L, B, lower=1)[0], O.dpotrs(oL, B, lower=1)[0])
    assert np.array_equal(linalg.dtrtrs(L, B)[0], O.dtrtrs(oL, B)[0])
    assert np.array_equal(linalg.dpotri(L)[0], O.dpotri(oL)[0])
    assert np.array_equal(linalg.dtrtri(L), O.dtrtri(oL))
    X = rng.standard_normal((50, 7))
    assert np.array_equal(linalg.tdot(X), O.tdot(X))
    S1 = rng.standard_normal((33, 33)); S2 = S1.copy()
    linalg.symmetrify(S1); O.symmetrify(S2)
    assert np.array_equal(S1, S2)
    S1 = rng.standard_normal((33, 33)); S2 = S1.copy()
    linalg.symmetrify(S1, upper=True); O.symmetrify(S2, upper=True)
    assert np.array_equal(S1, S2)
    D1 = A.copy(); D2 = A.copy()
    diag.add(D1, 0.37); O.diag_add(D2, 0.37)
    assert np.array_equal(D1, D2)
    # jitter ladder, as linalg_test.py:8-37
    rs = np.random.RandomState(0)
    N = 12
    Aj = rs.randn(N, N); Aj = Aj @ Aj.T
    w, v = np.linalg.eigh(Aj)
    w[0] = -np.mean(np.diag(Aj)) * 1e-6 * 10 ** 3.5
    Aj = (v * w) @ v.T
    Lr = linalg.jitchol(Aj, maxtries=5)
    Lo, jit = O.jitchol(Aj, maxtries=5)
    assert np.array_equal(Lr, Lo) and jit > 0
    for fn in (lambda: linalg.jitchol(Aj, maxtries=4), lambda: O.jitchol(Aj, maxtries=4)):
        try:
            fn()
        except np.linalg.LinAlgError:
            pass
        else:
            raise AssertionError("jitchol(maxtries=4) must fail")
    # normaliser
    Y = rng.standard_normal((40, 2)) * 3 + 1
    n1 = normalizer.Standardize(); n1.scale_by(Y)
    n2 = O.Standardize(); n2.scale_by(Y)
    assert np.array_equal(n1.normalize(Y), n2.normalize(Y))
    assert np.array_equal(n1.inverse_mean(Y), n2.inverse_mean(Y))
    assert np.array_equal(n1.inverse_variance(Y), n2.inverse_variance(Y))
    # GPyOpt general
    m = rng.standard_normal((30, 1)); s = np.abs(rng.standard_normal((30, 1))); s[3] = 1e-12
    q1 = general.get_quantiles(0.01, 0.2, m, s.copy())
    q2 = O.get_quantiles(0.01, 0.2, m, s.copy())
    for a, b in zip(q1, q2):
        assert np.array_equal(a, b)
    y = rng.standard_normal((25, 1))
    for t in ("stats", "maxmin"):
        assert np.array_equal(general.normalize(y, t), O.normalize(y, t))
    return "linalg/diag/normalizer/general: bit-identical"


def check_native(lib):
    # cython_tests.py:39-67 shapes
    rng = np.random.default_rng(1)
    k = O.RBF(10, variance=1.3, lengthscale=rng.uniform(0.5, 2, 10), ARD=True)
    X = rng.standard_normal((300, 10)); Z = rng.standard_normal((20, 10))
    dL_dK = rng.standard_normal((300, 20))
    invdist = k._inv_dist(X, Z)
    dL_dr = k.dK_dr(k._scaled_dist(X, Z)) * dL_dK
    tmp = np.ascontiguousarray(invdist * dL_dr)
    g = np.zeros(10)
    lib._lengthscale_grads(300, 20, 10, _dp(tmp), _dp(X), _dp(Z), _dp(g))
    ref_len = -g / k.lengthscale ** 3
    _, mine = k.update_gradients_full(dL_dK, X, Z)
    np.testing.assert_allclose(mine, ref_len, rtol=1e-12)
    grad = np.zeros((300, 10))
    lib._grad_X(300, 10, 20, _dp(X), _dp(Z), _dp(tmp), _dp(grad))
    np.testing.assert_allclose(k.gradients_X(dL_dK, X, Z), grad / k.lengthscale ** 2, rtol=1e-12, atol=1e-14)
    return "stationary_utils.c (_grad_X, _lengthscale_grads): rtol 1e-12"


def check_lml_through_reference(R):
    linalg, diag, _, _ = R
    rng = np.random.default_rng(2)
    X = rng.uniform(0, 1, (64, 3)); Y = rng.standard_normal((64, 1))
    for kern in (O.RBF(3, 1.2, 0.4), O.Matern52(3, 0.8, [0.3, 0.5, 0.9], ARD=True)):
        # reference's own building blocks, in the order of exact_gaussian_inference.py:50-70
        K = kern.K(X)
        Ky = K.copy(); diag.add(Ky, 0.05 + 1e-8)
        Wi, LW, LWi, W_logdet = linalg.pdinv(Ky)
        alpha, _ = linalg.dpotrs(LW, Y, lower=1)
        lml = 0.5 * (-Y.size * np.log(2 * np.pi) - Y.shape[1] * W_logdet - np.sum(alpha * Y))
        dL_dK = 0.5 * (linalg.tdot(alpha) - Y.shape[1] * Wi)
        post = O.exact_gaussian_inference(kern, X, Y, 0.05)
        assert lml == post["lml"]
        assert np.array_equal(dL_dK, post["dL_dK"])
        assert np.array_equal(alpha, post["alpha"])
    return "LML/alpha/dL_dK through reference linalg: bit-identical"


def check_invariants():
    rng = np.random.RandomState(0)
    # model_tests.py:63-82 -- pinv closed form
    N, M = 20, 50
    X = rng.randn(N, 1); Y = np.sin(X) + rng.randn(N, 1) * 0.05; Xn = rng.randn(M, 1)
    k = O.RBF(1, 1.0, 1.0)
    gp = O.OracleGP(X, Y, k, noise_var=0.5)
    Kinv = np.linalg.pinv(k.K(X) + np.eye(N) * 0.5)
    mu_hat = k.K(Xn, X).dot(Kinv).dot(Y)
    cov_hat = k.K(Xn) - k.K(Xn, X).dot(Kinv).dot(k.K(X, Xn))
    mu, cov = gp.predict_noiseless(Xn, full_cov=True)
    np.testing.assert_almost_equal(mu_hat, mu); np.testing.assert_almost_equal(cov_hat, cov)
    mu, var = gp.predict_noiseless(Xn)
    np.testing.assert_almost_equal(np.diag(cov_hat)[:, None], var)
    # model_tests.py:25-61 -- var >= 0 on a Branin grid (ARD RBF, noise 1e-5)
    rs = np.random.RandomState(3)
    x1, x2 = np.meshgrid(np.linspace(-5, 10, 5), np.linspace(0, 15, 5))
    Xb = np.c_[x1.ravel(), x2.ravel()]
    Yb = ((Xb[:, 1] - 5.1 / (4 * np.pi ** 2) * Xb[:, 0] ** 2 + 5 * Xb[:, 0] / np.pi - 6) ** 2
          + 10 * (1 - 1 / (8 * np.pi)) * np.cos(Xb[:, 0]) + 10)[:, None]
    kb = O.RBF(2, 2.0, [5.0, 5.0], ARD=True)
    gpb = O.OracleGP(Xb, Yb, kb, noise_var=1e-5)
    Xt = np.c_[rs.uniform(-5, 10, 20000), rs.uniform(0, 15, 20000)]
    _, v = gpb.predict(Xt)
    assert (v >= 0).all()
    # model_tests.py:84-119 -- normaliser equivalence
    Xr = rng.rand(30, 2); Yr = rng.randn(30, 1) * 4 + 7
    g1 = O.OracleGP(Xr, Yr, O.RBF(2, 1.0, 0.5), 0.1, normalizer=True)
    Ys = (Yr - Yr.mean(0)) / Yr.std(0)
    g2 = O.OracleGP(Xr, Ys, O.RBF(2, 1.0, 0.5), 0.1)
    m1, v1 = g1.predict(Xr[:7]); m2, v2 = g2.predict(Xr[:7])
    np.testing.assert_allclose(m1, m2 * Yr.std(0) + Yr.mean(0), rtol=1e-12)
    np.testing.assert_allclose(v1, v2 * Yr.std(0) ** 2, rtol=1e-12)
    # finite-difference gradients (checkgrad analogue), RBF/Mat52 x iso/ARD
    for cls in (O.RBF, O.Matern52):
        for ard in (False, True):
            D = 2
            X = rng.rand(25, D); Y = rng.randn(25, 1)
            ls = np.array([0.4, 0.7]) if ard else np.array([0.5])
            th = np.r_[1.3, ls, 0.2]

            def lml(th):
                k = cls(D, th[0], th[1:-1], ARD=ard)
                return O.OracleGP(X, Y, k, th[-1]).log_likelihood()
            k = cls(D, th[0], th[1:-1], ARD=ard)
            dv, dl, dn = O.OracleGP(X, Y, k, th[-1]).gradients()
            g = np.r_[dv, dl, dn]
            for i in range(th.size):
                e = np.zeros_like(th); e[i] = 1e-6
                fd = (lml(th + e) - lml(th - e)) / 2e-6
                assert abs(fd - g[i]) <= 1e-5 * max(1, abs(g[i])), (cls.__name__, ard, i, fd, g[i])
            # predictive gradients by finite differences
            gp = O.OracleGP(X, Y, k, th[-1])
            xs = rng.rand(3, D)
            dm, dvx = gp.predictive_gradients(xs)
            for q in range(D):
                e = np.zeros((1, D)); e[0, q] = 1e-6
                mp, vp = gp.predict(xs + e); mm, vm = gp.predict(xs - e)
                np.testing.assert_allclose(dm[:, q, 0], ((mp - mm) / 2e-6)[:, 0], rtol=1e-4, atol=1e-6)
                np.testing.assert_allclose(dvx[:, q], ((vp - vm) / 2e-6)[:, 0], rtol=1e-4, atol=1e-6)
    return "reference test invariants (pinv form, var>=0, normaliser, checkgrad): hold"


class _Space(object):
    """The two methods of Design_space this path calls (space.py:303-318, 436-445), for a box without constraints."""

    def __init__(self, bounds):
        self._b = list(bounds)

    def get_bounds(self):
        return self._b

    def indicator_constraints(self, x):
        return np.ones((np.atleast_2d(x).shape[0], 1))


def _oracle_model(kname="rbf", seed=3, noise=1e-2, N=80, D=2):
    X, Y, Xs = O.synthetic_problem(N, D, 60, seed=seed)
    gp = O.OracleGP(X, Y, O.make_kernel(kname, D, 1.3, O.default_lengthscale(D, False)), noise)
    gm = O.OracleGPModel(gp)
    gm.analytical_gradient_prediction = True   # BOModel flag read by AcquisitionBase.__init__ (base.py:22)
    return gm, Xs


def check_acquisitions(A):
    """The reference's own AcquisitionEI / LCB / MPI / LP classes (verbatim modules) evaluated on an oracle-backed
    model against the oracle's restatements: values, gradients and the negated acquisition_function."""
    worst = 0.0
    for kname in ("rbf", "Mat52"):
        for noise in (1e-2, 1e-6):
            gm, Xs = _oracle_model(kname, noise=noise)
            space = _Space([(0.0, 1.0)] * Xs.shape[1])
            fmin = gm.get_fmin()
            cases = (
                (A["EI"].AcquisitionEI(gm, space, None, None, jitter=0.01), O.acq_EI(gm, Xs, 0.01, fmin),
                 O.acq_EI_withGradients(gm, Xs, 0.01, fmin)),
                (A["LCB"].AcquisitionLCB(gm, space, None, None, exploration_weight=2), O.acq_LCB(gm, Xs, 2.0),
                 O.acq_LCB_withGradients(gm, Xs, 2.0)),
                (A["MPI"].AcquisitionMPI(gm, space, None, None, jitter=0.01), O.acq_MPI(gm, Xs, 0.01, fmin),
                 O.acq_MPI_withGradients(gm, Xs, 0.01, fmin)),
            )
            for acq, f0, (f1, df1) in cases:
                f = acq._compute_acq(Xs)
                fg, dfg = acq._compute_acq_withGradients(Xs)
                assert np.array_equal(f, f0) and np.array_equal(fg, f1) and np.array_equal(dfg, df1), type(acq).__name__
                assert np.array_equal(acq.acquisition_function(Xs), O.acquisition_function(f0))
                a, da = acq.acquisition_function_withGradients(Xs)
                assert np.array_equal(a, O.acquisition_function(f1)) and np.array_equal(da, -df1)
            # local penalisation around EI (transform 'none') and LCB (switches itself to 'softplus', LP.py:31-32)
            Xb = Xs[:3]
            for base, transform in ((cases[0][0], "none"), (cases[1][0], "softplus")):
                lp = A["LP"].AcquisitionLP(gm, space, None, base, transform="none")
                assert lp.transform == transform
                L, Min = 2.5, float(gm.model.Y.min())
                lp.update_batches(Xb, L, Min)
                r0, s0 = O.lp_hammer_precompute(gm, Xb, L, Min)
                assert np.array_equal(lp.r_x0, r0) and np.array_equal(lp.s_x0, s0)
                neg = base.acquisition_function(Xs)
                ref = lp.acquisition_function(Xs)
                mine = O.lp_penalized_acquisition(neg, Xs, Xb, r0, s0, transform)
                worst = max(worst, float(np.max(np.abs(ref - mine) / np.maximum(1.0, np.abs(ref)))))
                # gradient: the reference only broadcasts for one row at a time (LP.py:112-133)
                for r in range(4, Xs.shape[0], 7):   # rows 0..2 are the batch points themselves (|x - x0| = 0)
                    xr = Xs[r:r + 1]
                    negr, negdr = base.acquisition_function_withGradients(xr)   # as the reference evaluates it: one row
                    gref = lp.d_acquisition_function(xr)
                    gmine = O.lp_d_acquisition(negr, negdr, xr, Xb, r0, s0, transform)
                    worst = max(worst, float(np.max(np.abs(gref - gmine) / np.maximum(1.0, np.abs(gref)))))
    assert worst < 1e-13, worst
    return "AcquisitionEI/LCB/MPI verbatim == oracle bit-identical; AcquisitionLP value/gradient within %.1e" % worst


def check_lp_evaluator(A):
    """estimate_L and LocalPenalization.compute_batch of the verbatim evaluator module against the oracle's, same
    numpy seed, oracle-backed model; compute_batch with the acquisition's optimiser replaced by an arg-min over a fixed
    candidate table (the optimiser itself is scipy, out of scope)."""
    ev = A["lp_evaluator"]
    gm, Xs = _oracle_model("Mat52", seed=5)
    bounds = [(0.0, 1.0)] * Xs.shape[1]
    np.random.seed(7)
    try:
        L_ref = ev.estimate_L(gm.model, bounds)
    except (TypeError, IndexError, ValueError) as e:
        # scipy >= 1.5 hands minimize()'s scalar `fun` back as a float; the reference (pinned to scipy 1.2.0,
        # to_install.txt:12) indexes it as res.fun[0][0].  Its algorithm up to that line is what can run here.
        L_ref = None
        note = "reference estimate_L does not run on this scipy (%s: %s)" % (type(e).__name__, e)
    np.random.seed(7)
    L_mine = O.estimate_L(gm.model, bounds)
    if L_ref is not None:
        assert abs(L_ref - L_mine) <= 1e-12 * abs(L_ref), (L_ref, L_mine)
        note = "estimate_L verbatim == oracle (%.12g)" % L_mine
    else:
        # same sampling stream and the same starting point: replay the reference's lines 60-65 by hand
        np.random.seed(7)
        gen = sys.modules["GPyOpt.util.general"]
        samples = np.vstack([gen.samples_multidimensional_uniform(bounds, 500), gm.model.X])
        dm, _ = gm.model.predictive_gradients(samples)
        start = -np.sqrt((dm * dm).sum(1)).min()
        assert L_mine >= -start - 1e-12, (L_mine, start)
    space = _Space(bounds)
    table = Xs

    class TableLP(A["LP"].AcquisitionLP):
        def optimize(self, duplicate_manager=None):
            a = self.acquisition_function(table)
            i = int(np.argmin(a))
            return table[i:i + 1], a[i]
    gm.analytical_gradient_prediction = True
    base = A["EI"].AcquisitionEI(gm, space, None, None, jitter=0.01)
    sys.modules["GPyOpt.acquisitions"].AcquisitionLP = A["LP"].AcquisitionLP
    lp = TableLP(gm, space, None, base)
    saved = ev.estimate_L
    if L_ref is None:
        ev.estimate_L = O.estimate_L   # the verbatim batch loop around the one function that cannot run here
    try:
        np.random.seed(11)
        B_ref = ev.LocalPenalization(lp, 4).compute_batch()
    finally:
        ev.estimate_L = saved
    np.random.seed(11)
    B_mine = O.lp_compute_batch(TableLP(gm, space, None, base), 4)
    assert B_ref.shape == (4, Xs.shape[1]) and np.array_equal(B_ref, B_mine)
    note += "; LocalPenalization.compute_batch verbatim == oracle (4 points)"
    return note


def check_gower_space_and_table_loop(A):
    """The fork's mixed design space and the table loop of run.py:1234-1258 under the Gower kernel:
    (1) the reference's verbatim ``Design_space`` (GPyOpt/GPyOpt/core/task/space.py) against ``O.MixedSpace`` on the four
    methods the path calls; (2) the loop driven by the reference's verbatim AcquisitionLP / AcquisitionEI / AcquisitionLCB
    objects on an oracle-backed Gower model against the same loop driven by ``O.OracleLP`` -- identical rows, L and final
    score vector."""
    import importlib
    space_mod = importlib.import_module("GPyOpt.core.task.space")
    domains = [
        [{'name': 'a', 'type': 'discrete', 'domain': (0, 1, 2, 3)}, {'name': 'x', 'type': 'continuous', 'domain': (-2.0, 5.0)},
         {'name': 'b', 'type': 'discrete', 'domain': (10, 20)}, {'name': 'y', 'type': 'continuous', 'domain': (0.0, 0.5)}],
        [{'name': 'm', 'type': 'discrete', 'domain': tuple(range(5))}, {'name': 'p', 'type': 'discrete', 'domain': tuple(range(7))},
         {'name': 'q', 'type': 'discrete', 'domain': (0, 1, 2)}, {'name': 'r', 'type': 'discrete', 'domain': (0, 1)},
         {'name': 'c', 'type': 'continuous', 'domain': (12.0, 48.0)}, {'name': 'l', 'type': 'continuous', 'domain': (25.4, 100.0)}]]
    for dom in domains:
        ref, mine = space_mod.Design_space(dom), O.MixedSpace(dom)
        assert ref.lengthscales() == mine.lengthscales()
        assert ref.get_continuous_dims() == mine.get_continuous_dims()
        assert ref.get_discrete_dims() == mine.get_discrete_dims()
        assert ref.get_bounds() == mine.get_bounds()
    dom = domains[1]
    ref, mine = space_mod.Design_space(dom), O.MixedSpace(dom)
    rng = np.random.default_rng(8)
    X, table = mine.draw(rng, 70), mine.draw(rng, 400)
    Y = O.normalize(np.sin(X[:, 4:5] / 6.0) + 0.2 * X[:, 0:1] - 0.1 * (X[:, 2:3] == 1) + 0.02 * rng.standard_normal((70, 1)))
    worst = 0.0
    for base, cls_key, kw in (("EI", "EI", dict(jitter=0.01)), ("LCB", "LCB", dict(exploration_weight=2))):
        gp = O.OracleGP(X, Y, O.make_kernel("Mat52", 6, 0.9, [1.5], Gower=True, space=ref), 1e-6)
        gm = O.OracleGPModel(gp)
        gm.analytical_gradient_prediction = True
        cls = getattr(A[cls_key], "Acquisition" + cls_key)
        lp_ref = A["LP"].AcquisitionLP(gm, ref, None, cls(gm, ref, None, None, **kw))
        lp_mine = O.OracleLP(gm, mine, base)
        assert lp_ref.transform == lp_mine.transform
        np.random.seed(21)
        rows_ref, L_ref, Min_ref = O.lp_table_batch(lp_ref, table, 5)
        np.random.seed(21)
        rows_mine, L_mine, Min_mine = O.lp_table_batch(lp_mine, table, 5)
        assert rows_ref == rows_mine and L_ref == L_mine and Min_ref == Min_mine
        a, b = lp_ref.acquisition_function(table), lp_mine.acquisition_function(table)
        worst = max(worst, float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(a)))))
    assert worst < 1e-13, worst
    return "Design_space verbatim == MixedSpace; table loop on verbatim AcquisitionLP == OracleLP (rows, L; scores %.1e)" % worst


def main():
    if not ref_leaf.available():
        print("reference tree absent: cannot pin"); return 2
    R = ref_leaf.load()
    print(check_linalg(R))
    lib = ref_leaf.load_stationary_utils()
    if lib is None:
        print("oracle/_ref/libstationary_utils.so missing: run `make -C oracle`"); return 2
    print(check_native(lib))
    print(check_lml_through_reference(R))
    print(check_invariants())
    A = ref_leaf.load_acquisitions()
    print(check_acquisitions(A))
    print(check_lp_evaluator(A))
    print(check_gower_space_and_table_loop(A))
    print("ORACLE PINNED")
    return 0


if __name__ == "__main__":
    sys.exit(main())
