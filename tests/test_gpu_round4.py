"""GPU: round-4 additions -- the acquisition layer's values AND gradients through libgphip (gp_acq_grad, gp_acq_lp_grad), the
launch-error checks, the single-process device group, the flat-ridge optimiser case."""
import numpy as np
import pytest

import gaussian_process_optimization_amd as gpo
from conftest import Case, golden_tags, relmax
from gaussian_process_optimization_amd import _lib
from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu


def _model_of(c):
    """The host mirror's GPModel at a golden case's data and hyper-parameters (no optimisation)."""
    D = c.X.shape[1]
    ard = bool(int(c.ard))
    ls = c.lengthscale if ard else float(c.lengthscale[0])
    k = (gpo.kern.RBF if int(c.kernel) == 0 else gpo.kern.Matern52)(D, float(c.variance), ls, ARD=ard)
    gm = gpo.GPModel(kernel=k, noise_var=float(c.noise), max_iters=0, verbose=False)
    gm.updateModel(c.X, c.Y, None, None)
    return gm


def _no_host_formulas(acq):
    """After this, any use of the host scoring path of ``acq`` is a test failure."""
    def boom(*a, **k):
        raise AssertionError("host formula path taken although the device can score this acquisition")
    acq._compute_acq = boom
    acq._compute_acq_withGradients = boom
    return acq


def _plain_tags(golden):
    have = set(golden.files)
    return [t for t in golden_tags(golden) if t + "/neg_dEI" in have and float(golden[t + "/noise"]) >= 1e-4
            and golden[t + "/Y"].shape[1] == 1]


def test_acquisition_classes_values_and_gradients_on_the_device_match_golden(golden):
    """AcquisitionEI / LCB / MPI .acquisition_function[_withGradients] (base.py:33-50 over EI.py:32-51, LCB.py:31-46,
    MPI.py:32-51) with the HIP GPModel go to gp_acq / gp_acq_grad -- never to the host formulas -- and equal the golden
    neg_EI / neg_LCB / neg_MPI and neg_dEI / neg_dLCB / neg_dMPI (the reference's own modules on the reference's LAPACK
    posterior).  fmin comes from the model's own device get_fmin here, so the tolerance is the posterior's (1e-6 values,
    1e-5 for d/dx of the tail probabilities), not the formula's."""
    tags = _plain_tags(golden)
    assert len(tags) >= 12
    for tag in tags[::3]:
        c = Case(golden, tag)
        gm = _model_of(c)
        assert abs(gm.get_fmin() - float(c.fmin)) <= 1e-6 * max(1.0, abs(float(c.fmin)))
        for cls, kw, name in ((gpo.AcquisitionEI, dict(jitter=0.01), "EI"),
                              (gpo.AcquisitionLCB, dict(exploration_weight=2.0), "LCB"),
                              (gpo.AcquisitionMPI, dict(jitter=0.01), "MPI")):
            acq = _no_host_formulas(cls(gm, **kw))
            assert acq._device_ok()
            ref, dref = getattr(c, "neg_" + name), getattr(c, "neg_d" + name)
            a = acq.acquisition_function(c.Xs)
            f, df = acq.acquisition_function_withGradients(c.Xs)
            assert a.shape == ref.shape and f.shape == ref.shape and df.shape == dref.shape
            atol = 1e-6 * max(np.max(np.abs(ref)), 1e-300)
            assert np.max(np.abs(a - ref)) <= atol and np.max(np.abs(f - ref)) <= atol, (tag, name)
            assert np.max(np.abs(df - dref)) <= 1e-5 * max(np.max(np.abs(dref)), 1e-300), (tag, name)
            # single-row calls (what L-BFGS makes, optimizer.py:28-61) give the rows of the batched call
            f1, df1 = acq.acquisition_function_withGradients(c.Xs[7])
            assert f1.shape == (1, 1) and df1.shape == (1, c.Xs.shape[1])
            # (one row takes the matrix-vector solve of smallm.hip, the batch the tile path: the same sums in another order)
            np.testing.assert_allclose(f1[0], f[7], rtol=1e-9, atol=1e-300)
            np.testing.assert_allclose(df1[0], df[7], rtol=1e-8, atol=1e-10 * np.max(np.abs(dref)))
        gm.model.close()


@pytest.mark.parametrize("base", ["EI", "LCB", "MPI"])
def test_local_penalization_gradient_on_the_device_matches_the_oracle(base):
    """AcquisitionLP.acquisition_function_withGradients (LP.py:112-140) through gp_acq_lp_grad against the oracle's
    restatement (bit-identical to the verbatim LP.py, tests/test_oracle_pin.py) fed with the oracle's own posterior: value,
    gradient with and without a batch, a candidate sitting ON a batch centre (|x - x_k| = 0: the reference divides by it),
    and a centre so far inside its ball that Phi(z) < 1e-50 (term dropped, LP.py:101)."""
    X, Y, table = O.synthetic_problem(160, 3, 600, seed=41)
    gm = gpo.GPModel(kernel=gpo.kern.Matern52(3, 1.3, 0.55), noise_var=0.02, max_iters=0, verbose=False)
    gm.updateModel(X, Y, None, None)
    gm0 = O.OracleGPModel(O.OracleGP(X, Y, O.make_kernel("Mat52", 3, 1.3, [0.55]), 0.02))
    cls, par = {"EI": (gpo.AcquisitionEI, 0.01), "LCB": (gpo.AcquisitionLCB, 2.0), "MPI": (gpo.AcquisitionMPI, 0.01)}[base]
    fng = {"EI": O.acq_EI_withGradients, "LCB": O.acq_LCB_withGradients, "MPI": O.acq_MPI_withGradients}[base]
    space = gpo.Design_space([{'name': 'x', 'type': 'continuous', 'domain': (0, 1), 'dimensionality': 3}])
    inner = _no_host_formulas(cls(gm, space))
    lp = gpo.AcquisitionLP(gm, space, None, inner)
    lp._score_on_host = lp._penalty_slope = None      # the host formulas of the LP layer must not be reached
    tr = lp.transform
    xq = table[:200]
    a0, da0 = fng(gm0, xq, par)

    def check(Xb, r0, s0, rtol):
        f, df = lp.acquisition_function_withGradients(xq)
        assert f.shape == (200,) and df.shape == (200, 3)
        with np.errstate(divide="ignore", invalid="ignore"):
            ref_f = O.lp_penalized_acquisition(-a0, xq, Xb, r0, s0, tr)
            ref_d = O.lp_d_acquisition(-a0, -da0, xq, Xb, r0, s0, tr)
        fin = np.isfinite(ref_d).all(1)
        assert np.array_equal(np.isfinite(df).all(1), fin)                     # the same rows blow up as in the reference
        np.testing.assert_allclose(f[fin], ref_f[fin], rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(df[fin], ref_d[fin], rtol=rtol, atol=rtol * np.max(np.abs(ref_d[fin])))
        np.testing.assert_array_equal(lp.d_acquisition_function(xq[:3]), lp.acquisition_function_withGradients(xq[:3])[1])
    check(None, None, None, 1e-5)
    Xb = np.vstack([table[300], xq[11], table[450]])                            # centre 1 coincides with candidate 11
    lp.update_batches(Xb, 3.1, float(Y.min()))
    r0, s0 = O.lp_hammer_precompute(gm0, Xb, 3.1, float(Y.min()))
    np.testing.assert_allclose(lp.r_x0, r0, rtol=1e-6, atol=1e-9)
    check(Xb, r0, s0, 1e-5)
    # hand-set ball parameters: a huge radius with a tiny width puts every candidate at Phi(z) = 0 for centre 0
    lp.r_x0, lp.s_x0 = np.array([40.0, 0.2, 0.1]), np.array([0.05, 0.3, 0.2])
    f, df = lp.acquisition_function_withGradients(xq)
    with np.errstate(divide="ignore", invalid="ignore"):
        ref_f = O.lp_penalized_acquisition(-a0, xq, Xb, lp.r_x0, lp.s_x0, tr)
        ref_d = O.lp_d_acquisition(-a0, -da0, xq, Xb, lp.r_x0, lp.s_x0, tr)
    fin = np.isfinite(ref_d).all(1)
    # z ~ -790: log Phi(z) ~ -3e5 comes from the asymptotic series (finite), its gradient term is dropped
    assert fin.sum() >= 190 and (f[fin] > 1e5).all()
    np.testing.assert_allclose(f[fin], ref_f[fin], rtol=1e-9)
    np.testing.assert_allclose(df[fin], ref_d[fin], rtol=1e-5, atol=1e-5 * np.max(np.abs(ref_d[fin])))
    gm.model.close()


def test_local_penalization_host_route_equals_device_route():
    """The host formulas that remain for foreign models (acquisitions._Rule, _log_transform, _exclusion) against the device
    epilogue on the SAME device posterior: 1e-9."""
    X, Y, table = O.synthetic_problem(140, 2, 300, seed=8)
    gm = gpo.GPModel(kernel=gpo.kern.RBF(2, 0.9, 0.35), noise_var=0.03, max_iters=0, verbose=False)
    gm.updateModel(X, Y, None, None)
    space = gpo.Design_space([{'name': 'x', 'type': 'continuous', 'domain': (0, 1), 'dimensionality': 2}])
    for cls in (gpo.AcquisitionEI, gpo.AcquisitionLCB, gpo.AcquisitionMPI):
        dev = gpo.AcquisitionLP(gm, space, None, cls(gm, space))
        hostbase = cls(gm, space)
        hostbase._device_ok = lambda: False
        host = gpo.AcquisitionLP(gm, space, None, hostbase)
        for lp in (dev, host):
            lp.update_batches(table[[3, 77]], 2.2, float(Y.min()))
        assert dev._lp_device_ok() and not host._lp_device_ok()
        fd, dd = dev.acquisition_function_withGradients(table[:64])
        fh, dh = host.acquisition_function_withGradients(table[:64])
        np.testing.assert_allclose(fd, fh, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(dd, dh, rtol=1e-7, atol=1e-9 * np.max(np.abs(dh)))
    gm.model.close()


def test_refused_launch_fails_the_call_instead_of_returning_a_stale_result():
    """Every kernel launch and every stream-ordering call is checked (GP_LAUNCH / GP_NOTE / GP_SYNC): with the diagonal-tile
    kernel asked for more LDS than a CU has, its launch is refused -- gp_fit used to return success with info == 0 over an
    unfactored matrix (jitchol would have raised, linalg.py:56-81); now every entry point that factors reports a HIP error
    naming the kernel, the side streams are drained, and the same context works again once the hook is off."""
    X, Y, Xs = O.synthetic_problem(1700, 3, 300, seed=2)     # 14 tiles: look-ahead scheduler for the one-call entry points
    h = _lib.Handle(0)
    h.set_data(X, Y)
    h.set_params(0, 0, 1.0, [0.4], 1e-2)
    h.set_candidates(Xs)
    good = h.fit()
    mu, var = h.predict(True)
    h.set_option("debug_potrf_lds", 200 * 1024)
    try:
        for call in (h.fit, lambda: h.fit_predict(True), lambda: h.fit_grad(1)):
            with pytest.raises(RuntimeError) as ei:
                call()
            assert "potrf_tile_kernel" in str(ei.value)
        with pytest.raises(RuntimeError):
            h.predict(True)                                  # nothing fitted any more: a state error, not stale numbers
    finally:
        h.set_option("debug_potrf_lds", 0)
    again = h.fit()
    assert again == good
    mu2, var2 = h.predict(True)
    assert np.array_equal(mu, mu2) and np.array_equal(var, var2)
    h.close()


def _tied_table(seed, M, D, acq_single):
    """A candidate table whose best row (by ``acq_single(table) -> (row, value)``) also sits in the OTHER half."""
    table = np.random.default_rng(seed).uniform(0, 1, (M, D))
    i0, _ = acq_single(table)
    twin = (i0 + M // 2) % M                      # lands in the other block of a two-way split
    table[twin] = table[i0]
    return table, min(i0, twin), max(i0, twin)


@pytest.mark.parametrize("devices", [(0, 0), (0,), (0, 0, 0)])
def test_device_group_equals_single_device(devices):
    """gp_group_* (one host thread, one context per device) on a one-GPU box: the same device listed twice /
    three times exercises the split, the per-block scoring of every member and the merge (host merge: no communicator holds a
    device twice); one device alone takes the RCCL route with a one-rank communicator from ncclCommInitAll.  Winner and
    top-5 -- with a tie ACROSS blocks among the winners -- equal the single-context results bit for bit; the replicas'
    LML equals the single context's."""
    X, Y, _ = O.synthetic_problem(700, 3, 8, seed=17)
    single = _lib.Handle(0)
    single.set_data(X, Y)
    single.set_params(1, 0, 1.2, [0.5], 2e-2)
    lml = single.fit()[0]
    fmin = single.fmin()

    def best_single(t):
        single.set_candidates(t)
        return single.acq_argbest(_lib.GP_ACQ_EI, 0.01, fmin, -1)
    table, first, second = _tied_table(5, 1001, 3, best_single)   # 1001 rows: uneven blocks
    grp = _lib.Group(devices)
    info = grp.info()
    assert info["ndev"] == len(devices) and info["rccl"] == (len(set(devices)) == len(devices)), info
    grp.set_data(X, Y)
    grp.set_params(1, 0, 1.2, [0.5], 2e-2)
    assert grp.fit()[0] == lml
    assert grp.fmin() == fmin
    grp.set_candidates(table)
    single.set_candidates(table)
    for typ, par in ((_lib.GP_ACQ_EI, 0.01), (_lib.GP_ACQ_LCB, 2.0), (_lib.GP_ACQ_MPI, 0.01)):
        for sense in (-1, +1):
            assert grp.acq_argbest(typ, par, fmin, sense) == single.acq_argbest(typ, par, fmin, sense)
            gi, gv = grp.acq_topk(typ, par, fmin, sense, 5)
            si, sv = single.acq_topk(typ, par, fmin, sense, 5)
            assert np.array_equal(gi, si) and np.array_equal(gv, sv)
    gi, gv = grp.acq_topk(_lib.GP_ACQ_EI, 0.01, fmin, -1, 5)
    assert gi[0] == first and gi[1] == second and gv[0] == gv[1]          # the cross-block tie, lowest row first
    assert grp.acq_argbest(_lib.GP_ACQ_EI, 0.01, fmin, -1)[0] == first
    # fewer rows than members: the surplus members sit the round out, the tail of a top-k is marked empty
    grp.set_candidates(table[:2])
    single.set_candidates(table[:2])
    assert grp.acq_argbest(_lib.GP_ACQ_EI, 0.01, fmin, -1) == single.acq_argbest(_lib.GP_ACQ_EI, 0.01, fmin, -1)
    gi, gv = grp.acq_topk(_lib.GP_ACQ_EI, 0.01, fmin, -1, 4)
    si, sv = single.acq_topk(_lib.GP_ACQ_EI, 0.01, fmin, -1, 4)
    assert np.array_equal(gi, si) and np.array_equal(gv, sv) and list(gi[2:]) == [-1, -1]
    with pytest.raises(ValueError):
        grp.acq_topk(_lib.GP_ACQ_EI, 0.01, fmin, -1, 65)
    grp.close()
    single.close()


def test_acquisition_argbest_over_devices_from_one_process():
    """``Acquisition*.argbest / topk(table, devices=[...])``: what a single-process caller of run.py:1240-1241 /
    anchor_points_generator.py:59-61 uses to reach several GPUs; follows the model through new data and new hyper-parameters."""
    X, Y, table = O.synthetic_problem(400, 2, 3000, seed=9, standardize=False)
    Y = 5 * Y - 2
    gm = gpo.GPModel(kernel=gpo.kern.RBF(2, 1.0, 0.3), noise_var=0.02, max_iters=0, verbose=False)
    gm.model = gpo.models.GPRegression(X, Y, gpo.kern.RBF(2, 1.0, 0.3), normalizer=True, noise_var=0.02)
    for cls in (gpo.AcquisitionEI, gpo.AcquisitionLCB):
        acq = cls(gm)
        assert acq.argbest(table, -1, devices=[0, 0]) == acq.argbest(table, -1)
        i1, v1 = acq.topk(table, 5, -1, devices=[0, 0])
        i0, v0 = acq.topk(table, 5, -1)
        assert np.array_equal(i1, i0) and np.array_equal(v1, v0)
    acq = gpo.AcquisitionEI(gm)
    gm.model.kern.lengthscale[:] = 0.2                 # new hyper-parameters: the group refits its replicas
    assert acq.argbest(table, -1, devices=[0, 0]) == acq.argbest(table, -1)
    gm.model.set_XY(X[:300], Y[:300])                  # new data
    assert acq.argbest(table, +1, devices=[0, 0]) == acq.argbest(table, +1)
    gm.model.close()


@pytest.mark.parametrize("N", [2048, 5000])
def test_pair_step_factorisation_matches_the_tile_step(N):
    """Option inner_tiles = 2 (potrf_pair_kernel: two diagonal tiles per launch with the tile between them solved and the second
    one updated inside; trsm2_kernel: both tile columns of the rows below in one launch; one K = 256 update) against the
    128-column step on the single-stream (N = 2048) and the look-ahead (N = 5000, 40 tiles: ragged last panel) schedulers:
    BITWISE the same factor, LML and posterior -- the pair kernel, the strip kernel and the K = 256 update contract k in the order
    the 128-column step's launches do (the same four-k groups per matrix instruction, the same sequence of groups), and an fp64
    accumulator that goes through memory between two launches loses nothing -- and reproducible from run to run."""
    X, Y, Xs = O.synthetic_problem(N, 4, 300, seed=3)
    h = _lib.Handle(0)
    h.set_option("emulate_fp64", 0)
    h.set_option("lookahead_min_tiles", 20)
    h.set_data(X, Y)
    h.set_params(0, 0, 1.3, [0.45], 1e-2)
    h.set_candidates(Xs)
    lml1 = h.fit()[0]
    L1 = h.chol()
    mu1, v1 = h.predict(True)
    h.set_option("inner_tiles", 2)
    lml2 = h.fit()[0]
    L2 = h.chol()
    mu2, v2 = h.predict(True)
    assert lml2 == lml1 and np.array_equal(L2, L1)
    assert np.array_equal(mu2, mu1) and np.array_equal(v2, v1)
    assert h.fit()[0] == lml2 and np.array_equal(h.chol(), L2)
    (lml3, _, _), mu3, v3 = h.fit_predict(True)                      # the one-call entry point takes the same steps
    assert lml3 == lml2 and np.array_equal(mu3, mu2) and np.array_equal(v3, v2)
    h.close()


@pytest.mark.parametrize("N,keep", [(3000, 4), (6100, 10), (12416, 36)])
def test_chain_owned_columns_leave_the_factor_bitwise_unchanged(N, keep):
    """Look-ahead factorisation, options own_keep_per_row / own_keep_base (factor_lookahead): the last tile columns of the
    trailing matrix take their panel updates on the chain stream instead of the bulk stream.  Every tile still sees the same
    panels in the same order with the same K = 768 contraction, so factor, LML and posterior are BITWISE those of the plain
    schedule (own_keep_per_row = 0) -- for gp_fit, gp_fit_predict and gp_fit_grad, with a ragged last panel, and again on a
    refit.  N = 12416 is the default rule at a size where it owns columns (it owns none below ~90 tiles); the smaller cases
    force a wide owned range so that hand-back of columns to the bulk stream happens at every panel."""
    X, Y, Xs = O.synthetic_problem(N, 5, 700, seed=N)
    h = _lib.Handle(0)
    h.set_option("emulate_fp64", 0)
    h.set_option("lookahead_min_tiles", 0)
    h.set_data(X, Y)
    h.set_params(0, 0, 1.2, [0.5], 1e-2)
    h.set_candidates(Xs)
    h.set_option("own_keep_per_row", 0)
    lml0 = h.fit()[0]
    L0 = h.chol()
    (_, _, _), mu0, v0 = h.fit_predict(True)
    g0 = h.fit_grad(1)
    h.set_option("own_keep_per_row", keep)
    h.set_option("own_keep_base", 0 if keep < 36 else 200)
    for _ in range(2):
        assert h.fit()[0] == lml0 and np.array_equal(h.chol(), L0)
    (lml1, _, _), mu1, v1 = h.fit_predict(True)
    assert lml1 == lml0 and np.array_equal(mu1, mu0) and np.array_equal(v1, v0)
    g1 = h.fit_grad(1)
    assert g1[0] == g0[0] and g1[1][0] == g0[1][0] and np.array_equal(g1[1][1], g0[1][1]) and g1[1][2] == g0[1][2]
    # candidate stages released from the very first panel (pipe_start_pct = 0): the first owned range still follows the rule
    # (it used to become the WHOLE trailing matrix -- every update on the chain stream, the look-ahead overlap lost), and the
    # results stay the same bits; the timing check is coarse on purpose (a lost overlap costs tens of per cent at this size)
    import time
    h.set_option("pipe_start_pct", 0)
    h.set_option("pipe_start_pct_grad", 0)
    (lml2, _, _), mu2, v2 = h.fit_predict(True)
    assert lml2 == lml0 and np.array_equal(mu2, mu0) and np.array_equal(v2, v0)
    g2 = h.fit_grad(1)
    assert g2[0] == g0[0] and np.array_equal(g2[1][1], g0[1][1])
    if N >= 12000:
        def wall(fn):
            fn(); h.synchronize()
            t0 = time.perf_counter(); fn(); h.synchronize()
            return time.perf_counter() - t0
        t_zero = wall(lambda: h.fit_predict(True))
        h.set_option("pipe_start_pct", -1)
        t_default = wall(lambda: h.fit_predict(True))
        assert t_zero < 1.35 * t_default, (t_zero, t_default)
    h.close()


@pytest.mark.parametrize("N,D", [(700, 3), (5000, 6)])
def test_small_m_path_equals_the_tile_path(N, D):
    """Up to "small_m" (8) candidates take the matrix-vector solve of smallm.hip (forward substitution over the panels, row dots
    with Ky^-1 for the gradients) instead of the 128-row tile path: the same posterior, predictive gradients and acquisition
    gradients to rounding for M = 1 ... 8, rows of a larger batch reproduced by one-row calls, and gp_fit_predict == gp_fit +
    gp_predict bit for bit at these sizes (both take the same route).  The random-shape sweep checks the same path against the
    oracle (its M = 1, 2 cases)."""
    X, Y, Xs = O.synthetic_problem(N, D, 64, seed=N)
    h = _lib.Handle(0)
    h.set_option("emulate_fp64", 0)
    h.set_option("lookahead_min_tiles", 20)
    h.set_data(X, Y)
    h.set_params(1, 1, 1.4, 0.3 + 0.1 * np.arange(D), 2e-2)
    h.fit()
    fmin = h.fmin()
    h.set_candidates(Xs)
    mu_all, v_all = h.predict(True)
    a_all, da_all = h.acq_grad(_lib.GP_ACQ_EI, 0.01, fmin)
    for M in (1, 2, 3, 4, 5, 8):
        h.set_candidates(Xs[:M])
        res = {}
        for small in (8, 0):
            h.set_option("small_m", small)
            mu, v = h.predict(True)
            _, v0 = h.predict(False)
            dm, dv = h.predict_grad()
            a, da = h.acq_grad(_lib.GP_ACQ_EI, 0.01, fmin)
            res[small] = (mu, v, v0, dm, dv, a, da)
        h.set_option("small_m", 8)
        for x, y in zip(res[8], res[0]):
            assert np.max(np.abs(x - y)) <= 1e-10 * max(np.max(np.abs(y)), 1e-300), M
        assert relmax(res[8][0], mu_all[:M]) < 1e-10 and relmax(res[8][1], v_all[:M]) < 1e-10
        assert relmax(res[8][6], da_all[:M]) < 1e-9
        (lml, _, _), mu_f, v_f = h.fit_predict(True)
        h.fit()
        mu_s, v_s = h.predict(True)
        assert np.array_equal(mu_f, mu_s) and np.array_equal(v_f, v_s)
    # the phases of a one-row call name the route
    h.set_candidates(Xs[:1])
    h.predict(True)
    assert "cand_solve_rows" in [p["name"] for p in h.phases()]
    h.close()


def test_local_penalization_batch_over_a_device_group():
    """run.py:1238-1257 (LP-penalised scores of the candidate table, arg-max, rows already taken masked) through a device
    group: `LocalPenalization.compute_batch_from_table(table, devices=[0, 0])` picks the rows the single-context loop picks --
    the excluded rows are global rows of the table and reach the member whose block holds them."""
    np.random.seed(4)
    X, Y, table = O.synthetic_problem(150, 2, 3001, seed=14)
    gm = gpo.GPModel(kernel=gpo.kern.Matern52(2, 1.0, 0.3), noise_var=0.01, max_iters=0, verbose=False)
    gm.updateModel(X, Y, None, None)
    space = gpo.Design_space([{'name': 'x', 'type': 'continuous', 'domain': (0, 1), 'dimensionality': 2}])
    lp = gpo.AcquisitionLP(gm, space, None, gpo.AcquisitionEI(gm, space))
    np.random.seed(9)
    single = gpo.LocalPenalization(lp, 5).compute_batch_from_table(table, sense=+1)
    np.random.seed(9)                                  # estimate_L draws its sample points from numpy's global generator
    grouped = gpo.LocalPenalization(lp, 5).compute_batch_from_table(table, sense=+1, devices=[0, 0])
    assert grouped == single and len(set(single)) == 5
    # ... and they are the rows the oracle's loop picks (run.py:1234-1258 restated, O.lp_table_batch)
    gp0 = O.OracleGP(X, Y, O.make_kernel("Mat52", 2, 1.0, [0.3]), 0.01)
    lp0 = O.OracleLP(O.OracleGPModel(gp0), space, "EI")
    np.random.seed(9)
    want, L0, _ = O.lp_table_batch(lp0, table, 5)
    assert gpo.LocalPenalization(lp, 5).compute_batch_from_table(table, sense=+1, devices=[0, 0], lipschitz=L0) == want
    # a penalised arg-best with exclusions straddling both blocks, and every row of one block taken
    lp.update_batches(table[[10, 2500]], 3.0, float(Y.min()))
    taken = [int(i) for i in (0, 1499, 1500, 1501, 3000)]
    assert lp.argbest(table, +1, exclude=taken, devices=[0, 0]) == lp.argbest(table, +1, exclude=taken)
    small = table[:4]
    assert lp.argbest(small, +1, exclude=[0, 1], devices=[0, 0]) == lp.argbest(small, +1, exclude=[0, 1])
    gm.model.close()
