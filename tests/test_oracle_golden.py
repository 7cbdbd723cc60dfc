"""CPU: the oracle restatement reproduces the committed golden vectors (made through the reference's modules)."""
import numpy as np
import pytest

from conftest import Case, golden_tags, relmax
from oracle import cpu_ref as O


def _tags():
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "gp_golden.npz"))
    return golden_tags(g)


@pytest.mark.parametrize("tag", [t for t in _tags() if t.startswith("N64_") or t.startswith("N300_")])
def test_oracle_matches_golden(golden, tag):
    c = Case(golden, tag)
    kern = O.make_kernel("rbf" if int(c.kernel) == 0 else "Mat52", c.X.shape[1], float(c.variance), c.lengthscale,
                         ARD=bool(c.ard))
    gp = O.OracleGP(c.X, c.Y, kern, float(c.noise))
    p = gp.posterior
    rows = c.rows
    assert np.array_equal(p["K"][rows], c.K_rows)
    assert np.array_equal(np.tril(p["L"])[rows], c.L_rows)
    assert p["lml"] == float(c.lml)
    assert np.array_equal(p["alpha"], c.alpha)
    dv, dl, dn = gp.gradients()
    np.testing.assert_allclose(dv, c.dvariance, rtol=1e-13)
    np.testing.assert_allclose(dl, c.dlengthscale, rtol=1e-13)
    np.testing.assert_allclose(dn, c.dnoise, rtol=1e-13)
    mu, var = gp.predict(c.Xs)
    assert np.array_equal(mu, c.mu) and np.array_equal(var, c.var)
    gm = O.OracleGPModel(gp)
    fmin = gm.get_fmin()
    assert fmin == float(c.fmin)
    ei, dei = O.acq_EI_withGradients(gm, c.Xs, 0.01, fmin)
    np.testing.assert_allclose(-ei, c.neg_EI, rtol=1e-13, atol=0)
    np.testing.assert_allclose(-dei, c.neg_dEI, rtol=1e-12, atol=1e-300)
    assert int(np.argmin(-ei)) == int(c.argmin_EI)


def test_jitter_ladder_golden(golden):
    kd = O.RBF(2, 1.0, 0.5)
    for name, tries in (("jit1", 1), ("jit3", 3)):
        p = O.exact_gaussian_inference(kd, golden[name + "/X"], golden[name + "/Y"], float(golden[name + "/noise"]))
        assert p["jitter"] == float(golden[name + "/jitter"])
        diag0 = 1.0 + float(golden[name + "/noise"]) + 1e-8
        np.testing.assert_allclose(p["jitter"], diag0 * 1e-6 * 10 ** (tries - 1), rtol=1e-12)
    with pytest.raises(np.linalg.LinAlgError):
        O.exact_gaussian_inference(kd, golden["jitfail/X"], golden["jitfail/Y"], float(golden["jitfail/noise"]))


def test_multi_output_normalizer_golden(golden):
    kern = O.RBF(3, 0.9, [0.3, 0.5, 0.7], ARD=True)
    gp = O.OracleGP(golden["multi/X"], golden["multi/Y"], kern, 0.02, normalizer=True)
    mu, var = gp.predict(golden["multi/Xs"])
    assert np.array_equal(mu, golden["multi/mu"]) and np.array_equal(var, golden["multi/var"])
    assert gp.log_likelihood() == float(golden["multi/lml"])


def test_posterior_covariance_between_points_reference_golden():
    """The reference's one numeric golden on this path (GPy/GPy/testing/model_tests.py:1158-1174): a Poly(order 1)
    kernel, two training points, expected [[0.4, 2.2], [1, 1]] / 3.  It pins the algebra of
    Posterior.covariance_between_points (posterior.py:109-128) as restated in the oracle; the kernel is the test's own
    three-line stand-in for GPy.kern.Poly defaults (variance = scale = bias = 1)."""
    from oracle import cpu_ref as O

    class Poly(object):
        def K(self, X, X2=None):
            X2 = X if X2 is None else X2
            return X.dot(X2.T) + 1.0

    X1 = np.array([[-2., 2.], [-1., 1.]])
    X2 = np.array([[2., 3.], [-1., 3.]])
    Y = np.array([[1.], [2.]])
    gp = O.OracleGP(X1, Y, Poly(), noise_var=1.0)   # GPRegression default noise_var = 1 (gp_regression.py:29)
    result = gp.posterior_covariance_between_points(X1, X2)
    assert np.allclose(result, np.array([[0.4, 2.2], [1.0, 1.0]]) / 3.0)


def _gower():
    import json
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "gp_gower.npz"))
    dom = json.loads(str(g["domain_json"]))
    return g, dom, sorted({k.split("/")[0] for k in g.files if k.startswith("G_")})


@pytest.mark.parametrize("tag", [t for t in _gower()[2] if "_N64_" in t or "_N150_" in t or "_N100_" in t])
def test_oracle_matches_gower_golden(tag):
    """The Gower fixture (made through the reference's verbatim Design_space and leaf modules) from the oracle alone:
    posterior, the fork's predictive gradients, estimate_L under the fixture's seed, the run.py:1234-1258 rows."""
    g, dom, _ = _gower()
    c = Case(g, tag)
    space = O.MixedSpace(dom)
    kern = O.make_kernel("rbf" if int(c.kernel) == 0 else "Mat52", 6, float(c.variance), c.lengthscale,
                         ARD=bool(int(c.ard)), Gower=True, space=space)
    gp = O.OracleGP(c.X, c.Y, kern, float(c.noise))
    assert np.array_equal(gp.posterior["K"][c.rows], c.K_rows)
    assert gp.posterior["lml"] == float(c.lml)
    mu, var = gp.predict(c.Xs)
    assert np.array_equal(mu, c.mu) and np.array_equal(var, c.var)
    dm, dv = gp.predictive_gradients(c.Xs)
    np.testing.assert_allclose(dm, c.dmdx, rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(dv, c.dvdx, rtol=1e-12, atol=1e-300)
    gm = O.OracleGPModel(gp)
    with np.errstate(divide="ignore", invalid="ignore"):
        for base in ("EI", "LCB"):
            lp = O.OracleLP(gm, space, base)
            np.random.seed(int(c.np_seed))
            rows, L, Min = O.lp_table_batch(lp, c.table, 5)
            assert rows == [int(i) for i in getattr(c, "lp_rows_" + base)]
            np.testing.assert_allclose(L, float(c.L), rtol=1e-12)
            assert Min == float(c.Min)
