"""GPU: seeded random sweep over shapes, kernels and blocking options against the live oracle.

Every case draws (N, D, M, P, kernel, ARD, noise, panel_tiles, mc_max, pipelined?; every third one with inner_tiles = 2) and compares LML, alpha,
posterior mean / variance / full covariance, hyper-gradients, predictive gradients and the three acquisitions
with the oracle (tolerances as in test_gpu_parity.py: LML 1e-8 rel, posterior 1e-6 rel, gradients 1e-6 of scale).
The blocking options change the launch structure (panel width, candidate chunking, pipelining), never the results
beyond rounding -- this is the test that says so.
"""
import numpy as np
import pytest

from conftest import emulation_modes
from gaussian_process_optimization_amd import _lib
from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu


def relmax(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1.0, np.max(np.abs(b)))


def _cases(n=36):
    rng = np.random.default_rng(20240611)
    out = []
    for i in range(n):
        N = int(rng.choice([1, 3, 17, 64, 127, 128, 129, 255, 256, 257, 300, 511, 640, 777, 1025]))
        D = int(rng.choice([1, 2, 3, 5, 8, 16, 33, 64]))
        M = int(rng.choice([1, 2, 63, 128, 129, 200, 385]))
        P = int(rng.choice([1, 1, 1, 2, 3]))
        out.append((i, N, D, M, P, ["rbf", "Mat52"][int(rng.integers(2))], bool(rng.integers(2)),
                    float(rng.choice([1e-1, 1e-2, 1e-3])), int(rng.integers(1, 9)), int(rng.choice([128, 256, 16384])),
                    bool(rng.integers(2))))
    return out


@pytest.fixture(scope="module", params=emulation_modes())
def h(request):
    """One context per arithmetic mode: every test of this module runs in true fp64 AND with the bulk contractions
    emulated on the int8 matrix cores (same tolerances).  Contexts a test creates itself inherit the mode through the
    environment default (gp_create reads GPHIP_EMULATE_FP64)."""
    import os
    prev = os.environ.get("GPHIP_EMULATE_FP64")
    os.environ["GPHIP_EMULATE_FP64"] = str(request.param)
    hd = _lib.Handle(0)
    hd.set_option("emulate_fp64", request.param)
    yield hd
    hd.close()
    if prev is None:
        os.environ.pop("GPHIP_EMULATE_FP64", None)
    else:
        os.environ["GPHIP_EMULATE_FP64"] = prev


@pytest.mark.parametrize("case", _cases(), ids=lambda c: "c%d-N%d-D%d-M%d-P%d-%s-pt%d-mc%d%s" % (
    c[0], c[1], c[2], c[3], c[4], c[5], c[8], c[9], "-pipe" if c[10] else ""))
def test_random_case(h, case):
    i, N, D, M, P, kname, ard, noise, pt, mc, pipe = case
    rng = np.random.default_rng(1000 + i)
    X = rng.uniform(0, 1, (N, D))
    Y = rng.standard_normal((N, P))
    Xs = rng.uniform(-0.05, 1.05, (M, D))
    ls = rng.uniform(0.4, 1.5, D) * np.sqrt(D) * 0.5 if ard else np.array([0.35 * np.sqrt(D)])
    var = float(rng.uniform(0.5, 2.5))
    kern = O.make_kernel(kname, D, var, ls, ARD=ard)
    gp = O.OracleGP(X, Y, kern, noise)
    p = gp.posterior
    h.set_option("panel_tiles", pt)
    h.set_option("mc_max", mc)
    # every third case factors two tile columns per step of the in-panel factorisation (potrf_pair_kernel + trsm2_kernel)
    h.set_option("inner_tiles", 2 if i % 3 == 0 else 1)
    try:
        h.set_data(X, Y)
        h.set_params(0 if kname == "rbf" else 1, ard, var, ls, noise)
        h.set_candidates(Xs)
        if pipe:
            (lml, logdet, jit), mu, v = h.fit_predict(True)
        else:
            lml, logdet, jit = h.fit()
            mu, v = h.predict(True)
        assert jit == 0.0
        assert abs(lml - p["lml"]) <= 1e-8 * max(1.0, abs(p["lml"]))
        assert relmax(h.alpha(), p["alpha"]) < 1e-6
        m0, v0 = gp.predict(Xs)
        assert relmax(mu, m0) < 1e-6 and np.max(np.abs(v - v0) / np.abs(v0)) < 1e-6
        mu2, v2 = h.predict(False)
        m1, v1 = gp.predict_noiseless(Xs)
        assert relmax(mu2, m1) < 1e-6 and np.max(np.abs(v2 - v1)) < 1e-6 * max(1.0, np.max(np.abs(v1))) + 1e-9
        if M <= min(200, mc):  # gp_predict_full_cov keeps all candidates in one chunk
            mf, cf = h.predict_full_cov(True)
            _, c0 = gp.predict(Xs, full_cov=True)
            assert np.max(np.abs(cf - c0)) < 1e-6 * max(1.0, np.max(np.abs(c0)))
        dv, dl, dn = h.lml_grad(ls.size)
        r = gp.gradients()
        sc = max(1.0, abs(r[0]), np.max(np.abs(r[1])), abs(r[2]))
        assert abs(dv - r[0]) < 1e-6 * sc and np.max(np.abs(dl - r[1])) < 1e-6 * sc and abs(dn - r[2]) < 1e-6 * sc
        if P == 1:
            dm, dvx = h.predict_grad()
            dm0, dv0 = gp.predictive_gradients(Xs)
            assert np.max(np.abs(dm - dm0)) < 1e-6 * max(1.0, np.max(np.abs(dm0)))
            assert np.max(np.abs(dvx - dv0)) < 1e-6 * max(1.0, np.max(np.abs(dv0)))
            model = O.OracleGPModel(gp)
            fmin = h.fmin()
            assert abs(fmin - model.get_fmin()) <= 1e-6 * max(1.0, abs(model.get_fmin()))
            f0 = model.get_fmin()
            for t, par, a0 in [(_lib.GP_ACQ_EI, 0.01, O.acq_EI(model, Xs, 0.01, f0)),
                               (_lib.GP_ACQ_LCB, 2.0, O.acq_LCB(model, Xs, 2.0)),
                               (_lib.GP_ACQ_MPI, 0.01, O.acq_MPI(model, Xs, 0.01, f0))]:
                a = h.acq(t, par, fmin)
                a0 = O.acquisition_function(a0)
                assert np.max(np.abs(a - a0)) < 1e-6 * max(1.0, np.max(np.abs(a0)))
                idx, val = h.acq_argbest(t, par, fmin, -1)
                assert idx == int(np.argmin(a[:, 0])) and val == a[idx, 0]
        # the one-call entry points for a handful of locations (fused over the inverse factor where they apply: P = 1 and the
        # coordinates fit the kernel arguments; the batched calls inside otherwise) against the oracle, one and a few at a time
        for k in sorted({1, min(M, 3), min(M, 5)}):
            xs = Xs[:k]
            if P == 1:
                mr, vr, dmr, dvr = h.predict_rows(xs, True, grad=True)
                assert relmax(mr, m0[:k]) < 1e-6 and np.max(np.abs(vr - v0[:k]) / np.abs(v0[:k])) < 1e-6
                assert np.max(np.abs(dmr - dm0[:k])) < 1e-6 * max(1.0, np.max(np.abs(dm0)))
                assert np.max(np.abs(dvr - dv0[:k])) < 1e-6 * max(1.0, np.max(np.abs(dv0)))
                a0, da0 = O.acq_EI_withGradients(model, xs, 0.01, f0)
                ar, dar = h.acq_rows(xs, _lib.GP_ACQ_EI, 0.01, fmin, grad=True)
                assert np.max(np.abs(ar + a0)) < 1e-6 * max(1.0, np.max(np.abs(a0)))
                assert np.max(np.abs(dar + da0)) < 1e-5 * max(1.0, np.max(np.abs(da0)))
            else:
                mr, vr = h.predict_rows(xs, True)
                assert relmax(mr, m0[:k]) < 1e-6 and np.max(np.abs(vr - v0[:k]) / np.abs(v0[:k])) < 1e-6
    finally:
        h.set_option("panel_tiles", 6)
        h.set_option("mc_max", 16384)
        h.set_option("inner_tiles", 1)
