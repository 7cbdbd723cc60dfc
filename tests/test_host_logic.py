"""CPU: host-side logic that needs no device -- transforms, sharding arithmetic, design space."""
import numpy as np
import pytest

from gaussian_process_optimization_amd.parameterization import Logexp, Logistic, Param, Parameterized
from gaussian_process_optimization_amd.sharded import merge_best, shard_bounds
from gaussian_process_optimization_amd.bayesian_optimization import Design_space, normalize
from gaussian_process_optimization_amd.acquisitions import get_quantiles
from oracle import cpu_ref as O


@pytest.mark.parametrize("tr", [Logexp(), Logistic(1e-9, 1e6), Logistic(-2.0, 3.0)])
def test_transform_roundtrip_and_gradfactor(tr):
    x = np.linspace(-5, 5, 11)
    f = tr.f(x)
    np.testing.assert_allclose(tr.finv(f), x, rtol=1e-9, atol=1e-9)
    h = 1e-6
    fd = (tr.f(x + h) - tr.f(x - h)) / (2 * h)
    np.testing.assert_allclose(tr.gradfactor(f, np.ones_like(f)), fd, rtol=1e-6, atol=1e-12)


def test_logexp_large_values_pass_through():
    tr = Logexp()
    assert tr.f(np.array([50.0]))[0] == pytest.approx(50.0)
    assert tr.finv(np.array([50.0]))[0] == 50.0


def test_parameterized_flat_vector_and_notify():
    class Root(Parameterized):
        def __init__(self):
            super(Root, self).__init__("m")
            self.n = 0

        def _on_change(self):
            self.n += 1
    r = Root()
    k = Parameterized("rbf")
    k.variance = Param("variance", [1.5]); k.lengthscale = Param("lengthscale", [0.3, 0.7])
    k.link_parameters(k.variance, k.lengthscale)
    lik = Parameterized("Gaussian_noise"); lik.variance = Param("variance", [0.1]); lik.link_parameter(lik.variance)
    r.link_parameters(k, lik)
    np.testing.assert_array_equal(r.param_array, [1.5, 0.3, 0.7, 0.1])
    x = r.optimizer_array.copy()
    r.optimizer_array = x
    np.testing.assert_allclose(r.param_array, [1.5, 0.3, 0.7, 0.1], rtol=1e-12)
    assert r.n == 1
    lik.constrain_fixed(1e-6)
    assert r.optimizer_array.size == 3 and float(lik.variance) == 1e-6
    k.lengthscale[1] = 0.9
    assert r.param_array[2] == 0.9 and r.n == 3
    assert list(r.parameter_names_flat()) == ["m.rbf.variance", "m.rbf.lengthscale[[0]]", "m.rbf.lengthscale[[1]]",
                                              "m.Gaussian_noise.variance"]


def test_shard_bounds_cover_and_are_contiguous():
    for M in (0, 1, 7, 8, 9, 1000003):
        for n in (1, 2, 3, 8):
            b = [shard_bounds(M, r, n) for r in range(n)]
            assert b[0][0] == 0 and b[-1][1] == M
            assert all(b[i][1] == b[i + 1][0] for i in range(n - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_merge_best_numpy_tie_rule():
    a = np.array([0.5, 2.0, 2.0, -1.0, 2.0, -1.0])
    # shards of 2: per-shard (val, global idx) as the device would report them
    for sense, fn in ((+1, np.argmax), (-1, np.argmin)):
        vals, idxs = [], []
        for lo in range(0, 6, 2):
            blk = a[lo:lo + 2]
            i = int(fn(blk))
            vals.append(blk[i]); idxs.append(lo + i)
        gi, gv = merge_best(vals, idxs, sense)
        assert gi == int(fn(a)) and gv == a[gi]
    gi, _ = merge_best([1.0, 5.0], [-1, 3], +1)
    assert gi == 3
    with pytest.raises(ValueError):
        merge_best([1.0], [-1], 1)


def test_design_space_rounding_and_sampling():
    sp = Design_space([{'name': 'x', 'type': 'continuous', 'domain': (0, 1), 'dimensionality': 2},
                       {'name': 'k', 'type': 'discrete', 'domain': (1, 2, 4)}])
    assert sp.dimensionality == 3 and sp.get_bounds() == [(0, 1), (0, 1), (1, 4)]
    np.testing.assert_array_equal(sp.round_optimum([1.3, -0.2, 2.9]), [[1.0, 0.0, 2.0]])
    Z = sp.samples_uniform(200, np.random.RandomState(0))
    assert Z.shape == (200, 3) and set(np.unique(Z[:, 2])) <= {1.0, 2.0, 4.0}
    assert (sp.indicator_constraints(Z) == 1).all()


def test_design_space_string_constraints():
    """space.py:303-318 / random_design.py:21-35: feasible where the expression is < 0; samples are drawn by rejection;
    the negated acquisition is zero outside (acquisitions/base.py:33-39)."""
    dom = [{'name': 'x', 'type': 'continuous', 'domain': (0, 1), 'dimensionality': 2}]
    cons = [{'name': 'c0', 'constraint': 'x[:,0] + x[:,1] - 1'}, {'name': 'c1', 'constraint': '0.2 - x[:,0]'}]
    sp = Design_space(dom, cons)
    assert sp.has_constraints()
    x = np.array([[0.5, 0.4], [0.5, 0.6], [0.1, 0.1], [0.3, 0.7]])
    np.testing.assert_array_equal(sp.indicator_constraints(x), [[1.0], [0.0], [0.0], [0.0]])   # strict <
    Z = sp.samples_uniform(300, np.random.RandomState(1))
    assert Z.shape == (300, 2) and (Z.sum(1) < 1).all() and (Z[:, 0] > 0.2).all()
    with pytest.raises(SyntaxError):
        Design_space(dom, [{'name': 'bad', 'constraint': 'x[:,0] +* 1'}])

    from gaussian_process_optimization_amd.acquisitions import AcquisitionBase

    class _Acq(AcquisitionBase):
        def _compute_acq(self, x):
            return np.ones((x.shape[0], 1)) * 2.0

    class _M(object):
        pass
    a = _Acq(_M(), sp)
    a._device_ok = lambda: False
    np.testing.assert_array_equal(a.acquisition_function(x), [[-2.0], [0.0], [0.0], [0.0]])


def test_host_formulas_equal_oracle():
    rng = np.random.default_rng(0)
    y = rng.standard_normal((30, 1))
    for t in ("stats", "maxmin"):
        np.testing.assert_array_equal(normalize(y, t), O.normalize(y, t))
    m = rng.standard_normal((20, 1)); s = np.abs(rng.standard_normal((20, 1))); s[2] = 1e-13
    for a, b in zip(get_quantiles(0.01, 0.3, m, s.copy()), O.get_quantiles(0.01, 0.3, m, s.copy())):
        np.testing.assert_array_equal(a, b)


def test_host_acquisitions_and_lp_evaluator_equal_oracle():
    """The host-side formula path of acquisitions.py (what a foreign BOModel, constraints or a cost function fall back
    to) and the LP evaluator, on an oracle-backed model: identical to the oracle's restatements, which
    tests/test_oracle_pin.py holds bit-identical to the reference's own EI / LCB / MPI / LP modules."""
    import gaussian_process_optimization_amd as gpo
    from gaussian_process_optimization_amd.acquisitions import estimate_L
    X, Y, Xs = O.synthetic_problem(80, 2, 60, seed=5)
    gp = O.OracleGP(X, Y, O.make_kernel("Mat52", 2, 1.3, O.default_lengthscale(2, False)), 1e-2)
    gm = O.OracleGPModel(gp)
    gm.analytical_gradient_prediction = True
    fmin = gm.get_fmin()

    class _Space(object):
        def get_bounds(self):
            return [(0.0, 1.0), (0.0, 1.0)]

        def indicator_constraints(self, x):
            return np.ones((np.atleast_2d(x).shape[0], 1))
    sp = _Space()
    pairs = ((gpo.AcquisitionEI(gm, sp, jitter=0.01), O.acq_EI_withGradients(gm, Xs, 0.01, fmin)),
             (gpo.AcquisitionLCB(gm, sp, exploration_weight=2), O.acq_LCB_withGradients(gm, Xs, 2.0)),
             (gpo.AcquisitionMPI(gm, sp, jitter=0.01), O.acq_MPI_withGradients(gm, Xs, 0.01, fmin)))
    for acq, (f0, df0) in pairs:
        assert not acq._device_ok()          # a foreign model never takes the device path
        np.testing.assert_array_equal(acq.acquisition_function(Xs), -f0)
        a, da = acq.acquisition_function_withGradients(Xs)
        np.testing.assert_array_equal(a, -f0)
        # (the product's chain rule  d acq/d mean * dm/dx + d acq/d std * ds/dx  regroups MPI's gradient: last-bit differences)
        np.testing.assert_allclose(da, -df0, rtol=1e-13, atol=1e-15 * np.max(np.abs(df0)))
        i, v = acq.argbest(Xs, -1)
        assert i == int(np.argmin(-f0)) and v == float((-f0)[i, 0])
        idx, val = acq.topk(Xs, 5, -1)
        np.testing.assert_array_equal(idx, np.argsort(-f0[:, 0], kind="stable")[:5])
    # local penalisation: hammer parameters, penalised value, evaluator constants and the batch loop
    base = pairs[0][0]
    lp = gpo.AcquisitionLP(gm, sp, None, base)
    Xb = Xs[:3]
    lp.update_batches(Xb, 2.5, float(Y.min()))
    r0, s0 = O.lp_hammer_precompute(gm, Xb, 2.5, float(Y.min()))
    np.testing.assert_array_equal(lp.r_x0, r0)
    np.testing.assert_array_equal(lp.s_x0, s0)
    np.testing.assert_allclose(lp.acquisition_function(Xs).ravel(),
                               O.lp_penalized_acquisition(base.acquisition_function(Xs), Xs, Xb, r0, s0, "none"),
                               rtol=1e-14, atol=0)
    np.random.seed(7)
    L1 = estimate_L(gp, sp.get_bounds())
    np.random.seed(7)
    L0 = O.estimate_L(gp, sp.get_bounds())
    assert L1 == pytest.approx(L0, rel=1e-12)

    class TableLP(gpo.AcquisitionLP):
        def optimize(self, duplicate_manager=None):
            a = self.acquisition_function(Xs)
            i = int(np.argmin(a))
            return Xs[i:i + 1], a[i]
    np.random.seed(11)
    B1 = gpo.LocalPenalization(TableLP(gm, sp, None, base), 4).compute_batch()
    np.random.seed(11)
    B0 = O.lp_compute_batch(TableLP(gm, sp, None, base), 4)
    np.testing.assert_array_equal(B1, B0)


def test_bcast_fit_receiver_side_record():
    """gp_comm_bcast_fit's receiver side (api_comm.hip: pack_fit_record / apply_fit_record), driven host-only through
    gp_comm_selftest_fit_record: RCCL with two ranks cannot run on a one-GPU lease, so the state a receiving rank ends
    up in -- the ROOT's jitter / LML / log det, fitted, and every result derived from its previous factor dropped -- is
    pinned here (run.py:1240-1241 scores shards against replicas of one fit)."""
    import ctypes
    from gaussian_process_optimization_amd import _lib
    lib = _lib.load_library()
    root = np.array([3.5e-6, -1234.56789, 4321.125])
    state = np.zeros(3)
    flags = (ctypes.c_int * 6)()
    rc = lib.gp_comm_selftest_fit_record(_lib.dptr(root), _lib.dptr(state), flags)
    assert rc == 0
    assert np.array_equal(state, root)                      # bitwise: the record is moved, not recomputed
    fitted, fmin_valid, wi_valid, invp_valid, lr_valid, predicted = list(flags)
    assert fitted == 1
    assert [fmin_valid, wi_valid, invp_valid, lr_valid, predicted] == [0, 0, 0, 0, 0]
    assert lib.gp_comm_selftest_fit_record(None, _lib.dptr(state), flags) == _lib.GP_ERR_ARG


def test_group_merges_follow_numpy_tie_rules():
    """gp_merge_best / gp_merge_topk (api_group.hip: what gp_group_* applies to the gathered (value, global row) pairs, host only)
    against sharded.merge_best / merge_topk and against NumPy on the unsharded vector: lowest row among equal values, empty slots
    (idx < 0) ignored, a top-k tail that cannot be filled marked -1."""
    import ctypes
    from gaussian_process_optimization_amd import _lib
    from gaussian_process_optimization_amd.sharded import merge_best, merge_topk, shard_bounds
    lib = _lib.load_library()
    rng = np.random.default_rng(11)
    for trial in range(200):
        M = int(rng.integers(1, 60)); n = int(rng.integers(1, 7)); k = int(rng.integers(1, 8)); sense = int(rng.choice([-1, 1]))
        scores = rng.integers(0, 6, M).astype(float)          # many ties
        vals, idxs, kv, ki = [], [], [], []
        for r in range(n):
            lo, hi = shard_bounds(M, r, n)
            blk = scores[lo:hi]
            if hi > lo:
                j = int(np.argmin(blk) if sense < 0 else np.argmax(blk))
                vals.append(blk[j]); idxs.append(lo + j)
                order = np.argsort(blk if sense < 0 else -blk, kind="stable")[:k]
            else:
                vals.append(np.inf if sense < 0 else -np.inf); idxs.append(-1)
                order = np.zeros(0, dtype=int)
            kv += list(blk[order]) + [np.inf if sense < 0 else -np.inf] * (k - order.size)
            ki += list(lo + order) + [-1] * (k - order.size)
        v = np.array(vals); ix = np.array(idxs, dtype=np.int64)
        oi, ov = ctypes.c_int64(), ctypes.c_double()
        assert lib.gp_merge_best(n, _lib.dptr(v), ix.ctypes.data_as(_lib.c_int64_p), sense, ctypes.byref(oi), ctypes.byref(ov)) == 0
        ref = int(np.argmin(scores) if sense < 0 else np.argmax(scores))
        assert (oi.value, ov.value) == (ref, scores[ref]) == merge_best(v, ix, sense)
        kv = np.array(kv); ki = np.array(ki, dtype=np.int64)
        ti = np.empty(k, dtype=np.int64); tv = np.empty(k)
        assert lib.gp_merge_topk(n * k, _lib.dptr(kv), ki.ctypes.data_as(_lib.c_int64_p), sense, k, ti.ctypes.data_as(_lib.c_int64_p), _lib.dptr(tv)) == 0
        full = np.argsort(scores if sense < 0 else -scores, kind="stable")[:k]
        assert list(ti[:full.size]) == list(full) and list(ti[full.size:]) == [-1] * (k - full.size)
        pi, pv = merge_topk(kv, ki, k, sense)
        assert list(pi) == list(full) and list(tv[:full.size]) == list(scores[full])
    assert lib.gp_merge_best(0, None, None, 1, None, None) == _lib.GP_ERR_ARG


def test_kernel_objects_deepcopy_and_pickle_like_the_reference():
    """A kernel holds hyper-parameters only (the scratch context kern.K evaluates on is per device, module level): it
    deep-copies and pickles, as GPy's kernels do, and the copy is independent of the original."""
    import copy
    import pickle
    import gaussian_process_optimization_amd as gpo
    k = gpo.kern.Matern52(3, 1.3, [0.2, 0.3, 0.4], ARD=True)
    for twin in (copy.deepcopy(k), pickle.loads(pickle.dumps(k)), k.copy()):
        assert type(twin) is type(k) and twin.ARD and float(twin.variance) == 1.3
        assert np.array_equal(twin.lengthscale.values, [0.2, 0.3, 0.4])
        twin.variance[:] = 2.0
        assert float(k.variance) == 1.3
    assert "_kh" not in k.__dict__
