"""GPU parity of the stateless operator mirrors (schpf_amd.hpf_hip, through the C ABI)
against the golden vectors produced by the reference and against the CPU oracle.

Tolerances are the reference's own for these functions (reference
tests/test_inference.py:24-121): rtol 1e-7 (f64), 1e-5 / 1e-6 (f32), atol 0.
"""
import numpy as np
import pytest
from numpy.testing import assert_allclose

from conftest import load_golden, synthetic_counts

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from schpf_amd import hpf_hip, _lib
    _lib.require_gpu()
    return hpf_hip


def _rt(dtype, f64=1e-7, f32=1e-5):
    return f32 if np.dtype(dtype) == np.float32 else f64


def test_digamma_gammaln_vs_scipy_values(hip):
    g = load_golden("psi_gammaln.npz")
    x = g["x"]
    assert_allclose(hip.psi(x), g["psi"], rtol=4e-15, atol=4e-15)
    assert_allclose(hip.cgammaln(x), g["gammaln"], rtol=5e-15, atol=5e-15)
    n = int(g["n_test_points"])      # the reference's own points, default rtol 1e-7
    for v, p, l in zip(x[:n], g["psi"][:n], g["gammaln"][:n]):
        assert_allclose(hip.psi(float(v)), p)
        assert_allclose(hip.cgammaln(float(v)), l)
        assert_allclose(hip.psi(np.float32(v)), p, rtol=1e-6)


def test_xphi_golden(hip, ops):
    dt = ops["theta_shape"].dtype
    got = hip.compute_Xphi_data(ops["x"], ops["row"], ops["col"], ops["theta_shape"], ops["theta_rate"],
                                ops["beta_shape"], ops["beta_rate"])
    assert got.dtype == dt and got.shape == ops["xphi"].shape
    assert_allclose(got, ops["xphi"], rtol=_rt(dt), atol=0)
    assert_allclose(got, ops["xphi_numpy"], rtol=_rt(dt), atol=0)
    # rows of X*phi sum to x
    assert_allclose(got.sum(1), ops["x"], rtol=1e-6 if dt == np.float32 else 1e-13)


def test_xphi_numpy_entry_point_golden(hip, ops):
    """compute_Xphi_data_numpy(X, theta, beta, theta_ix=None), the sixth callable of the seam
    (reference hpf_numba.py:117-125): HPF_Gamma arguments, optional row selection."""
    from scipy.sparse import coo_matrix
    from schpf import HPF_Gamma
    import schpf.hpf_numba as seam
    assert seam.compute_Xphi_data_numpy is hip.compute_Xphi_data_numpy
    dt = ops["theta_shape"].dtype
    X = coo_matrix((ops["x"], (ops["row"], ops["col"])), shape=tuple(int(v) for v in ops["shape"]))
    theta = HPF_Gamma(ops["theta_shape"], ops["theta_rate"])
    beta = HPF_Gamma(ops["beta_shape"], ops["beta_rate"])
    got = hip.compute_Xphi_data_numpy(X, theta, beta)
    assert_allclose(got, ops["xphi_numpy"], rtol=_rt(dt), atol=0)
    # theta_ix: X's rows index into a selection of theta's rows (the minibatch call, scHPF_.py:658-660)
    perm = np.random.RandomState(0).permutation(X.shape[0])
    inv = np.argsort(perm)                     # theta[perm][inv] == theta
    big = HPF_Gamma(np.ascontiguousarray(ops["theta_shape"][perm]), np.ascontiguousarray(ops["theta_rate"][perm]))
    got_ix = hip.compute_Xphi_data_numpy(X, big, beta, theta_ix=inv)
    assert_allclose(got_ix, ops["xphi_numpy"], rtol=_rt(dt), atol=0)


def test_shape_updates_golden(hip, ops):
    dt = ops["xphi_in"].dtype
    N, G = (int(v) for v in ops["shape"])
    th = hip.compute_loading_shape_update(ops["xphi_in"], ops["row"], N, float(ops["a"]))
    be = hip.compute_loading_shape_update(ops["xphi_in"], ops["col"], G, float(ops["c"]))
    assert_allclose(th, ops["theta_shape_upd"], rtol=_rt(dt, 1e-7, 1e-6), atol=0)
    assert_allclose(be, ops["beta_shape_upd"], rtol=_rt(dt, 1e-7, 1e-6), atol=0)


def test_rate_updates_golden(hip, ops):
    dt = ops["theta_shape"].dtype
    th = hip.compute_loading_rate_update(ops["xi_shape"], ops["xi_rate"], ops["beta_shape"], ops["beta_rate"])
    be = hip.compute_loading_rate_update(ops["eta_shape"], ops["eta_rate"], ops["theta_shape"], ops["theta_rate"])
    assert_allclose(th, ops["theta_rate_upd"], rtol=_rt(dt, 1e-7, 1e-6))
    assert_allclose(be, ops["beta_rate_upd"], rtol=_rt(dt, 1e-7, 1e-6))
    eta = hip.compute_capacity_rate_update(ops["beta_shape"], ops["beta_rate"], float(ops["dp"]))
    xi = hip.compute_capacity_rate_update(ops["theta_shape"], ops["theta_rate"], float(ops["bp"]))
    assert_allclose(eta, ops["eta_rate_upd"], rtol=_rt(dt, 1e-7, 1e-6), atol=0)
    assert_allclose(xi, ops["xi_rate_upd"], rtol=_rt(dt, 1e-7, 1e-6), atol=0)


def test_pois_llh_golden(hip, ops):
    dt = ops["theta_shape"].dtype
    got = hip.compute_pois_llh(ops["x"], ops["row"], ops["col"], ops["theta_shape"], ops["theta_rate"],
                               ops["beta_shape"], ops["beta_rate"])
    assert_allclose(got, ops["llh"], rtol=_rt(dt, 1e-7, 1e-6), atol=0)
    assert_allclose(got, ops["llh_numpy"], rtol=_rt(dt, 1e-7, 2e-6), atol=0)


def test_estimator_llh_helpers_golden(ops):
    """scHPF.pois_llh_pointwise / mean_negative / cellmean (reference scHPF_.py:372-422)."""
    from conftest import golden_coo
    from schpf import scHPF, HPF_Gamma
    dt = ops["theta_shape"].dtype
    X = golden_coo(ops)
    m = scHPF(4, dtype=dt)
    m.theta = HPF_Gamma(ops["theta_shape"], ops["theta_rate"])
    m.beta = HPF_Gamma(ops["beta_shape"], ops["beta_rate"])
    assert_allclose(m.mean_negative_pois_llh(X), float(ops["mean_neg_llh"]), rtol=_rt(dt, 1e-9, 1e-6))
    assert_allclose(m.cellmean_negative_pois_llh(X), ops["cellmean_neg_llh"], rtol=_rt(dt, 1e-7, 1e-5))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("K", [1, 7, 50])
def test_ops_vs_oracle_seeded(hip, oracle, dtype, K):
    """Larger seeded case, unsorted COO, int64 counts; oracle as checker."""
    X = synthetic_counts(2000, 1500, 0.02, seed=11)
    rng = np.random.RandomState(K)
    perm = rng.permutation(X.nnz)
    x, row, col = X.data[perm].astype(np.int64), X.row[perm], X.col[perm]
    N, G = X.shape
    ths = rng.uniform(0.15, 3.0, (N, K)).astype(dtype); thr = rng.uniform(0.1, 2.0, (N, K)).astype(dtype)
    bes = rng.uniform(0.15, 3.0, (G, K)).astype(dtype); ber = rng.uniform(0.1, 2.0, (G, K)).astype(dtype)
    got = hip.compute_Xphi_data(x, row, col, ths, thr, bes, ber)
    want = oracle.compute_Xphi_data(x, row, col, ths, thr, bes, ber)
    assert_allclose(got, want, rtol=_rt(dtype, 1e-12, 1e-5), atol=0)
    assert_allclose(hip.compute_pois_llh(x, row, col, ths, thr, bes, ber),
                    oracle.compute_pois_llh(x, row, col, ths, thr, bes, ber), rtol=_rt(dtype, 1e-12, 2e-6))
    assert_allclose(hip.compute_loading_shape_update(want, col, G, 0.3),
                    oracle.compute_loading_shape_update(want, col, G, 0.3), rtol=_rt(dtype, 1e-12, 1e-5))
    assert_allclose(hip.compute_loading_rate_update(ths[:, 0], thr[:, 0], bes, ber),
                    oracle.compute_loading_rate_update(ths[:, 0].copy(), thr[:, 0].copy(), bes, ber),
                    rtol=_rt(dtype, 1e-12, 1e-5))


def test_ops_edge_cases(hip):
    """empty input, a single nonzero, bad indices and dtypes fail loudly."""
    ths = np.full((3, 2), 0.5); thr = np.ones((3, 2)); bes = np.full((4, 2), 0.7); ber = np.ones((4, 2))
    e = np.zeros(0, np.int32)
    assert hip.compute_Xphi_data(e, e, e, ths, thr, bes, ber).shape == (0, 2)
    assert hip.compute_pois_llh(e, e, e, ths, thr, bes, ber).shape == (0,)
    one = hip.compute_Xphi_data(np.array([5]), np.array([2]), np.array([3]), ths, thr, bes, ber)
    assert_allclose(one, [[2.5, 2.5]], rtol=1e-14)
    empty_rows = hip.compute_loading_shape_update(np.ones((1, 2)), np.array([2]), 4, 0.3)
    assert_allclose(empty_rows, [[0.3, 0.3], [0.3, 0.3], [1.3, 1.3], [0.3, 0.3]])
    with pytest.raises(ValueError):
        hip.compute_Xphi_data(np.array([5]), np.array([3]), np.array([0]), ths, thr, bes, ber)
    with pytest.raises(TypeError):
        hip.compute_Xphi_data(np.array([5]), np.array([0]), np.array([0]), ths.astype(np.float16),
                              thr, bes, ber)
