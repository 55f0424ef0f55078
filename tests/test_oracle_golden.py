"""Pin the CPU oracle (oracle/) to the reference's own outputs (tests/golden/).

Tolerances are the reference's own (reference tests/test_inference.py:40-121):
rtol 1e-7 (f64) / 1e-5..1e-6 (f32), atol 0 unless noted.
"""
import numpy as np
import pytest
from numpy.testing import assert_allclose

from conftest import load_golden, golden_coo


def _rt(dtype, f64=1e-7, f32=1e-5):
    return f32 if np.dtype(dtype) == np.float32 else f64


def test_psi_gammaln_match_scipy_values(oracle):
    g = load_golden("psi_gammaln.npz")
    x = g["x"]
    # absolute accuracy is what E[log x] needs; near the root 1.4616 psi ~ 0
    assert_allclose(oracle.psi(x), g["psi"], rtol=2e-15, atol=1e-15)
    assert_allclose(oracle.cgammaln(x), g["gammaln"], rtol=1e-14, atol=2e-15)
    # the reference's own 8 test points (tests/test_inference.py:24-37), default rtol
    n = int(g["n_test_points"])
    assert_allclose(oracle.psi(x[:n]), g["psi"][:n])
    assert_allclose(oracle.cgammaln(x[:n]), g["gammaln"][:n])


def test_xphi(oracle, ops):
    dt = ops["theta_shape"].dtype
    got = oracle.compute_Xphi_data(ops["x"], ops["row"], ops["col"], ops["theta_shape"],
                                   ops["theta_rate"], ops["beta_shape"], ops["beta_rate"])
    assert got.dtype == dt
    assert_allclose(got, ops["xphi"], rtol=_rt(dt), atol=0)
    assert_allclose(got, ops["xphi_numpy"], rtol=_rt(dt), atol=0)
    second = oracle.compute_Xphi_data_numpy(ops["x"], ops["row"], ops["col"], ops["theta_shape"],
                                            ops["theta_rate"], ops["beta_shape"], ops["beta_rate"])
    assert_allclose(second, ops["xphi_numpy"], rtol=_rt(dt), atol=0)


def test_shape_updates(oracle, ops):
    dt = ops["xphi_in"].dtype
    N, G = (int(v) for v in ops["shape"])
    th = oracle.compute_loading_shape_update(ops["xphi_in"], ops["row"], N, float(ops["a"]))
    be = oracle.compute_loading_shape_update(ops["xphi_in"], ops["col"], G, float(ops["c"]))
    assert_allclose(th, ops["theta_shape_upd"], rtol=_rt(dt, 1e-7, 1e-6), atol=0)
    assert_allclose(be, ops["beta_shape_upd"], rtol=_rt(dt, 1e-7, 1e-6), atol=0)


def test_rate_updates(oracle, ops):
    dt = ops["theta_shape"].dtype
    th = oracle.compute_loading_rate_update(ops["xi_shape"], ops["xi_rate"],
                                            ops["beta_shape"], ops["beta_rate"])
    be = oracle.compute_loading_rate_update(ops["eta_shape"], ops["eta_rate"],
                                            ops["theta_shape"], ops["theta_rate"])
    assert_allclose(th, ops["theta_rate_upd"], rtol=_rt(dt, 1e-7, 1e-6))
    assert_allclose(be, ops["beta_rate_upd"], rtol=_rt(dt, 1e-7, 1e-6))
    eta = oracle.compute_capacity_rate_update(ops["beta_shape"], ops["beta_rate"], float(ops["dp"]))
    xi = oracle.compute_capacity_rate_update(ops["theta_shape"], ops["theta_rate"], float(ops["bp"]))
    assert_allclose(eta, ops["eta_rate_upd"], rtol=_rt(dt, 1e-7, 1e-6), atol=0)
    assert_allclose(xi, ops["xi_rate_upd"], rtol=_rt(dt, 1e-7, 1e-6), atol=0)


def test_pois_llh(oracle, ops):
    dt = ops["theta_shape"].dtype
    got = oracle.compute_pois_llh(ops["x"], ops["row"], ops["col"], ops["theta_shape"],
                                  ops["theta_rate"], ops["beta_shape"], ops["beta_rate"])
    assert_allclose(got, ops["llh"], rtol=_rt(dt, 1e-7, 1e-6), atol=0)
    assert_allclose(got, ops["llh_numpy"], rtol=_rt(dt, 1e-7, 2e-6), atol=0)
    second = oracle.pois_llh_numpy(ops["x"], ops["row"], ops["col"], ops["theta_shape"],
                                   ops["theta_rate"], ops["beta_shape"], ops["beta_rate"])
    assert_allclose(second, ops["llh_numpy"], rtol=_rt(dt, 1e-7, 1e-6), atol=0)
    assert_allclose(np.mean(-got), float(ops["mean_neg_llh"]), rtol=_rt(dt, 1e-9, 1e-6))


FITS = [
    ("fit_data_k5_s0_f64.npz", np.float64, {}),
    ("fit_data_k5_s1_f64.npz", np.float64, {}),
    ("fit_conf_k4_s0_f64.npz", np.float64, {}),
    ("fit_data_k5_s0_f64_simul.npz", np.float64, {"simultaneous": True}),
    ("fit_data_k5_s0_f64_single.npz", np.float64, {}),   # reference numpy path, same answers
    ("fit_data_k5_s0_f32.npz", np.float32, {}),
    ("fit_conf_k4_s0_f32.npz", np.float32, {}),
]


@pytest.mark.parametrize("fname,dtype,kw", FITS)
def test_whole_fit_trace(oracle, fname, dtype, kw):
    """oracle_fit reproduces the reference's fit(): same seed -> same bp/dp, the
    same per-check loss list (same number of checks = same stop decision) and
    the same final variational parameters."""
    g = load_golden(fname)
    X = golden_coo(g)
    np.random.seed(int(g["seed"]))
    res = oracle.oracle_fit(X, int(g["nfactors"]), dtype=dtype, max_iter=int(g["max_iter"]), **kw)
    assert res["bp"] == float(g["bp"]) and res["dp"] == float(g["dp"])
    assert len(res["loss"]) == len(g["loss"])
    f32 = np.dtype(dtype) == np.float32
    assert_allclose(res["loss"], g["loss"], rtol=2e-5 if f32 else 1e-10)
    st = res["state"]
    for name in ("xi", "theta", "eta", "beta"):
        for part in ("shape", "rate"):
            got = getattr(st, "%s_%s" % (name, part))
            assert_allclose(got, g["%s_%s" % (name, part)], rtol=5e-3 if f32 else 1e-8,
                            err_msg="%s %s" % (name, part))


def test_project_trace(oracle):
    """freeze_genes path (scHPF.project, reference scHPF_.py:448-503)."""
    full = load_golden("fit_data_k5_s0_f64.npz")
    g = load_golden("project_data_k5_f64.npz")
    X = golden_coo(g)
    frozen = (full["eta_shape"], full["eta_rate"], full["beta_shape"], full["beta_rate"])
    np.random.seed(int(g["seed"]))
    res = oracle.oracle_fit(X, int(g["nfactors"]), bp=float(full["bp"]), dp=float(full["dp"]),
                            min_iter=2, max_iter=20, check_freq=2, frozen=frozen,
                            self_max_iter=60)
    assert len(res["loss"]) == len(g["loss"])
    assert_allclose(res["loss"], g["loss"], rtol=1e-10)
    st = res["state"]
    assert_allclose(st.theta_shape, g["theta_shape"], rtol=1e-8)
    assert_allclose(st.theta_rate, g["theta_rate"], rtol=1e-8)
    assert_allclose(st.xi_rate, g["xi_rate"], rtol=1e-8)
    assert np.array_equal(st.beta_shape, g["beta_shape"])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_fused_cpu_variant_matches_reference_structure(oracle, dtype):
    """bench.py's second CPU comparator (SURVEY 8(d) variant ii, oracle/cavi_fused_impl.h: exp
    hoisted, no Xphi, parallel CSR + CSC passes) computes the same iteration as the
    reference-structure oracle (variant i) -- so both CPU numbers time the same mathematics."""
    from conftest import synthetic_counts
    X = synthetic_counts(400, 600, 0.08, seed=5)
    K, a, c = 12, 0.3, 0.3
    np.random.seed(3)
    bp, dp, st = oracle.setup_state(X, K, np.dtype(dtype), a, 1.0, c, 1.0)
    st.xi_shape[:] = 1.0 + K * a
    st.eta_shape[:] = 1.0 + K * c
    fused = st.copy()
    M = oracle.FusedMatrix(X, dtype)
    for it in range(3):
        oracle.cavi_iteration(X.data, X.row, X.col, st, a, c, bp, dp)
        oracle.fused_iteration(M, fused, a, c, bp, dp, nthreads=4)
        rtol = 1e-11 if np.dtype(dtype) == np.float64 else 5e-5 * (it + 1)
        for got, want in zip(fused.arrays(), st.arrays()):
            assert_allclose(got, want, rtol=rtol)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("flags", [{}, {"simultaneous": True}, {"freeze_genes": True}])
def test_split_scatter_is_the_serial_scatter_bit_for_bit(oracle, dtype, flags):
    """The parity tests at BASELINE sizes run the oracle's two scatter-adds split by destination row over threads
    (cavi_oracle_impl.h orc_shape_update, scatter_threads > 1): every sum receives the same terms in the same order as
    in the reference's serial loop (hpf_numba.py:152-155), so the whole iteration is the same bits."""
    from conftest import synthetic_counts
    X = synthetic_counts(700, 333, 0.08, seed=5)
    perm = np.random.RandomState(1).permutation(X.nnz)       # COO order is not sorted in general
    x, row, col = X.data[perm], X.row[perm], X.col[perm]
    K, a, c = 7, 0.3, 0.3
    np.random.seed(3)
    bp, dp, st = oracle.setup_state(X, K, np.dtype(dtype), a, 1.0, c, 1.0)
    serial, split = st.copy(), st.copy()
    for _ in range(3):
        oracle.cavi_iteration(x, row, col, serial, a, c, bp, dp, nthreads=2, scatter_threads=1, **flags)
        oracle.cavi_iteration(x, row, col, split, a, c, bp, dp, nthreads=2, scatter_threads=5, **flags)
    for got, want in zip(split.arrays(), serial.arrays()):
        assert np.array_equal(got, want)
    xphi = oracle.compute_Xphi_data(x, row, col, st.theta_shape, st.theta_rate, st.beta_shape, st.beta_rate)
    assert np.array_equal(oracle.compute_loading_shape_update(xphi, col, X.shape[1], c, scatter_threads=4),
                          oracle.compute_loading_shape_update(xphi, col, X.shape[1], c))
