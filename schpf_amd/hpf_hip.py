"""Device counterparts of the reference's numba kernels, same names and signatures.

Mirror of /root/reference/schpf/hpf_numba.py: `compute_Xphi_data` (:54-114),
`compute_pois_llh` (:24-51), `compute_loading_shape_update` (:128-156),
`compute_loading_rate_update` (:159-177), `compute_capacity_rate_update`
(:180-188), `psi` / `cgammaln` (:16-22).  Array in, array out; every call runs
HIP kernels through the C ABI (include/schpf_hip.h).  scHPF._fit does not use
these per iteration -- it drives the device-resident engine (engine.py) -- they
exist so that code written against the reference's operator interface, and the
reference's own unit tests, keep working.
"""
import ctypes

import numpy as np

from . import _lib

__all__ = ["psi", "cgammaln", "compute_pois_llh", "compute_Xphi_data", "compute_Xphi_data_numpy",
           "compute_loading_shape_update", "compute_loading_rate_update",
           "compute_capacity_rate_update", "coo_marginals"]


def _code(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return _lib.F64
    if dtype == np.float32:
        return _lib.F32
    raise TypeError("model dtype must be float64 or float32, got %s" % dtype)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _special(fn, x):
    scalar = np.ndim(x) == 0
    xs = np.atleast_1d(np.asarray(x, dtype=np.float64)).ravel()
    out = np.empty_like(xs)
    _lib.check(fn(xs.size, _p(_c(xs, np.float64)), _p(out)))
    return float(out[0]) if scalar else out.reshape(np.shape(x))


def psi(x):
    """digamma, double -> double (hpf_numba.py:16-18)."""
    return _special(_lib.load().schpf_digamma, x)


def cgammaln(x):
    """log-gamma, double -> double (hpf_numba.py:20-22)."""
    return _special(_lib.load().schpf_gammaln, x)


def _coo_call(fn, out_cols, X_data, X_row, X_col, theta_vi_shape, theta_vi_rate,
              beta_vi_shape, beta_vi_rate):
    dt = theta_vi_shape.dtype
    code = _code(dt)
    ncells, nfactors = theta_vi_shape.shape
    ngenes = beta_vi_shape.shape[0]
    if beta_vi_shape.shape[1] != nfactors:
        raise ValueError("theta and beta must have the same number of factors")
    nnz = X_data.shape[0]
    x = _c(X_data, dt)
    row, col = _c(X_row, np.int32), _c(X_col, np.int32)
    ths, thr = _c(theta_vi_shape, dt), _c(theta_vi_rate, dt)
    bes, ber = _c(beta_vi_shape, dt), _c(beta_vi_rate, dt)
    out = np.empty((nnz, nfactors) if out_cols else (nnz,), dtype=dt)
    _lib.check(fn(code, nnz, ncells, ngenes, nfactors, _p(x), _p(row), _p(col),
                  _p(ths), _p(thr), _p(bes), _p(ber), _p(out)))
    return out


def compute_pois_llh(X_data, X_row, X_col, theta_vi_shape, theta_vi_rate,
                     beta_vi_shape, beta_vi_rate):
    """Pointwise Poisson log-likelihood of the nonzeros (hpf_numba.py:24-51)."""
    return _coo_call(_lib.load().schpf_pois_llh_pointwise, False, X_data, X_row, X_col,
                     theta_vi_shape, theta_vi_rate, beta_vi_shape, beta_vi_rate)


def compute_Xphi_data(X_data, X_row, X_col, theta_vi_shape, theta_vi_rate,
                      beta_vi_shape, beta_vi_rate):
    """X * phi, (nnz, K) (hpf_numba.py:54-114)."""
    return _coo_call(_lib.load().schpf_xphi, True, X_data, X_row, X_col,
                     theta_vi_shape, theta_vi_rate, beta_vi_shape, beta_vi_rate)


def compute_Xphi_data_numpy(X, theta, beta, theta_ix=None):
    """X * phi from a COO matrix and two HPF_Gamma-like objects (.vi_shape / .vi_rate), the sixth
    callable of the reference's operator module (hpf_numba.py:117-125; its numpy fallback for
    single_process=True).  `theta_ix` selects the rows of theta that X's rows refer to (the
    minibatch case, scHPF_.py:658-660).  Same device kernel as compute_Xphi_data."""
    ths, thr = theta.vi_shape, theta.vi_rate
    if theta_ix is not None:
        ths, thr = ths[theta_ix], thr[theta_ix]
    return compute_Xphi_data(X.data, X.row, X.col, ths, thr, beta.vi_shape, beta.vi_rate)


def compute_loading_shape_update(Xphi_data, X_keep, nkeep, shape_prior):
    """Gamma shape update for theta or beta (hpf_numba.py:128-156)."""
    dt = Xphi_data.dtype
    nnz, nfactors = Xphi_data.shape
    out = np.empty((int(nkeep), nfactors), dtype=dt)
    _lib.check(_lib.load().schpf_shape_update(
        _code(dt), nnz, nfactors, _p(_c(Xphi_data, dt)), _p(_c(X_keep, np.int32)), int(nkeep),
        float(shape_prior), _p(out)))
    return out


def compute_loading_rate_update(prior_vi_shape, prior_vi_rate,
                                other_loading_vi_shape, other_loading_vi_rate):
    """Gamma rate update for theta or beta (hpf_numba.py:159-177)."""
    dt = prior_vi_shape.dtype
    n = prior_vi_shape.shape[0]
    m, nfactors = other_loading_vi_shape.shape
    out = np.empty((n, nfactors), dtype=dt)
    _lib.check(_lib.load().schpf_rate_update(
        _code(dt), n, m, nfactors, _p(_c(prior_vi_shape, dt)), _p(_c(prior_vi_rate, dt)),
        _p(_c(other_loading_vi_shape, dt)), _p(_c(other_loading_vi_rate, dt)), _p(out)))
    return out


def compute_capacity_rate_update(loading_vi_shape, loading_vi_rate, prior_rate):
    """Gamma rate update for xi or eta (hpf_numba.py:180-188)."""
    dt = loading_vi_shape.dtype
    n, nfactors = loading_vi_shape.shape
    out = np.empty((n,), dtype=dt)
    _lib.check(_lib.load().schpf_capacity_rate_update(
        _code(dt), n, nfactors, _p(_c(loading_vi_shape, dt)), _p(_c(loading_vi_rate, dt)),
        float(prior_rate), _p(out)))
    return out


def coo_marginals(X):
    """(row sums, column sums) of a COO matrix as float64 vectors: X.sum(1), X.sum(0) of
    scHPF._get_empirical_hypers (scHPF_.py:847-879) in one threaded host pass."""
    import ctypes
    lib = _lib.load()
    if not hasattr(X, "row"):
        X = X.tocoo()
    kinds = {np.dtype(np.int32): _lib.VAL_I32, np.dtype(np.int64): _lib.VAL_I64,
             np.dtype(np.float32): _lib.VAL_F32, np.dtype(np.float64): _lib.VAL_F64}
    data = np.ascontiguousarray(X.data)
    if data.dtype not in kinds:
        data = data.astype(np.float64)
    row = np.ascontiguousarray(X.row, dtype=np.int32)
    col = np.ascontiguousarray(X.col, dtype=np.int32)
    rs = np.empty(X.shape[0], dtype=np.float64)
    cs = np.empty(X.shape[1], dtype=np.float64)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)   # noqa: E731
    _lib.check(lib.schpf_coo_marginals(data.shape[0], p(row), p(col), p(data), kinds[data.dtype],
                                       X.shape[0], X.shape[1], p(rs), p(cs)))
    return rs, cs
