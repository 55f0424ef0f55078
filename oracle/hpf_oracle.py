"""CPU ORACLE for the scHPF CAVI hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates, on the CPU, what the reference computes on the path this repository
accelerates.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this module; nothing under schpf_amd/ does.

Two layers:
  * liboracle.so (oracle/cavi_oracle.c): the kernels of
    /root/reference/schpf/hpf_numba.py and the per-iteration update order of
    schpf/scHPF_.py:657-714 in plain C, in the reference's execution shape.
  * this file: ctypes bindings, numpy second opinions of the same formulas
    (following the reference's numpy fallbacks hpf_numba.py:117-125 and
    loss.py:132-134), and `oracle_fit`, a restatement of scHPF._setup/_fit
    host logic (scHPF_.py:526-879): RNG draw order, empirical bp/dp, loss
    cadence and the stop rules.

Pinned against the reference's own outputs in tests/golden/ by
tests/test_oracle_golden.py.
"""
import ctypes
import os
import subprocess

import numpy as np
from scipy.special import digamma, gammaln, logsumexp

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None

_c_int_p = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    """Compile liboracle.so with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(os.path.join(_HERE, f))
                                              for f in ("cavi_oracle.c", "cavi_oracle_impl.h",
                                                        "cavi_fused.c", "cavi_fused_impl.h"))):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_psi.restype = ctypes.c_double
        _lib.orc_psi.argtypes = [ctypes.c_double]
        _lib.orc_gammaln.restype = ctypes.c_double
        _lib.orc_gammaln.argtypes = [ctypes.c_double]
    return _lib


def _suffix(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return "_f64"
    if dtype == np.float32:
        return "_f32"
    raise TypeError("oracle supports float64/float32, got %s" % dtype)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


# ---------------------------------------------------------------- scalars ---
def psi(x):
    x = np.atleast_1d(np.asarray(x, dtype=np.float64))
    out = np.empty_like(x)
    lib().orc_psi_array(ctypes.c_long(x.size), _p(x), _p(out))
    return out


def cgammaln(x):
    x = np.atleast_1d(np.asarray(x, dtype=np.float64))
    out = np.empty_like(x)
    lib().orc_gammaln_array(ctypes.c_long(x.size), _p(x), _p(out))
    return out


# ---------------------------------------------------------------- kernels ---
def compute_Xphi_data(X_data, X_row, X_col, theta_vi_shape, theta_vi_rate,
                      beta_vi_shape, beta_vi_rate, nthreads=1):
    """hpf_numba.py:54-114."""
    dt = theta_vi_shape.dtype
    N, K = theta_vi_shape.shape
    G = beta_vi_shape.shape[0]
    nnz = X_data.shape[0]
    out = np.empty((nnz, K), dtype=dt)
    getattr(lib(), "orc_xphi" + _suffix(dt))(
        ctypes.c_long(nnz), N, G, K, _p(_c(X_data, dt)), _p(_c(X_row, np.int32)),
        _p(_c(X_col, np.int32)), _p(_c(theta_vi_shape, dt)), _p(_c(theta_vi_rate, dt)),
        _p(_c(beta_vi_shape, dt)), _p(_c(beta_vi_rate, dt)), _p(out), int(nthreads))
    return out


def compute_Xphi_data_numpy(x, row, col, theta_shape, theta_rate, beta_shape, beta_rate):
    """hpf_numba.py:117-125 with HPF_Gamma.e_logx (scHPF_.py:107-111) inlined."""
    elt = digamma(theta_shape) - np.log(theta_rate)
    elb = digamma(beta_shape) - np.log(beta_rate)
    logrho = elt[row, :] + elb[col, :]
    logphi = logrho - logsumexp(logrho, axis=1)[:, None]
    return x[:, None] * np.exp(logphi)


def compute_loading_shape_update(Xphi_data, X_keep, nkeep, shape_prior, scatter_threads=1):
    """hpf_numba.py:128-156.  scatter_threads > 1: the same sums in the same order, destination rows split over
    threads (cavi_oracle_impl.h orc_shape_update) -- bit-identical to the serial loop."""
    dt = Xphi_data.dtype
    nnz, K = Xphi_data.shape
    out = np.empty((nkeep, K), dtype=dt)
    getattr(lib(), "orc_shape_update" + _suffix(dt))(
        ctypes.c_long(nnz), K, _p(_c(Xphi_data, dt)), _p(_c(X_keep, np.int32)), int(nkeep),
        ctypes.c_double(shape_prior), _p(out), int(scatter_threads))
    return out


def compute_loading_rate_update(prior_vi_shape, prior_vi_rate, other_vi_shape, other_vi_rate):
    """hpf_numba.py:159-177."""
    dt = prior_vi_shape.dtype
    n = prior_vi_shape.shape[0]
    m, K = other_vi_shape.shape
    out = np.empty((n, K), dtype=dt)
    getattr(lib(), "orc_rate_update" + _suffix(dt))(
        n, m, K, _p(_c(prior_vi_shape, dt)), _p(_c(prior_vi_rate, dt)),
        _p(_c(other_vi_shape, dt)), _p(_c(other_vi_rate, dt)), _p(out))
    return out


def compute_capacity_rate_update(loading_vi_shape, loading_vi_rate, prior_rate):
    """hpf_numba.py:180-188."""
    dt = loading_vi_shape.dtype
    n, K = loading_vi_shape.shape
    out = np.empty((n,), dtype=dt)
    getattr(lib(), "orc_capacity_rate" + _suffix(dt))(
        n, K, _p(_c(loading_vi_shape, dt)), _p(_c(loading_vi_rate, dt)),
        ctypes.c_double(prior_rate), _p(out))
    return out


def compute_pois_llh(X_data, X_row, X_col, theta_vi_shape, theta_vi_rate,
                     beta_vi_shape, beta_vi_rate, nthreads=1):
    """hpf_numba.py:24-51."""
    dt = theta_vi_shape.dtype
    N, K = theta_vi_shape.shape
    G = beta_vi_shape.shape[0]
    nnz = X_data.shape[0]
    out = np.empty((nnz,), dtype=dt)
    getattr(lib(), "orc_pois_llh" + _suffix(dt))(
        ctypes.c_long(nnz), N, G, K, _p(_c(X_data, dt)), _p(_c(X_row, np.int32)),
        _p(_c(X_col, np.int32)), _p(_c(theta_vi_shape, dt)), _p(_c(theta_vi_rate, dt)),
        _p(_c(beta_vi_shape, dt)), _p(_c(beta_vi_rate, dt)), _p(out), int(nthreads))
    return out


def pois_llh_numpy(x, row, col, theta_shape, theta_rate, beta_shape, beta_rate):
    """loss.py:132-134 (the single_process branch)."""
    e_rate = ((theta_shape / theta_rate)[row] * (beta_shape / beta_rate)[col]).sum(axis=1)
    return x * np.log(e_rate) - e_rate - gammaln(x + 1)


def mean_negative_pois_llh(x, row, col, ths, thr, bes, ber, nthreads=1):
    """loss.py:142-168: mean over the stored nonzeros only."""
    return np.mean(-compute_pois_llh(x, row, col, ths, thr, bes, ber, nthreads))


# ------------------------------------------------------------ iteration -----
class State(object):
    """The eight variational arrays + hypers, plain numpy."""

    def __init__(self, xi_shape, xi_rate, theta_shape, theta_rate,
                 eta_shape, eta_rate, beta_shape, beta_rate):
        self.xi_shape, self.xi_rate = xi_shape, xi_rate
        self.theta_shape, self.theta_rate = theta_shape, theta_rate
        self.eta_shape, self.eta_rate = eta_shape, eta_rate
        self.beta_shape, self.beta_rate = beta_shape, beta_rate

    def copy(self):
        return State(*[a.copy() for a in self.arrays()])

    def arrays(self):
        return [self.xi_shape, self.xi_rate, self.theta_shape, self.theta_rate,
                self.eta_shape, self.eta_rate, self.beta_shape, self.beta_rate]

    def cast(self, dtype):
        return State(*[np.ascontiguousarray(a, dtype=dtype) for a in self.arrays()])


def cavi_iteration(x, row, col, st, a, c, bp, dp, xphi=None, freeze_genes=False,
                   simultaneous=False, nthreads=1, scatter_threads=1):
    """One iteration, scHPF_.py:657-714 (non-batched), in place on `st`.

    `xphi`: optional (nnz, K) array holding X*phi to use instead of
    compute_Xphi_data (the t==0 random responsibilities, scHPF_.py:652-655).
    All state arrays must share one dtype and be C-contiguous.
    `scatter_threads` > 1 runs the two scatter-adds (serial in the reference, hpf_numba.py:128) split by
    destination row -- the same bits, sooner: for the parity tests at BASELINE sizes; bench.py's CPU baseline
    keeps the reference's serial loops (1).
    """
    dt = st.theta_shape.dtype
    N, K = st.theta_shape.shape
    G = st.beta_shape.shape[0]
    nnz = x.shape[0]
    for arr in st.arrays():
        assert arr.dtype == dt and arr.flags.c_contiguous
    if xphi is None:
        ws = np.empty((nnz, K), dtype=dt)
        given = 0
    else:
        ws = np.ascontiguousarray(xphi, dtype=dt)
        given = 1
    getattr(lib(), "orc_cavi_iteration" + _suffix(dt))(
        ctypes.c_long(nnz), N, G, K, _p(_c(x, dt)), _p(_c(row, np.int32)), _p(_c(col, np.int32)),
        ctypes.c_double(a), ctypes.c_double(c), ctypes.c_double(bp), ctypes.c_double(dp),
        _p(st.xi_shape), _p(st.xi_rate), _p(st.theta_shape), _p(st.theta_rate),
        _p(st.eta_shape), _p(st.eta_rate), _p(st.beta_shape), _p(st.beta_rate),
        _p(ws), given, int(bool(freeze_genes)), int(bool(simultaneous)), int(nthreads),
        int(scatter_threads))
    return st


class FusedMatrix(object):
    """CSR + CSC copies of X for the fused CPU comparator (built once, like the GPU's plans)."""

    def __init__(self, X, dtype):
        csr = X.tocsr()
        csc = X.tocsc()
        self.shape = X.shape
        self.dtype = np.dtype(dtype)
        self.rptr = np.ascontiguousarray(csr.indptr, dtype=np.int64)
        self.rcol = np.ascontiguousarray(csr.indices, dtype=np.int32)
        self.rval = np.ascontiguousarray(csr.data, dtype=dtype)
        self.cptr = np.ascontiguousarray(csc.indptr, dtype=np.int64)
        self.crow = np.ascontiguousarray(csc.indices, dtype=np.int32)
        self.cval = np.ascontiguousarray(csc.data, dtype=dtype)


def fused_iteration(M, st, a, c, bp, dp, nthreads=1):
    """SURVEY.md 8(d) CPU variant (ii), "fused OpenMP": one default-order iteration
    (scHPF_.py:697-714) with exp hoisted, no Xphi, a parallel CSR pass and a parallel CSC pass
    (oracle/cavi_fused_impl.h).  In place on `st`; M is a FusedMatrix."""
    dt = st.theta_shape.dtype
    assert dt == M.dtype
    N, K = st.theta_shape.shape
    G = st.beta_shape.shape[0]
    assert K <= 256 and (N, G) == tuple(M.shape)
    for arr in st.arrays():
        assert arr.dtype == dt and arr.flags.c_contiguous
    getattr(lib(), "orc_fused_iteration" + _suffix(dt))(
        N, G, K, _p(M.rptr), _p(M.rcol), _p(M.rval), _p(M.cptr), _p(M.crow), _p(M.cval),
        ctypes.c_double(a), ctypes.c_double(c), ctypes.c_double(bp), ctypes.c_double(dp),
        _p(st.xi_shape), _p(st.xi_rate), _p(st.theta_shape), _p(st.theta_rate),
        _p(st.eta_shape), _p(st.eta_rate), _p(st.beta_shape), _p(st.beta_rate), int(nthreads))
    return st


# ----------------------------------------------------- host logic of _fit ---
def mean_var_ratio(X, axis):
    """scHPF_.py:863-865 (population variance, np.var default ddof=0)."""
    axis_sum = np.asarray(X.sum(axis=axis))
    return np.mean(axis_sum) / np.var(axis_sum)


def empirical_hypers(X, ap, cp, bp=None, dp=None, freeze_genes=False, clip=True):
    """scHPF._get_empirical_hypers, scHPF_.py:847-879."""
    if bp is None:
        bp = ap * mean_var_ratio(X, axis=1)
    if dp is None:
        if freeze_genes:
            raise ValueError("dp is None and cannot be set when freeze_genes is True.")
        dp = cp * mean_var_ratio(X, axis=0)
        if clip and bp > 1000 * dp:
            dp = bp / 1000
    return bp, dp


def random_gamma(dims, shape_prior, rate_prior, dtype):
    """HPF_Gamma.random_gamma_factory, scHPF_.py:49-70: shape draw, then rate draw."""
    s = np.random.uniform(0.5 * shape_prior, 1.5 * shape_prior, dims).astype(dtype)
    r = np.random.uniform(0.5 * rate_prior, 1.5 * rate_prior, dims).astype(dtype)
    return s, r


def setup_state(X, K, dtype, a, ap, c, cp, bp=None, dp=None, frozen=None):
    """scHPF._setup with reinit=True, scHPF_.py:783-844.  Draw order: xi, theta, eta, beta.

    `frozen` = (eta_shape, eta_rate, beta_shape, beta_rate) for freeze_genes.
    """
    N, G = X.shape
    bp, dp = empirical_hypers(X, ap, cp, bp, dp, freeze_genes=frozen is not None)
    xis, xir = random_gamma((N,), ap, bp, dtype)
    ths, thr = random_gamma((N, K), a, bp, dtype)
    if frozen is None:
        ets, etr = random_gamma((G,), cp, dp, dtype)
        bes, ber = random_gamma((G, K), c, dp, dtype)
    else:
        ets, etr, bes, ber = [np.array(v, dtype=dtype) for v in frozen]
    return bp, dp, State(xis, xir, ths, thr, ets, etr, bes, ber)


def oracle_fit(X, K, dtype=np.float64, a=0.3, ap=1.0, c=0.3, cp=1.0, bp=None, dp=None,
               min_iter=30, max_iter=1000, check_freq=10, epsilon=0.001,
               better_than_n_ago=5, frozen=None, simultaneous=False, nthreads=1,
               self_max_iter=None, scatter_threads=1):
    """Restatement of scHPF._fit (scHPF_.py:526-780) for reinit=True, no minibatching.

    Uses the global np.random state exactly as the reference does (seed it
    before calling).  Returns dict(bp, dp, state, loss).

    dtype handling: for a float32 model the reference runs the t==0 update from
    float64 random responsibilities (scHPF_.py:653-655); here the t==0 update
    is done in float64 and rounded to the model dtype afterwards, later
    iterations run in the model dtype.  That reproduces the reference to within
    float32 rounding (documented in DESIGN.md, "dtype quirks").
    """
    X = X.tocoo() if not hasattr(X, "row") else X
    dtype = np.dtype(dtype)
    x, row, col = X.data, X.row, X.col
    freeze = frozen is not None
    bp, dp, st = setup_state(X, K, dtype, a, ap, c, cp, bp, dp, frozen)
    st.xi_shape[:] = ap + K * a                                   # scHPF_.py:616
    if not freeze:
        st.eta_shape[:] = cp + K * c                              # :618
    self_max_iter = max_iter if self_max_iter is None else self_max_iter

    loss, pct_change = [], []
    for t in range(max_iter):
        if t == 0:                                                # :652-655
            random_phi = np.random.dirichlet(np.ones(K), x.shape[0])
            xphi = x[:, None] * random_phi
            st64 = st.cast(np.float64)
            cavi_iteration(x, row, col, st64, a, c, bp, dp, xphi=xphi, freeze_genes=freeze,
                           simultaneous=simultaneous, nthreads=nthreads, scatter_threads=scatter_threads)
            st = st64.cast(dtype)
        else:
            cavi_iteration(x, row, col, st, a, c, bp, dp, freeze_genes=freeze,
                           simultaneous=simultaneous, nthreads=nthreads, scatter_threads=scatter_threads)

        if t % check_freq == 0:                                   # :718-744
            curr = float(mean_negative_pois_llh(x, row, col, st.theta_shape, st.theta_rate,
                                                st.beta_shape, st.beta_rate, nthreads))
            loss.append(curr)
            if len(loss) >= 2:
                prev = loss[-2]
                pct_change.append(100 * (curr - prev) / np.abs(prev))
            else:
                pct_change.append(100)
            if len(loss) > 3 and t >= min_iter:                   # :750-774
                prev = loss[-2]
                current_small = np.abs(pct_change[-1]) < epsilon
                prev_small = np.abs(pct_change[-2]) < epsilon
                not_inflection = not ((np.abs(loss[-3]) < np.abs(prev))
                                      and (np.abs(prev) > np.abs(curr)))
                if current_small and prev_small and not_inflection:
                    break
                if len(loss) > better_than_n_ago and better_than_n_ago:
                    nprev = loss[-better_than_n_ago]
                    if np.abs(nprev) < np.abs(curr) and np.abs(prev) < np.abs(curr):
                        break
        if t >= self_max_iter:                                    # :777
            break
    return dict(bp=bp, dp=dp, state=st, loss=loss)
