"""scHPF estimator with the CAVI loop running on an MI355X.

Host-side mirror of /root/reference/schpf/scHPF_.py for the hot path: the
`HPF_Gamma` container (:27-178), the `scHPF` estimator (:181-892) with the same
constructor, properties, `fit` / `project` / `cell_score` / `gene_score` / llh
helpers and `_fit` keyword arguments, and `load_model` / `save_model` /
`combine_across_cells` (:895-965).  What differs is where the work happens: `_fit`
uploads the count matrix and the four Gammas once, runs every iteration and every
default loss evaluation as HIP kernels (schpf_amd.engine.DeviceCAVI) and downloads
the result at the end.  The NumPy global RNG is consumed in the reference's order
(:49-70, :819-842, :652-655) so that equal seeds give equal initialisations.

The classes pickle under the reference's module path (`schpf.scHPF_`), so joblib
model files are interchangeable with the reference (see the `schpf` alias package).
"""
from copy import deepcopy
from warnings import warn

import joblib
import numpy as np
from scipy.sparse import coo_matrix
from scipy.special import digamma, gammaln
from sklearn.base import BaseEstimator

from . import _lib
from . import hpf_hip
from . import loss as ls
from ._version import __version__
from .engine import DeviceCAVI

__all__ = ["HPF_Gamma", "scHPF", "load_model", "save_model", "combine_across_cells"]

_REFERENCE_MODULE = "schpf.scHPF_"
# above this many float64 draws the t=0 responsibilities are generated on the GPU
_HOST_PHI_LIMIT = 1 << 28


class HPF_Gamma(object):
    """A block of independent variational Gamma distributions.

    Same contract as the reference's container (scHPF_.py:27-178): two equal-shape,
    equal-dtype, strictly positive arrays `vi_shape`, `vi_rate`.
    """

    @staticmethod
    def random_gamma_factory(dims, shape_prior, rate_prior, dtype=np.float64):
        """Uniform jitter around the priors, shape drawn before rate (scHPF_.py:49-70)."""
        lo_hi = lambda v: (0.5 * v, 1.5 * v)  # noqa: E731
        vi_shape = np.random.uniform(*lo_hi(shape_prior), dims).astype(dtype)
        vi_rate = np.random.uniform(*lo_hi(rate_prior), dims).astype(dtype)
        return HPF_Gamma(vi_shape, vi_rate)

    def __init__(self, vi_shape, vi_rate):
        assert vi_shape.shape == vi_rate.shape
        assert vi_shape.dtype == vi_rate.dtype
        assert np.all(vi_shape > 0)
        assert np.all(vi_rate > 0)
        self.vi_shape = vi_shape
        self.vi_rate = vi_rate
        self.dtype = vi_shape.dtype

    def __eq__(self, other):
        if not isinstance(other, self.__class__):
            return False
        return (np.array_equal(self.vi_shape, other.vi_shape)
                and np.array_equal(self.vi_rate, other.vi_rate)
                and self.dtype == other.dtype)

    __hash__ = None

    @property
    def dims(self):
        assert self.vi_shape.shape == self.vi_rate.shape
        return self.vi_shape.shape

    @property
    def e_x(self):
        """E[x] = shape / rate."""
        return self.vi_shape / self.vi_rate

    @property
    def e_logx(self):
        """E[log x] = psi(shape) - log(rate)."""
        return digamma(self.vi_shape) - np.log(self.vi_rate)

    @property
    def entropy(self):
        s, r = self.vi_shape, self.vi_rate
        return s - np.log(r) + gammaln(s) + (1 - s) * digamma(s)

    def sample(self, nsamples=1):
        """Draws from the variational distributions; last axis indexes the sample."""
        draws = [np.random.gamma(self.vi_shape, 1 / self.vi_rate).T for _ in range(nsamples)]
        return np.stack(draws).T

    def combine(self, other, other_ixs):
        """Interleave with `other`, whose rows land at `other_ixs` (scHPF_.py:142-178)."""
        assert other.dims[0] == len(other_ixs)
        assert len(np.unique(other_ixs)) == len(other_ixs)
        total = self.dims[0] + other.dims[0]
        assert total > np.max(other_ixs)
        mine = np.setdiff1d(np.arange(total), other_ixs)
        merged = []
        for own, theirs in ((self.vi_shape, other.vi_shape), (self.vi_rate, other.vi_rate)):
            out = np.empty((total,) + tuple(self.dims[1:]), dtype=self.dtype)
            out[mine] = own
            out[other_ixs] = theirs
            merged.append(out)
        return HPF_Gamma(*merged)


class _LossMonitor(object):
    """Loss bookkeeping and the two stop rules of scHPF._fit (scHPF_.py:718-774)."""

    def __init__(self, epsilon, better_than_n_ago, min_iter, smoothing):
        self.epsilon = epsilon
        self.better_than_n_ago = better_than_n_ago
        self.min_iter = min_iter
        self.smoothing = smoothing
        self.loss, self.pct_change, self._window = [], [], []

    def record(self, value):
        self._window.append(value)
        if len(self._window) > self.smoothing:
            self._window = self._window[1:]
        self.loss.append(np.mean(self._window))
        if len(self.loss) >= 2:
            curr, prev = self.loss[-1], self.loss[-2]
            self.pct_change.append(100 * (curr - prev) / np.abs(prev))
        else:
            self.pct_change.append(100)
        return self.loss[-1], self.pct_change[-1]

    def verdict(self, t):
        """None, 'converged' or 'getting worse break'."""
        loss = self.loss
        if not (len(loss) > 3 and t >= self.min_iter):
            return None
        curr, prev = loss[-1], loss[-2]
        small_now = np.abs(self.pct_change[-1]) < self.epsilon
        small_before = np.abs(self.pct_change[-2]) < self.epsilon
        inflection = (np.abs(loss[-3]) < np.abs(prev)) and (np.abs(prev) > np.abs(curr))
        if small_now and small_before and not inflection:
            return "converged"
        n = self.better_than_n_ago
        if n and len(loss) > n:
            if np.abs(loss[-n]) < np.abs(curr) and np.abs(prev) < np.abs(curr):
                return "getting worse break"
        return None


class _EarlyUpload(object):
    """Engine creation + upload of X on a helper thread while the caller initialises the model."""

    def __init__(self, X, nfactors, dtype, device):
        import threading
        self._out = {}

        def work():
            eng = None
            try:
                eng = DeviceCAVI(X.shape[0], X.shape[1], nfactors, dtype=dtype, device=device)
                # warnings.catch_warnings mutates process-global state and is not thread-safe: the
                # worker never touches it; result() issues the upload's warning on the caller's thread
                eng.upload(X, warn=False)
                self._out["engine"] = eng
            except BaseException as exc:     # re-raised on the caller's thread
                if eng is not None:
                    eng.close()
                self._out["error"] = exc
        self._thread = threading.Thread(target=work, name="schpf-upload")
        self._thread.start()

    def result(self):
        self._thread.join()
        if "error" in self._out:
            raise self._out["error"]
        self._out["engine"].rounding_warning(stacklevel=3)      # "values were rounded to float32"
        return self._out["engine"]

    def abandon(self):
        self._thread.join()
        eng = self._out.get("engine")
        if eng is not None:
            eng.close()


class scHPF(BaseEstimator):
    """Single-cell hierarchical Poisson factorization (Levitin et al., MSB 2019).

    Constructor arguments, attributes and defaults are the reference's
    (scHPF_.py:181-269): nfactors, a, ap, bp, c, cp, dp, min_iter, max_iter,
    check_freq, epsilon, better_than_n_ago, dtype, xi, theta, eta, beta, loss,
    verbose.  `a` / `c` equal to -2 mean 1/sqrt(nfactors).
    """

    def __init__(self, nfactors, a=0.3, ap=1, bp=None, c=0.3, cp=1, dp=None, min_iter=30,
                 max_iter=1000, check_freq=10, epsilon=0.001, better_than_n_ago=5,
                 dtype=np.float64, xi=None, theta=None, eta=None, beta=None, loss=[],
                 verbose=True):
        self.version = __version__
        self.nfactors = nfactors
        self.a = a
        self.ap = ap
        self.bp = bp
        self.c = c
        self.cp = cp
        self.dp = dp
        self.min_iter = min_iter
        self.max_iter = max_iter
        self.check_freq = check_freq
        self.epsilon = epsilon
        self.better_than_n_ago = better_than_n_ago
        self.dtype = dtype
        self.verbose = verbose
        self.xi = xi
        self.eta = eta
        self.theta = theta
        self.beta = beta
        self.loss = []      # the constructor argument is ignored, as in the reference (:269)

    # ---- a / c: stored as _a / _c; -2 selects 1/sqrt(K); pre-0.5 files lack them (:272-319)
    def _shape_prior_get(self, attr, letter):
        try:
            return getattr(self, attr)
        except AttributeError:
            warn("Automatically using {0}=0.3. If you are loading a model generated with scHPF "
                 "version < 0.5 and set a custom value for {0}, you must manually reset it and "
                 "re-save the model.".format(letter), RuntimeWarning)
            return 0.3

    def _shape_prior_set(self, attr, val):
        if val == -2:
            if self.nfactors is None:
                raise ValueError("Can only set a as a function of nfactors when nfactors is not None")
            setattr(self, attr, 1 / np.sqrt(self.nfactors))
        else:
            assert val > 0
            setattr(self, attr, val)

    a = property(lambda self: self._shape_prior_get("_a", "a"),
                 lambda self, v: self._shape_prior_set("_a", v))
    c = property(lambda self: self._shape_prior_get("_c", "c"),
                 lambda self, v: self._shape_prior_set("_c", v))

    @property
    def ngenes(self):
        return self.eta.dims[0] if self.eta is not None else None

    @property
    def ncells(self):
        return self.xi.dims[0] if self.xi is not None else None

    # ------------------------------------------------------------------- scores
    def cell_score(self, xi=None, theta=None):
        """ncells x nfactors hierarchically normalised cell loadings (scHPF_.py:332-348)."""
        return self._score(self.xi if xi is None else xi, self.theta if theta is None else theta)

    def gene_score(self, eta=None, beta=None):
        """ngenes x nfactors hierarchically normalised gene loadings (scHPF_.py:351-369)."""
        return self._score(self.eta if eta is None else eta, self.beta if beta is None else beta)

    def _score(self, capacity, loading):
        assert loading.dims[0] == capacity.dims[0]
        return loading.e_x * capacity.e_x[:, None]

    # --------------------------------------------------------------- likelihoods
    def pois_llh_pointwise(self, X, theta=None, beta=None):
        """Poisson log-likelihood of every stored nonzero of X (scHPF_.py:372-392)."""
        theta = self.theta if theta is None else theta
        beta = self.beta if beta is None else beta
        return ls.pois_llh_pointwise(X=X, theta=theta, beta=beta)

    def cellmean_negative_pois_llh(self, X, theta=None, beta=None):
        """Mean negative llh of the nonzeros of each cell (scHPF_.py:395-411)."""
        theta = self.theta if theta is None else theta
        assert theta.vi_shape.shape[0] == X.shape[0]
        beta = self.beta if beta is None else beta
        neg = -self.pois_llh_pointwise(X=X, theta=theta, beta=beta)
        as_csr = coo_matrix((neg, (X.row, X.col)), shape=X.shape).tocsr()
        sums = np.asarray(as_csr.sum(axis=1)).ravel()
        counts = np.diff(as_csr.indptr)
        averages = sums / counts
        assert averages.shape[0] == theta.vi_shape.shape[0]
        return averages

    def mean_negative_pois_llh(self, X, theta=None, beta=None, **kwargs):
        """Mean negative llh over the nonzeros of X (scHPF_.py:416-422)."""
        theta = self.theta if theta is None else theta
        beta = self.beta if beta is None else beta
        return ls.mean_negative_pois_llh(X=X, theta=theta, beta=beta)

    # ---------------------------------------------------------------- fit/project
    def fit(self, X, **kwargs):
        """Fit the model to the cell x gene count matrix X (scHPF_.py:425-445)."""
        (self.bp, self.dp, self.xi, self.eta, self.theta, self.beta, self.loss) = self._fit(X, **kwargs)
        return self

    def project(self, X, recalc_bp=False, replace=False, min_iter=2, max_iter=50, check_freq=2,
                **kwargs):
        """Fit xi/theta of new cells against frozen eta/beta (scHPF_.py:448-503)."""
        if replace and recalc_bp:
            raise ValueError("Cannot replace `bp` with recalculated value")
        model = self if replace else deepcopy(self)
        if recalc_bp:
            model.bp = None
        bp, _, xi, _, theta, _, loss = model._fit(X, min_iter=min_iter, max_iter=max_iter,
                                                 check_freq=check_freq, freeze_genes=True, **kwargs)
        if replace:
            self.xi, self.theta = xi, theta
            return loss
        model.bp, model.xi, model.theta, model.loss = bp, xi, theta, loss
        return model

    def _fit(self, X, freeze_genes=False, reinit=True, loss_function=None, min_iter=None,
             max_iter=None, epsilon=None, check_freq=None, single_process=False,
             checkstep_function=None, verbose=None, batchsize=None,
             beta_theta_simultaneous=False, loss_smoothing=1, device=None, init="auto", engine=None,
             devices=None):
        """The CAVI loop (scHPF_.py:526-780) on the GPU.

        Keyword arguments are the reference's.  `single_process` is accepted and
        ignored (there is one execution path, the device).  Two additions:
        `device` (HIP device ordinal, default $SCHPF_DEVICE or 0) and `init`
        ('numpy': t=0 responsibilities drawn with the NumPy global RNG exactly like
        the reference; 'device': drawn on the GPU; 'auto': numpy unless nnz*K is
        beyond what a host draw can reasonably do); `engine`, a DeviceCAVI that already holds
        X (run_trials reuses one upload for all restarts); `devices`, a list of HIP device
        ordinals: with more than one the cells are row-sharded over those GPUs (nnz-balanced
        blocks), every iteration does one RCCL all-reduce of the gene-side sums
        (schpf_amd.sharded.ThreadedShards) and the result equals the single-GPU fit up to the
        summation order of those sums.
        `batchsize` (minibatch CAVI, scHPF_.py:626-650, 688-695) runs every iteration on the
        device too: the matrix goes up once and each batch's rows are gathered in HBM
        (_fit_minibatch); only when the whole matrix does not fit beside its plans are the
        batch's rows sliced on the host and uploaded per iteration, like the reference re-slices.
        Returns (bp, dp, xi, eta, theta, beta, loss) like the reference.
        """
        assert loss_smoothing > 0
        if not hasattr(X, "row"):
            X = X.tocoo()
        batched = batchsize is not None and 1 < batchsize <= X.shape[0]
        nfactors, (ncells, ngenes) = self.nfactors, X.shape
        a, ap, c, cp = self.a, self.ap, self.c, self.cp

        if device is None:
            import os
            device = int(os.environ.get("SCHPF_DEVICE", "0"))
        if devices is not None and len(devices) == 1:
            device = int(devices[0])
        model_dtype = np.dtype(self.dtype)
        # The upload (validation, PCIe copy, both plans built on the device: ~0.1 s at 1e8 nonzeros)
        # does not depend on the initialisation, and the initialisation (marginals for bp/dp, the
        # uniform draws for the four Gammas: ~0.06 s) does not depend on the device: run them side by
        # side.  The library calls release the GIL; NumPy's RNG is only touched on this thread.
        early = None
        if engine is None and not batched and not (devices is not None and len(devices) > 1):
            early = _EarlyUpload(X, nfactors, model_dtype, device)
        try:
            bp, dp, xi, eta, theta, beta = self._setup(X, freeze_genes, reinit)
        except BaseException:
            if early is not None:
                early.abandon()
            raise
        # the hierarchical shapes are constants of the model (scHPF_.py:616-618)
        xi.vi_shape[:] = ap + nfactors * a
        if not freeze_genes:
            eta.vi_shape[:] = cp + nfactors * c

        min_iter = self.min_iter if min_iter is None else min_iter
        max_iter = self.max_iter if max_iter is None else max_iter
        check_freq = self.check_freq if check_freq is None else check_freq
        verbose = self.verbose if verbose is None else verbose
        # NB the reference computes an `epsilon` override and then tests self.epsilon
        # (scHPF_.py:639 vs :752-753); reproduced so iteration counts match.
        monitor = _LossMonitor(self.epsilon, self.better_than_n_ago, min_iter, loss_smoothing)

        if batched:
            if devices is not None and len(devices) > 1:
                raise ValueError("minibatch fits (batchsize=...) run on one device: a batch is a row subset that one "
                                 "GPU holds whole; give one device or drop batchsize")
            return self._fit_minibatch(X, bp, dp, xi, eta, theta, beta, monitor, freeze_genes, reinit,
                                       loss_function, max_iter, check_freq, checkstep_function, verbose,
                                       batchsize, beta_theta_simultaneous, device)
        own_engine = engine is None
        sharded = own_engine and devices is not None and len(devices) > 1
        if sharded:
            from .sharded import ThreadedShards
            import os
            eng = ThreadedShards(X, nfactors, model_dtype, devices,      # uploads its row blocks itself
                                 comm=os.environ.get("SCHPF_SHARD_COMM", "rccl"))
        elif early is not None:
            eng = early.result()                     # the engine with X uploaded (raises what the upload raised)
        else:
            eng = DeviceCAVI(ncells, ngenes, nfactors, dtype=model_dtype, device=device) if own_engine else engine
        if not own_engine and ((eng.ncells, eng.ngenes, eng.nfactors) != (ncells, ngenes, nfactors)
                               or eng.dtype != model_dtype or eng.nnz != X.data.shape[0]):
            raise ValueError("engine was built for a different matrix, nfactors or dtype")
        try:
            if own_engine and not sharded and early is None:
                eng.upload(X)
            eng.set_hypers(a, c, bp, dp)
            for name, g in (("xi", xi), ("theta", theta), ("eta", eta), ("beta", beta)):
                eng.set_gamma(name, g.vi_shape, g.vi_rate)

            def download():
                return [HPF_Gamma(*eng.get_gamma(n)) for n in ("xi", "eta", "theta", "beta")]

            # The reference's loop is `for t in range(max_iter): iterate; if t % check_freq == 0: check`
            # (scHPF_.py:642-778).  Nothing on the host changes between two checks, so the stretch
            # t .. next check runs as ONE library call (schpf_steps: one hipGraph launch).
            t = 0
            while t < max_iter:
                if t == 0 and reinit:   # random responsibilities, scHPF_.py:652-655
                    use_host = init == "numpy" or (init == "auto"
                                                   and X.data.shape[0] * nfactors <= _HOST_PHI_LIMIT)
                    if use_host:
                        random_phi = np.random.dirichlet(np.ones(nfactors), X.data.shape[0])
                        eng.init_phi_host(X.data[:, None] * random_phi)
                        del random_phi
                    else:
                        eng.init_phi_device(np.random.randint(0, 2 ** 31 - 1))
                check_at = t if t % check_freq == 0 else (t // check_freq + 1) * check_freq
                # (the reference also leaves after iteration self.max_iter when the local override is larger, :777)
                last = min(check_at, max_iter - 1, max(self.max_iter, t))
                eng.steps(last - t + 1, freeze_genes=freeze_genes, simultaneous=beta_theta_simultaneous)
                t = last

                if t % check_freq == 0:
                    if loss_function is None and checkstep_function is None:
                        curr = eng.mean_negative_pois_llh()
                    else:
                        xi, eta, theta, beta = download()
                        if loss_function is None:
                            curr = eng.mean_negative_pois_llh()
                        else:
                            curr = loss_function(a=a, ap=ap, bp=bp, c=c, cp=cp, dp=dp, xi=xi,
                                                 eta=eta, theta=theta, beta=beta)
                    curr, pct = monitor.record(curr)
                    if verbose:
                        print("[Iter. {0: >4}]  loss:{1:.6f}  pct:{2:.9f}".format(t, curr, pct))
                    if checkstep_function is not None:
                        checkstep_function(bp=bp, dp=dp, xi=xi, eta=eta, theta=theta, beta=beta, t=t)
                    outcome = monitor.verdict(t)
                    if outcome is not None:
                        if verbose:
                            print(outcome)
                        break
                if t >= self.max_iter:
                    break
                t += 1

            xi_new, eta_new, theta_new, beta_new = download()
        finally:
            if own_engine:
                eng.close()
        if freeze_genes:    # the reference hands back the very objects it was given
            eta_new, beta_new = eta, beta
        return (bp, dp, xi_new, eta_new, theta_new, beta_new, monitor.loss)

    def _fit_minibatch(self, X, bp, dp, xi, eta, theta, beta, monitor, freeze_genes, reinit, loss_function,
                       max_iter, check_freq, checkstep_function, verbose, batchsize,
                       beta_theta_simultaneous, device):
        """Minibatch CAVI (scHPF_.py:643-650, 688-704): each iteration updates a batch of cells
        first (theta.rate from the current beta), then the genes from that batch alone.

        The matrix goes to the device ONCE: a whole-matrix engine keeps it as plans (for the default
        all-cells loss) and as a row-sorted copy (DeviceCAVI.keep_rows); per iteration the batch engine
        -- `batchsize` cells, alive for the whole fit, eta/beta and their tables resident -- gathers the
        batch's rows from that copy and plans them by device passes (upload_rows: work proportional to
        the batch, nothing over PCIe but the row numbers), where the reference re-slices X on the host
        (X[batch_ix], :643-650).  The batch's xi/theta rows travel with it (a few MB) and come back
        after the step.  Only iteration 0 of a reinitialised fit slices on the host: its random
        responsibilities are drawn per nonzero of X_batch in the reference's order."""
        from .util import minibatch_ix_generator
        nfactors = self.nfactors
        a, ap, c, cp = self.a, self.ap, self.c, self.cp
        Xcsr = X.tocsr()
        batches = minibatch_ix_generator(X.shape[0], batchsize)
        dtype = np.dtype(self.dtype)
        default_loss = loss_function is None
        # the reference batches X.tocsr(), which sums duplicate entries, and scores X as given
        duplicates = Xcsr.nnz != X.data.shape[0]
        from contextlib import ExitStack
        with ExitStack() as stack:
            eng = stack.enter_context(DeviceCAVI(batchsize, X.shape[1], nfactors, dtype=dtype, device=device))
            eng.hint_transient()          # a new batch every iteration: plans built the cheapest way
            eng.set_hypers(a, c, bp, dp)
            eng.set_gamma("eta", eta.vi_shape, eta.vi_rate)
            eng.set_gamma("beta", beta.vi_shape, beta.vi_rate)
            # The whole matrix on the device: as a row-sorted copy the batches are gathered from, and (default loss)
            # as plans for the all-cells loss.  A matrix that does not fit HBM beside those falls back step by step:
            # without the row copy (batches sliced on the host and uploaded, 8 B/nnz less) -- and with a caller's own
            # loss function nothing of the whole matrix is needed on the device at all.
            Xsum = Xcsr.tocoo() if duplicates else X

            def whole_matrix_engine(M, keep):
                # None = "does not fit": only HIP's out-of-memory takes the fallback chain below; invalid input
                # or any other failure is the caller's to see
                e = None
                try:
                    e = DeviceCAVI(M.shape[0], M.shape[1], nfactors, dtype=dtype, device=device)
                    if keep:
                        e.keep_rows()
                    e.upload(M)
                except _lib.SchpfHipError as exc:
                    if e is not None:
                        e.close()
                    if not _lib.is_out_of_memory(exc):
                        raise
                    return None
                except BaseException:
                    if e is not None:
                        e.close()
                    raise
                return stack.enter_context(e)

            source = whole_matrix_engine(Xsum, True)
            resident = source is not None and source.upload_info()["rows"]   # False for host-built / gather plans
            whole = None
            if default_loss:
                whole = source if not duplicates else None
                if whole is None:
                    whole = whole_matrix_engine(X, False)
                if whole is None and source is not None and resident:
                    # no room for both: give up the resident rows, keep the loss
                    source.close()
                    source, resident = None, False
                    whole = whole_matrix_engine(X, False)
                if whole is None:
                    raise _lib.SchpfHipError("the count matrix does not fit this device's memory for the all-cells loss "
                                             "(%s); pass a loss_function evaluated elsewhere or use more devices"
                                             % _lib.load().schpf_last_error().decode("utf-8", "replace"))
                whole.set_hypers(a, c, bp, dp)
            if not resident and source is not None and source is not whole:
                source.close()      # its plans serve nothing then
                source = None

            def gene_side():      # eta.vi_shape is a constant of the model (:618); its rate and beta live on the device
                if freeze_genes:
                    return eta, beta
                return HPF_Gamma(eta.vi_shape, eng.get_gamma("eta")[1]), HPF_Gamma(*eng.get_gamma("beta"))

            for t in range(max_iter):
                batch_ix = next(batches)
                if (t == 0 and reinit) or not resident:
                    X_batch = Xcsr[batch_ix, :].tocoo()
                    eng.upload(X_batch)
                    if t == 0 and reinit:
                        random_phi = np.random.dirichlet(np.ones(nfactors), X_batch.data.shape[0])
                        eng.init_phi_host(X_batch.data[:, None] * random_phi)
                else:
                    eng.upload_rows(source, batch_ix)
                eng.set_gamma("xi", xi.vi_shape[batch_ix], xi.vi_rate[batch_ix])
                eng.set_gamma("theta", theta.vi_shape[batch_ix], theta.vi_rate[batch_ix])
                eng.step(freeze_genes=freeze_genes, simultaneous=beta_theta_simultaneous,
                         cells_first=not beta_theta_simultaneous)
                ths, thr = eng.get_gamma("theta")
                theta.vi_shape[batch_ix], theta.vi_rate[batch_ix] = ths, thr
                xi.vi_rate[batch_ix] = eng.get_gamma("xi")[1]
                if t % check_freq == 0:
                    eta, beta = gene_side()
                    if default_loss:
                        whole.set_gamma("theta", theta.vi_shape, theta.vi_rate)
                        whole.set_gamma("beta", beta.vi_shape, beta.vi_rate)
                        curr = whole.mean_negative_pois_llh()
                    else:
                        curr = loss_function(a=a, ap=ap, bp=bp, c=c, cp=cp, dp=dp, xi=xi, eta=eta, theta=theta,
                                             beta=beta)
                    curr, pct = monitor.record(curr)
                    if verbose:
                        print("[Iter. {0: >4}]  loss:{1:.6f}  pct:{2:.9f}".format(t, curr, pct))
                    if checkstep_function is not None:
                        checkstep_function(bp=bp, dp=dp, xi=xi, eta=eta, theta=theta, beta=beta, t=t)
                    outcome = monitor.verdict(t)
                    if outcome is not None:
                        if verbose:
                            print(outcome)
                        break
                if t >= self.max_iter:
                    break
            eta, beta = gene_side()
        return (bp, dp, xi, eta, theta, beta, monitor.loss)

    def _setup(self, X, freeze_genes=False, reinit=True, clip=True):
        """Empirical bp/dp and (re)initialised Gammas, draw order xi, theta, eta, beta
        (scHPF_.py:783-844)."""
        nfactors, (ncells, ngenes) = self.nfactors, X.shape
        a, ap, c, cp = self.a, self.ap, self.c, self.cp
        xi, eta, theta, beta = self.xi, self.eta, self.theta, self.beta
        bp, dp = self._get_empirical_hypers(X, freeze_genes, clip)

        make = HPF_Gamma.random_gamma_factory
        if reinit or xi is None:
            xi = make((ncells,), ap, bp, dtype=self.dtype)
        if reinit or theta is None:
            theta = make((ncells, nfactors), a, bp, dtype=self.dtype)
        if freeze_genes:
            if eta is None or beta is None:
                raise ValueError("To fit with frozen gene variational distributions "
                                 "(`freeze_genes`==True), eta and beta must be set to valid "
                                 "HPF_Gamma instances.")
        else:
            if reinit or eta is None:
                eta = make((ngenes,), cp, dp, dtype=self.dtype)
            if reinit or beta is None:
                beta = make((ngenes, nfactors), c, dp, dtype=self.dtype)
        return (bp, dp, xi, eta, theta, beta)

    def _get_empirical_hypers(self, X, freeze_genes=False, clip=True):
        """bp = ap * mean/var of the cell sums, dp = cp * mean/var of the gene sums, only where
        unset; dp is clipped to bp/1000 (scHPF_.py:847-879)."""
        bp, dp = self.bp, self.dp

        marginals = []

        def mean_over_var(axis):
            # X.sum(axis) of the reference, both axes in one threaded pass of the library
            # (schpf_coo_marginals): at 1e8 nonzeros SciPy's single-threaded sums were a sixth
            # of a whole fit.  Counts are integers, so the sums -- and bp/dp -- are the same bits.
            if not marginals:
                marginals.extend(hpf_hip.coo_marginals(X))
            sums = marginals[0] if axis == 1 else marginals[1]
            return np.mean(sums) / np.var(sums)

        if bp is None:
            bp = self.ap * mean_over_var(1)
        if dp is None:
            if freeze_genes:
                raise ValueError("dp is None and cannot be set when freeze_genes is True.")
            dp = self.cp * mean_over_var(0)
            if clip and bp > 1000 * dp:
                clipped = bp / 1000
                print("Clipping dp: was {} now {}".format(dp, clipped))
                dp = clipped
        return bp, dp

    def _initialize(self, X, freeze_genes=False):
        """Randomly initialise and store all distributions (scHPF_.py:882-892)."""
        (self.bp, self.dp, self.xi, self.eta, self.theta, self.beta) = self._setup(
            X, freeze_genes, reinit=True)


def load_model(file_name):
    """Load a joblib model file written by this package or by the reference (:895-909)."""
    return joblib.load(file_name)


def save_model(model, file_name):
    """Write the model as a joblib file readable by the reference (:912-925)."""
    joblib.dump(model, file_name)


def combine_across_cells(x, y, y_ixs):
    """Merge the cell-side distributions of two models sharing eta/beta (:928-965)."""
    assert x.dp == y.dp
    assert x.eta == y.eta
    assert x.beta == y.beta
    xy = deepcopy(x)
    if y.bp != x.bp:
        xy.bp = None
    xy.xi = x.xi.combine(y.xi, y_ixs)
    xy.theta = x.theta.combine(y.theta, y_ixs)
    return xy


# pickle under the reference's module path so model files are interchangeable
HPF_Gamma.__module__ = _REFERENCE_MODULE
scHPF.__module__ = _REFERENCE_MODULE
