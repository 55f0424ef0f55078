"""Loss functions, evaluated on the GPU.

Mirror of /root/reference/schpf/loss.py: `pois_llh_pointwise` (:107-139) and
`mean_negative_pois_llh` (:142-168) plus the two higher-order helpers
`loss_function_for_data` (:17-34) and `projection_loss_function` (:37-102).
As in the reference every loss takes the data positionally/as `X` and everything
else as keyword arguments, ignoring the ones it does not use.
"""
import functools

import numpy as np

from .hpf_hip import compute_pois_llh

__all__ = ["loss_function_for_data", "projection_loss_function", "pois_llh_pointwise",
           "mean_negative_pois_llh"]


def loss_function_for_data(loss_function, X):
    """Bind the data argument `X` of a loss function."""
    return functools.partial(loss_function, X=X)


def projection_loss_function(loss_function, X, nfactors, model_kwargs={}, proj_kwargs={}, device=None):
    """Loss of held-out cells `X` after projecting them onto the model being trained.

    `device` (an addition to the reference's signature): HIP device the held-out cells live on;
    default $SCHPF_DEVICE or 0, like project().  The returned function has a `.close()` that
    releases the engine it keeps between checks.

    Returns f(*, a, ap, bp, c, cp, dp, eta, beta, **ignored): it copies the
    hyperparameters and gene distributions into a private scHPF, runs project(X,
    replace=True) (defaults reinit=False, max_iter=min_iter=10, no intermediate
    loss checks) and evaluates `loss_function` on the projection.
    """
    from .scHPF_ import scHPF   # late import: scHPF_ imports this module
    from .engine import DeviceCAVI

    pmodel = scHPF(nfactors=nfactors, **model_kwargs)
    if not hasattr(X, "row"):
        X = X.tocoo()
    import os
    if device is None:
        device = int(os.environ.get("SCHPF_DEVICE", "0"))
    held = {}    # the held-out cells stay on the device between checks: one upload, one pair of plans

    def _projection_loss_function(*, a, ap, bp, c, cp, dp, eta, beta, **kwargs):
        assert eta.dims[0] == beta.dims[0]
        assert beta.dims[1] == nfactors
        pmodel.a, pmodel.ap, pmodel.bp = a, ap, bp
        pmodel.c, pmodel.cp, pmodel.dp = c, cp, dp
        pmodel.eta, pmodel.beta = eta, beta

        proj_kwargs.setdefault("reinit", False)
        proj_kwargs.setdefault("max_iter", 10)
        proj_kwargs.setdefault("min_iter", 10)
        proj_kwargs.setdefault("check_freq", proj_kwargs["max_iter"] + 1)
        dtype = np.dtype(pmodel.dtype)
        eng = held.get("engine")
        if eng is None or eng.dtype != dtype or "engine" in proj_kwargs:
            eng = proj_kwargs.get("engine")
            if eng is None:
                old = held.pop("engine", None)
                if old is not None:
                    old.close()
                eng = DeviceCAVI(X.shape[0], X.shape[1], nfactors, dtype=dtype, device=device)
                eng.upload(X)
                held["engine"] = eng
        pmodel.project(X, replace=True, **dict(proj_kwargs, engine=eng, device=device))

        if getattr(loss_function, "func", loss_function) is mean_negative_pois_llh:   # also a functools.partial of it
            # the engine holds exactly the state project() just returned: evaluate there (one scalar back)
            return eng.mean_negative_pois_llh()
        return loss_function(X, a=pmodel.a, ap=pmodel.ap, bp=pmodel.bp, c=pmodel.c, cp=pmodel.cp,
                             dp=pmodel.dp, xi=pmodel.xi, eta=pmodel.eta, theta=pmodel.theta,
                             beta=pmodel.beta)

    def close():
        eng = held.pop("engine", None)
        if eng is not None:
            eng.close()
    _projection_loss_function.close = close
    return _projection_loss_function


def pois_llh_pointwise(X, *, theta, beta, single_process=False, **kwargs):
    """Poisson log-likelihood of each stored nonzero of X, in X's COO order.

    `single_process` is accepted for signature compatibility and ignored: there is
    one execution path, the GPU.
    """
    return compute_pois_llh(X.data, X.row, X.col, theta.vi_shape, theta.vi_rate,
                            beta.vi_shape, beta.vi_rate)


def mean_negative_pois_llh(X, *, theta, beta, single_process=False, **kwargs):
    """Mean over the stored nonzeros of X of the negative Poisson log-likelihood."""
    return np.mean(-pois_llh_pointwise(X=X, theta=theta, beta=beta))
