"""Random restarts: mirror of the reference's run_trials / run_trials_pool
(/root/reference/schpf/scHPF_.py:968-1332) on top of the device engine.

Same arguments, defaults, printed messages and return values.  What differs: the count matrix
is uploaded (and its sweep plans built) ONCE per nfactors and reused by every restart, and the
default loss -- mean negative Poisson log-likelihood of the training matrix -- is evaluated on
the device instead of through a host callback.  `run_trials_pool` keeps its signature (`njobs`,
`max_threads` are accepted) but there is no process pool: restarts run back to back on the GPU,
which is where the time goes; give `devices=[0, 1, ...]` to spread restarts over several GPUs
(one host thread and one engine per device).
"""
from functools import partial
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import loss as ls
from .engine import DeviceCAVI
from .scHPF_ import scHPF

__all__ = ["run_trials", "run_trials_pool"]

_GENE_WARNING = ("WARNING: you are running scHPF with {} genes, which is more than the ~20k protein "
                 "coding genes in the human genome. We suggest running scHPF on protein-coding genes only.")


def _default_device():
    import os
    return int(os.environ.get("SCHPF_DEVICE", "0"))


def _loss_plumbing(X, nfactors, check_freq, vcells, vX, loss_function, single_process, device=None):
    """Which loss `fit` should use: None = the device's own training loss (the reference's
    default, mean_negative_pois_llh on X); otherwise a host callable as in the reference."""
    if vcells is not None:
        assert X.shape[1] == vcells.shape[1]
    if vX is not None:
        assert vX.shape == X.shape
    default = loss_function is None
    if default:
        loss_function = partial(ls.mean_negative_pois_llh, single_process=single_process)
    if vcells is not None:       # validation loss: project the held-out cells at every check
        proj_kwargs = dict(reinit=False, min_iter=1, max_iter=min(10, check_freq),
                           check_freq=check_freq + 1, verbose=False)
        return loss_function, ls.projection_loss_function(loss_function, vcells, nfactors,
                                                          proj_kwargs=proj_kwargs, device=device)
    if default and (vX is None or vX is X):
        return loss_function, None
    return loss_function, ls.loss_function_for_data(loss_function, X if vX is None else vX)


def run_trials(X, nfactors, ntrials=5, min_iter=30, max_iter=1000, check_freq=10, epsilon=0.001,
               better_than_n_ago=5, dtype=np.float64, verbose=True, vcells=None, vX=None,
               loss_function=None, model_kwargs={}, return_all=False, reproject=False,
               reproject_kwargs={}, batchsize=0, beta_theta_simultaneous=False, loss_smoothing=1,
               device=None):
    """Train `ntrials` randomly initialised models, return the one with the lowest final loss
    (and, with return_all, the others ordered by increasing loss)."""
    if not hasattr(X, "row"):
        X = X.tocoo()
    ncells, ngenes = X.shape
    if ngenes >= 20000:
        print(_GENE_WARNING.format(ngenes))
    device = _default_device() if device is None else device
    raw_loss, data_loss_function = _loss_plumbing(X, nfactors, check_freq, vcells, vX, loss_function, False, device)
    batched = batchsize is not None and 1 < batchsize <= ncells

    engine = None
    if not batched:     # one upload / plan build for all restarts
        engine = DeviceCAVI(ncells, ngenes, nfactors, dtype=dtype, device=device)
        engine.upload(X)
    try:
        best_loss, best_model, best_t = np.finfo(np.float64).max, None, None
        models, losses = [], []
        for t in range(ntrials):
            model = scHPF(nfactors=nfactors, min_iter=min_iter, max_iter=max_iter, check_freq=check_freq,
                          epsilon=epsilon, better_than_n_ago=better_than_n_ago, verbose=verbose,
                          dtype=dtype, **model_kwargs)
            checkstep_function = None
            if vcells is not None:
                def checkstep_function(**kwargs):
                    train = ls.loss_function_for_data(raw_loss, X)
                    print("\ttrain:", "{0:.6f}".format(train(**kwargs)))
            model.fit(X, loss_function=data_loss_function, checkstep_function=checkstep_function,
                      batchsize=batchsize, loss_smoothing=loss_smoothing,
                      beta_theta_simultaneous=beta_theta_simultaneous, device=device, engine=engine)
            if reproject:
                print("Reprojecting data...")
                reproject_kwargs["replace"] = True
                reproject_kwargs["reinit"] = False
                proj_loss = model.project(X, device=device, **reproject_kwargs)
                model.loss.append(proj_loss)
                loss = proj_loss[-1]
            else:
                loss = model.loss[-1]
            if loss < best_loss:
                best_model, best_loss, best_t = model, loss, t
                if verbose:
                    print("New best!")
            if return_all:
                models.append(model)
                losses.append(loss)
            if verbose:
                print("Trial {0} loss: {1:.6f}".format(t, loss))
                print("Best loss: {0:.6f} (trial {1})".format(best_loss, best_t))
    finally:
        if engine is not None:
            engine.close()
        if hasattr(data_loss_function, "close"):      # the held-out cells' engine (validation loss)
            data_loss_function.close()
    if return_all:
        order = np.argsort(losses)
        ordered = [models[i] for i in order]
        assert ordered[0] is best_model or ordered[0].loss[-1] == best_model.loss[-1]
        return best_model, ordered[1:]
    return best_model


def run_trials_pool(X, nfactors, ntrials=5, njobs=0, max_threads=None, min_iter=30, max_iter=1000,
                    check_freq=10, epsilon=0.001, better_than_n_ago=5, dtype=np.float64, verbose=True,
                    vcells=None, vX=None, loss_function=None, model_kwargs={}, return_all=False,
                    reproject=False, reproject_kwargs={}, batchsize=0, beta_theta_simultaneous=False,
                    loss_smoothing=1, devices=None):
    """`ntrials` restarts for every K in `nfactors` (int or list); per K the model with the lowest
    final loss.  Returns a list of best models (and, with return_all, a list of lists of the
    rejected ones ordered by increasing loss), like the reference."""
    if not hasattr(X, "row"):
        X = X.tocoo()
    if X.shape[1] >= 20000:
        print(_GENE_WARNING.format(X.shape[1]))
    if isinstance(nfactors, (int, np.integer)):
        nfactors = [int(nfactors)]
    devices = [_default_device()] if not devices else list(devices)
    batched = batchsize is not None and 1 < batchsize <= X.shape[0]

    def fit_all(K, device, count):
        _, dlf = _loss_plumbing(X, K, check_freq, vcells, vX, loss_function, True, device)
        engine = None
        if not batched:
            engine = DeviceCAVI(X.shape[0], X.shape[1], K, dtype=dtype, device=device)
            engine.upload(X)
        out = []
        try:
            for _ in range(count):
                model = scHPF(nfactors=K, min_iter=min_iter, max_iter=max_iter, check_freq=check_freq,
                              epsilon=epsilon, better_than_n_ago=better_than_n_ago, verbose=False,
                              dtype=dtype, **model_kwargs)
                model.fit(X, loss_function=dlf, checkstep_function=None, single_process=True,
                          batchsize=batchsize, loss_smoothing=loss_smoothing, device=device, engine=engine)
                if reproject:
                    reproject_kwargs["replace"] = True
                    model.loss.append(model.project(X, loss_function=dlf, device=device, **reproject_kwargs))
                out.append(model)
        finally:
            if engine is not None:
                engine.close()
            if hasattr(dlf, "close"):
                dlf.close()
        return out

    # restarts of one K are dealt to the devices in contiguous shares
    jobs = []
    for K in nfactors:
        share, extra = divmod(ntrials, len(devices))
        for d, device in enumerate(devices):
            count = share + (1 if d < extra else 0)
            if count:
                jobs.append((K, device, count))
    if len(devices) == 1:
        results = [fit_all(*job) for job in jobs]
    else:
        with ThreadPoolExecutor(max_workers=len(devices)) as pool:
            results = list(pool.map(lambda job: fit_all(*job), jobs))
    ordered_best, ordered_reject = [], []
    for K in nfactors:
        candidates = [m for job, res in zip(jobs, results) if job[0] == K for m in res]
        final = [m.loss[-1][-1] if reproject else m.loss[-1] for m in candidates]
        order = np.argsort(final)
        ordered_best.append(candidates[order[0]])
        ordered_reject.append([candidates[i] for i in order[1:]])
    if return_all:
        return ordered_best, ordered_reject
    return ordered_best
