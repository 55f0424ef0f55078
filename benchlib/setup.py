"""Engine set-up exactly as scHPF._setup does it (whole matrix, a rank's block of it, a per-rank draw), the
algorithmic byte count of an iteration, and the wall-clock-to-convergence fits."""
import time

import numpy as np

from .data import planted_block


def convergence_run(N, G, K, dtype, density):
    """Wall-clock of a whole scHPF.fit() under the reference's default stop rule (min 30 / max
    1000 iterations, loss every 10, epsilon 0.001 %; scHPF_.py:234-238, 750-761) on planted data,
    host COO in, fitted model out: upload + plan build + iterations + loss checks + download."""
    from schpf import scHPF
    t_gen = time.perf_counter()
    X = planted_block(N, G, K, target_events=int(N * G * density * 1.6), seed=42)
    t_gen = time.perf_counter() - t_gen
    # the number of iterations the stop rule takes depends on the random start: three seeds, each a
    # complete fit from the host matrix; the headline is the median wall-clock
    runs = []
    for seed in (0, 1, 2):
        np.random.seed(seed)
        model = scHPF(K, dtype=dtype, verbose=False)
        t0 = time.perf_counter()
        model.fit(X, init="device")
        wall = time.perf_counter() - t0
        checks = len(model.loss)
        runs.append({"seed": seed, "fit_wall_s": wall, "loss_checks": checks,
                     "iterations": (checks - 1) * model.check_freq + 1,
                     "first_loss": float(model.loss[0]), "final_loss": float(model.loss[-1])})
    med = sorted(runs, key=lambda r: r["fit_wall_s"])[1]
    return {"data": "planted Gamma-Poisson, %d x %d, nnz %d (density %.4f), max count %d"
                    % (N, G, X.nnz, X.nnz / float(N) / G, int(X.data.max())),
            "what": "median over 3 random starts of the wall-clock of scHPF.fit(X) -- host COO in, fitted "
                    "model out: validation, H2D, plan build, t=0 responsibilities, every iteration and loss "
                    "check, download -- under the reference's default stop rule (min_iter 30, max_iter "
                    "1000, check_freq 10, epsilon 0.001 %, scHPF_.py:234-238, 750-761)",
            "unit": "s", "nnz": int(X.nnz), "data_generation_s": t_gen,
            "fit_wall_s": med["fit_wall_s"], "loss_checks": med["loss_checks"], "iterations": med["iterations"],
            "first_loss": med["first_loss"], "final_loss": med["final_loss"], "runs": runs}


def algorithmic_bytes(nnz, N, G, K, itemsize):
    """B_iter = 12*nnz + 4*K*s*(N+G) + 2*s*(N+G)."""
    return 12 * nnz + 4 * K * itemsize * (N + G) + 2 * itemsize * (N + G)


def init_engine(eng, X, K, dtype, seed=0, whole=None, rows=None):
    """Random init exactly as scHPF._setup (reference scHPF_.py:783-844), hypers empirical.  A rank of a
    sharded run passes the WHOLE matrix as `whole` and its row range as `rows`: hyperparameters and the
    random start are those of the unsharded fit (what scHPF.fit(X, devices=[...]) does), the rank uploads
    its block `X` and its slices of xi / theta."""
    from schpf import scHPF
    np.random.seed(seed)
    m = scHPF(K, dtype=dtype)
    bp, dp, xi, eta, theta, beta = m._setup(X if whole is None else whole, freeze_genes=False, reinit=True)
    xi.vi_shape[:] = m.ap + K * m.a
    eta.vi_shape[:] = m.cp + K * m.c
    eng.upload(X)
    eng.set_hypers(m.a, m.c, bp, dp)
    sl = slice(None) if rows is None else slice(int(rows[0]), int(rows[1]))
    eng.set_gamma("xi", xi.vi_shape[sl], xi.vi_rate[sl])
    eng.set_gamma("theta", theta.vi_shape[sl], theta.vi_rate[sl])
    eng.set_gamma("eta", eta.vi_shape, eta.vi_rate)
    eng.set_gamma("beta", beta.vi_shape, beta.vi_rate)
    return bp, dp, (xi, eta, theta, beta)


def init_engine_of_rank(eng, X, K, dtype, row_sums, col_sums, rank, seed=0):
    """init_engine for a rank that holds ONLY its row block (the per-rank draw of C5): the empirical hyperparameters
    (reference scHPF_.py:847-879) from the all-reduced marginals of the whole matrix -- bp = ap mean / var of the cell
    sums, dp = cp mean / var of the gene sums, clipped to bp / 1000 --, eta / beta drawn from one stream on every
    rank (identical replicas), the rank's xi / theta from a stream of its own.  The same distributions as
    scHPF._setup's (:49-70, :783-844); not the unsharded fit's draws, which would take drawing all N x K of them on
    every rank."""
    from schpf import scHPF
    from schpf.scHPF_ import HPF_Gamma
    m = scHPF(K, dtype=dtype)
    bp = m.ap * np.mean(row_sums) / np.var(row_sums)
    dp = m.cp * np.mean(col_sums) / np.var(col_sums)
    if bp > 1000 * dp:
        dp = bp / 1000
    make = HPF_Gamma.random_gamma_factory
    np.random.seed(seed)
    eta = make((X.shape[1],), m.cp, dp, dtype=dtype)
    beta = make((X.shape[1], K), m.c, dp, dtype=dtype)
    np.random.seed(seed + 1 + rank)
    xi = make((X.shape[0],), m.ap, bp, dtype=dtype)
    theta = make((X.shape[0], K), m.a, bp, dtype=dtype)
    xi.vi_shape[:] = m.ap + K * m.a
    eta.vi_shape[:] = m.cp + K * m.c
    eng.upload(X)
    eng.set_hypers(m.a, m.c, bp, dp)
    eng.set_gamma("xi", xi.vi_shape, xi.vi_rate)
    eng.set_gamma("theta", theta.vi_shape, theta.vi_rate)
    eng.set_gamma("eta", eta.vi_shape, eta.vi_rate)
    eng.set_gamma("beta", beta.vi_shape, beta.vi_rate)
    return bp, dp
