#!/usr/bin/env python
"""Benchmark of the scHPF CAVI hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c2|c5-shard] [--dtype f64|f32]

A "step" is one CAVI iteration (the loop body of the reference's scHPF._fit,
schpf/scHPF_.py:657-714) over the whole synthetic count matrix.  The default workload is
BASELINE.json's headline configuration C3: 100k cells x 20k genes, ~5 % nonzeros, K = 20,
float64 (the reference's default dtype), inputs resident in HBM before timing starts.

For N > 1 (launched by torch.distributed.run, one rank per GPU) the SAME global shape is
row-sharded over the ranks (strong scaling, BASELINE.json configs[3]): rank r draws its own
block of N/P cells, and every iteration does one RCCL all-reduce of the G*K + K gene-side
sums.

Prints ONE JSON line on rank 0, carrying the driver's contract fields plus
  roofline     : dominant kernel (the sweep) against the HBM roof, HIP-event timed
  cpu_baseline : the CPU oracle in the reference's execution shape on this box's cores
"""
import argparse
import json
import os
import sys
import time

import numpy as np
from scipy.sparse import coo_matrix

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (ncells, ngenes, density, K)
    "c2": (10_000, 5_000, 0.03, 10),
    "c3": (100_000, 20_000, 0.05, 20),
    "c5-shard": (125_000, 25_000, 0.02, 50),     # one GPU's 1/8 share of C5 (1M x 25k)
    "c4-shard": (12_500, 20_000, 0.05, 20),      # one GPU's 1/8 share of C3/C4 (what a rank of --gpus 8 holds)
}
HBM_PEAK_GBS = 8000.0   # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)


def synthetic_block(ncells, ngenes, density, seed):
    """Generator A of SURVEY.md 8(d) = the reference's test-fixture recipe
    (tests/conftest.py:14-25): negative-binomial counts at uniform positions, dups summed."""
    rng = np.random.RandomState(seed)
    nnz = int(round(ncells * ngenes * density))
    x = rng.negative_binomial(2, 0.5, nnz).astype(np.int32)
    x[x == 0] = 1
    row = rng.randint(0, ncells, nnz).astype(np.int32)
    col = rng.randint(0, ngenes, nnz).astype(np.int32)
    X = coo_matrix((x, (row, col)), shape=(ncells, ngenes), dtype=np.int32)
    X.sum_duplicates()
    return X


def planted_block(ncells, ngenes, K, target_events, seed):
    """Generator B of SURVEY.md 8(d): counts from a planted Gamma-Poisson factor model, so that
    the reference's stop rule has something to converge to.  x_ig ~ Poisson(sum_k theta_ik
    beta_gk) is sampled factor by factor: the events of factor k are Poisson(S_theta_k *
    S_beta_k) many, each landing on cell i with probability theta_ik / S_theta_k and gene g with
    probability beta_gk / S_beta_k (independent because the rate factorises)."""
    rng = np.random.RandomState(seed)
    theta = rng.gamma(0.3, 1.0, (ncells, K)) * rng.gamma(2.0, 0.5, (ncells, 1))
    beta = rng.gamma(0.3, 1.0, (ngenes, K)) * rng.gamma(2.0, 0.5, (ngenes, 1))
    st, sb = theta.sum(0), beta.sum(0)
    scale = target_events / float((st * sb).sum())
    rows, cols = [], []
    for k in range(K):
        n_k = rng.poisson(st[k] * sb[k] * scale)
        rows.append(np.searchsorted(np.cumsum(theta[:, k]) / st[k], rng.random_sample(n_k)).astype(np.int32))
        cols.append(np.searchsorted(np.cumsum(beta[:, k]) / sb[k], rng.random_sample(n_k)).astype(np.int32))
    row = np.minimum(np.concatenate(rows), ncells - 1)
    col = np.minimum(np.concatenate(cols), ngenes - 1)
    X = coo_matrix((np.ones(row.shape[0], np.int32), (row, col)), shape=(ncells, ngenes), dtype=np.int32)
    X.sum_duplicates()
    return X


def convergence_run(N, G, K, dtype, density):
    """Wall-clock of a whole scHPF.fit() under the reference's default stop rule (min 30 / max
    1000 iterations, loss every 10, epsilon 0.001 %; scHPF_.py:234-238, 750-761) on planted data,
    host COO in, fitted model out: upload + plan build + iterations + loss checks + download."""
    from schpf import scHPF
    X = planted_block(N, G, K, target_events=int(N * G * density * 1.6), seed=42)
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
            "fit_wall_s": med["fit_wall_s"], "loss_checks": med["loss_checks"], "iterations": med["iterations"],
            "first_loss": med["first_loss"], "final_loss": med["final_loss"], "runs": runs}


def algorithmic_bytes(nnz, N, G, K, itemsize):
    """B_iter = 12*nnz + 4*K*s*(N+G) + 2*s*(N+G)."""
    return 12 * nnz + 4 * K * itemsize * (N + G) + 2 * itemsize * (N + G)


def init_engine(eng, X, K, dtype, seed=0):
    """Random init exactly as scHPF._setup (reference scHPF_.py:783-844), hypers empirical."""
    from schpf import scHPF
    np.random.seed(seed)
    m = scHPF(K, dtype=dtype)
    bp, dp, xi, eta, theta, beta = m._setup(X, freeze_genes=False, reinit=True)
    xi.vi_shape[:] = m.ap + K * m.a
    eta.vi_shape[:] = m.cp + K * m.c
    eng.upload(X)
    eng.set_hypers(m.a, m.c, bp, dp)
    for name, g in (("xi", xi), ("theta", theta), ("eta", eta), ("beta", beta)):
        eng.set_gamma(name, g.vi_shape, g.vi_rate)
    return bp, dp, (xi, eta, theta, beta)


def cpu_baseline(X, K, dtype, budget_s=20.0):
    """The CPU oracle (oracle/cavi_oracle.c: the reference's numba execution shape --
    thread-parallel Xphi + llh, SERIAL scatter-adds, hpf_numba.py:24,54 vs :128,159) timed
    on a bounded row-subsample of the same matrix, scaled to whole-matrix iterations/s."""
    from oracle import hpf_oracle as orc
    orc.build()
    cores = os.cpu_count() or 1
    N, G = X.shape
    nnz_full = X.nnz
    # ~2e9 nnz*K element-ops per ~10 s of this code on one socket: keep nnz*K <= 2.5e8
    target_nnz = min(nnz_full, int(2.5e8 / K))
    rows = max(1, int(N * target_nnz / max(nnz_full, 1)))
    keep = X.row < rows
    Xs = coo_matrix((X.data[keep], (X.row[keep], X.col[keep])), shape=(rows, G))
    np.random.seed(0)
    bp, dp, st = orc.setup_state(Xs, K, np.dtype(dtype), 0.3, 1.0, 0.3, 1.0)
    st.xi_shape[:] = 1.0 + K * 0.3
    st.eta_shape[:] = 1.0 + K * 0.3
    x, row, col = Xs.data, Xs.row, Xs.col
    orc.cavi_iteration(x, row, col, st, 0.3, 0.3, bp, dp, nthreads=cores)   # warm-up
    t0 = time.perf_counter()
    iters = 0
    while True:
        orc.cavi_iteration(x, row, col, st, 0.3, 0.3, bp, dp, nthreads=cores)
        iters += 1
        if time.perf_counter() - t0 > budget_s or iters >= 5:
            break
    dt = (time.perf_counter() - t0) / iters
    scale = Xs.nnz / float(nnz_full)
    return {
        "value": (1.0 / dt) * scale, "unit": "iterations/s", "cores": cores, "kind": "port",
        "sample": "first %d of %d cells (nnz %d of %d), %d timed iterations of the reference-"
                  "structure C oracle (parallel Xphi, serial scatter-adds), %.3f s/iter on the "
                  "sample, scaled by nnz to the full matrix" % (rows, N, Xs.nnz, nnz_full, iters, dt),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-sharded", action="store_true",
                    help="use the sharded driver (process group, exchange all-reduce) even with one rank")
    ap.add_argument("--converge", action="store_true",
                    help="also time a whole fit() to convergence on planted data (N=1 only, adds minutes)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py "
                             "--gpus %d ..." % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    sharded = world > 1 or args.force_sharded
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    from schpf_amd import DeviceCAVI
    from schpf_amd.sharded import ShardedCAVI, exchange_tensor_of

    N, G, density, K = CONFIGS[args.config]
    dtype = np.float64 if args.dtype == "f64" else np.float32
    itemsize = np.dtype(dtype).itemsize
    n_local = N // world + (1 if rank < N % world else 0)
    X = synthetic_block(n_local, G, density, seed=42 + 1000 * rank)

    # the engine enqueues on a stream of its own; ShardedCAVI orders the collective with it
    eng = DeviceCAVI(n_local, G, K, dtype=dtype, device=local_rank)
    t_up = time.perf_counter()
    init_engine(eng, X, K, dtype)
    upload_s = time.perf_counter() - t_up
    nnz_local = X.nnz
    if sharded:
        drv = ShardedCAVI(eng, exchange_tensor_of(eng, local_rank))
        step = drv.step
        nnz_t = torch.tensor([nnz_local], dtype=torch.int64, device="cuda")
        dist.all_reduce(nnz_t)
        nnz_total = int(nnz_t.item())
        loss_fn = drv.mean_negative_pois_llh
    else:
        step = eng.step
        nnz_total = nnz_local
        loss_fn = eng.mean_negative_pois_llh

    def fence():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    eng.init_phi_device(12345)          # t = 0 responsibilities (device generator)
    for _ in range(args.warmup):
        step()
    loss_start = loss_fn()
    eng.profile(True)
    eng.profile_read()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile(False)
    if sharded:
        el = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())

    # loss evaluation (every check_freq = 10 iterations in a default fit), outside the timed region
    fence()
    t1 = time.perf_counter()
    loss_end = loss_fn()
    torch.cuda.synchronize()
    loss_ms = (time.perf_counter() - t1) * 1e3

    ms_per_step = elapsed / args.steps * 1e3
    value = args.steps / elapsed
    b_iter = algorithmic_bytes(nnz_local, n_local, G, K, itemsize)
    sweeps = prof["cell_sweep"]["launches"] + prof["gene_sweep"]["launches"]
    sweep_ms = (prof["cell_sweep"]["ms"] + prof["gene_sweep"]["ms"]) / max(sweeps, 1)
    # launches per iteration: 2 (cell-side + gene-side) or 1 (both sides in one dual launch)
    per_iter = max(1, int(round(sweeps / float(args.steps))))
    b_launch = b_iter / per_iter
    achieved = b_launch / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0
    info = eng.plan_info()
    # HBM bytes per launch from the committed rocprofv3 PMC passes (collected separately, as PMC
    # must be; profiles/r01/pmc_traffic.json), only for the configuration they were taken on
    traffic, traffic_src = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")) as fh:
            rec = json.load(fh).get("%s/%s" % (args.config, args.dtype))
        if rec and world == 1:
            traffic = rec["bytes_per_launch"] / 1e9
            traffic_src = "profiles/r01/pmc_traffic.json: (2*FETCH_SIZE + WRITE_SIZE) per launch, GB"
    except (OSError, ValueError, KeyError):
        pass

    out = {
        "metric": "CAVI iterations/sec, 100kx20k K=20" if args.config == "c3"
                  else "CAVI iterations/sec (%s)" % args.config,
        "value": value, "unit": "iterations/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {
            "workload": "%s: synthetic %d cells x %d genes, density %.3f (negative-binomial counts, "
                        "RandomState(42+1000*rank) per row block), nnz %d after summing duplicates, "
                        "K=%d, one CAVI iteration per step (no loss evaluation inside the step)"
                        % (args.config.upper(), N, G, density, nnz_total, K),
            "parallelism": "cells row-sharded x%d, one RCCL all-reduce of G*K+K per iteration" % world
                           if world > 1 else "single GPU",
            "plan": info,
        },
        "roofline": {
            "bound": "hbm",
            "kernel": ("tile_sweep_dual_kernel (cell-side + gene-side sweep in one launch; algorithmic "
                       "bytes per launch = B_iter" if per_iter == 1 else
                       "tile_sweep_kernel (cell + gene launches; algorithmic bytes per launch = B_iter/2")
                      + ", B_iter = 12*nnz + 4*K*s*(N+G) + 2*s*(N+G))",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_gb_per_launch": b_launch / 1e9, "sweep_launches_per_iteration": per_iter,
            "avg_launch_ms": sweep_ms, "launches": sweeps,
            "cell_sweep_ms": prof["cell_sweep"]["ms"] / max(prof["cell_sweep"]["launches"], 1),
            "gene_sweep_ms": prof["gene_sweep"]["ms"] / max(prof["gene_sweep"]["launches"], 1),
            "gamma_updates_ms": prof["gamma_updates"]["ms"] / max(prof["gamma_updates"]["launches"], 1),
            "iteration_frac_of_hbm_peak": (b_iter / (ms_per_step * 1e-3) / 1e9) / HBM_PEAK_GBS,
        },
        "loss_eval_ms": loss_ms, "loss_after_warmup": loss_start, "loss_after_steps": loss_end,
        "iterations_per_s_with_loss_every_10": 10.0 / (10 * ms_per_step * 1e-3 + loss_ms * 1e-3),
        "upload_and_plan_s": upload_s,
    }
    if rank == 0 and world == 1 and args.converge:
        eng.close()
        out["convergence"] = convergence_run(N, G, K, dtype, density)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(X, K, dtype)
    elif rank == 0:
        out["cpu_baseline"] = None
    eng.close()
    if sharded:
        dist.barrier()
        dist.destroy_process_group()
    # RCCL prints its version banner through C stdio, which is flushed at exit when stdout is a
    # pipe: flush it now so that the JSON line is the LAST line of stdout
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if rank == 0:
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
