#!/usr/bin/env python
"""Benchmark of the scHPF CAVI hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c2|c5-shard] [--dtype f64|f32]

A "step" is one CAVI iteration (the loop body of the reference's scHPF._fit,
schpf/scHPF_.py:657-714) over the whole synthetic count matrix.  The default workload is
BASELINE.json's headline configuration C3: 100k cells x 20k genes, ~5 % nonzeros, K = 20,
float64 (the reference's default dtype), inputs resident in HBM before timing starts.

For N > 1 (one rank per GPU: launched by torch.distributed.run, or -- when `--gpus N` is given
without a launcher -- by this script re-executing itself under it) the SAME matrix (seed 42) is
row-sharded over the ranks by the product's own nnz-balanced partition
(schpf_amd.sharded.row_partition; strong scaling, BASELINE.json configs[3]): rank r keeps rows
[bounds[r], bounds[r+1]), and every iteration does one RCCL all-reduce of the G*K + K
gene-side sums.

Prints ONE JSON line on rank 0, carrying the driver's contract fields plus
  roofline     : dominant kernel (the sweep), HIP-event timed, against THREE roofs -- algorithmic bytes vs HBM, the
                 essential FMAs vs the vector-FP peak of the dtype, the LDS bytes vs 256 B/clk/CU -- with the shader
                 clock measured under the kernel (sclk_mhz); `bound` names the nearest one
  cpu_baseline : the CPU oracle in the reference's execution shape on this box's cores
"""
import argparse
import json
import os
import re
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
    "c5": (1_000_000, 25_000, 0.02, 50),         # all of C5 on ONE GPU (nnz ~5e8: minutes of host-side generation)
    "c5-small": (50_000, 25_000, 0.02, 50),      # C5 at 1/20 of its cells, drawn slab by slab like C5 (tests of the per-rank draw)
    "c4-shard": (12_500, 20_000, 0.05, 20),      # one GPU's 1/8 share of C3/C4 (what a rank of --gpus 8 holds)
    "c4-shard2": (50_000, 20_000, 0.05, 20),     # ... of --gpus 2
    "c4-shard4": (25_000, 20_000, 0.05, 20),     # ... of --gpus 4
}
# the row shards of C3 a rank of `--gpus N` holds, as configurations of their own for one-GPU boxes: (whole, ranks)
SHARD_OF = {"c4-shard": ("c3", 8), "c4-shard4": ("c3", 4), "c4-shard2": ("c3", 2), "c5-shard": ("c5", 8)}
HBM_PEAK_GBS = 8000.0   # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
# vector (VALU) FMA peaks: FP32 157.3 TFLOP/s (MI355X_MICROARCH.md, chip-level table); FP64 vector is half of it,
# 78.6 TFLOP/s (public MI355X spec; 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz) -- the sweep has no MFMA work
VALU_PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3}
MAX_CLOCK_MHZ = 2400            # MI355X_MICROARCH.md chip table: the clock the peaks above are quoted at
LDS_BYTES_PER_CLK_PER_CU = 256  # same guide, LDS section: 64 dwords wide per clock


from benchlib.data import (SLAB_CONFIGS, SLAB_ROWS, planted_block, synthetic_block, synthetic_slabs,   # noqa: E402,F401
                           synthetic_slabs_of_rank)
from benchlib.setup import algorithmic_bytes, convergence_run, init_engine, init_engine_of_rank   # noqa: E402,F401
from benchlib.traffic import _pmc_child, live_traffic, static_traffic   # noqa: E402,F401


def _oracle_state(orc, X, K, dtype):
    np.random.seed(0)
    bp, dp, st = orc.setup_state(X, K, np.dtype(dtype), 0.3, 1.0, 0.3, 1.0)
    st.xi_shape[:] = 1.0 + K * 0.3
    st.eta_shape[:] = 1.0 + K * 0.3
    return bp, dp, st


def _time_iterations(fn, budget_s, max_iters):
    fn()                                            # warm-up (page faults, thread pool)
    t0 = time.perf_counter()
    iters = 0
    while True:
        fn()
        iters += 1
        if time.perf_counter() - t0 > budget_s or iters >= max_iters:
            break
    return (time.perf_counter() - t0) / iters, iters


def host_cpus():
    """What this process can actually run on: logical CPUs, the affinity mask, and the cgroup CPU quota
    (cgroup v2 cpu.max / v1 cfs_quota).  A box may show 256 logical CPUs to a container that is allowed
    16 CPU-seconds per second: threads beyond the quota only add throttling."""
    import math
    info = {"logical": os.cpu_count() or 1, "affinity": len(os.sched_getaffinity(0)), "cgroup_cpu_max": None}
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max":
            info["cgroup_cpu_max"] = float(quota) / float(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, per = float(fq.read()), float(fp.read())
            if q > 0:
                info["cgroup_cpu_max"] = q / per
        except (OSError, ValueError):
            pass
    usable = info["affinity"]
    if info["cgroup_cpu_max"]:
        usable = max(1, min(usable, int(math.ceil(info["cgroup_cpu_max"]))))
    info["usable"] = usable
    return info


def _mem_available_gb():
    try:
        with open("/proc/meminfo") as fh:
            for line in fh:
                if line.startswith("MemAvailable:"):
                    return float(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def _team_sizes(host):
    """Candidate thread counts, largest first: the usable cores and its halvings down to 4, plus twice
    the usable cores when the affinity mask allows (SMT siblings) -- never the raw logical count of a
    quota-limited container."""
    sizes, n = [], host["usable"]
    if 2 * n <= host["affinity"]:
        sizes.append(2 * n)
    while n >= 4:
        sizes.append(n)
        n //= 2
    return sizes or [max(1, host["usable"])]


def cpu_baseline(X, K, dtype):
    """Both CPU comparators of SURVEY.md 8(d), timed on this box's host cores.

    (i)  "numba-structure" (oracle/cavi_oracle_impl.h): the reference's execution shape --
         thread-parallel Xphi (nnz x K materialised, K exp per nonzero) and llh, SERIAL
         scatter-adds and rate updates (hpf_numba.py:24,54 parallel; :128,159 serial).  This is
         the stand-in for "the reference numba CPU path" (numba itself is not installable here) and
         is what `value` reports.  Its team size is chosen by measurement on a 1/16 row sample
         (like (ii)'s), then the WHOLE matrix is iterated when the host has the memory for the
         materialised Xphi (nnz * K * itemsize, 16 GB at C3 f64, + the matrix): one untimed
         iteration, then timed ones.  Only a host without that memory gets the two-point
         extrapolation from 1/8 and 1/4 of the cells (and the line says so).
    (ii) "fused OpenMP" (oracle/cavi_fused_impl.h): exp hoisted, no Xphi, parallel CSR + CSC
         passes, AVX2 -- the best CPU form this build knows, on the WHOLE matrix, so that the
         GPU/CPU ratio is not flattered by the reference's serial scatter.

    `cores` is what the box lets this process use (affinity mask capped by the cgroup CPU quota), `threads`
    the team size the reported figure actually ran with; `host` has the raw facts (logical CPUs, affinity,
    cgroup quota)."""
    from oracle import hpf_oracle as orc
    orc.build()
    host = host_cpus()
    N, G = X.shape
    nnz_full = X.nnz
    itemsize = np.dtype(dtype).itemsize

    def sample(target_elems):
        target_nnz = min(nnz_full, int(target_elems / K))
        rows = max(1, int(N * target_nnz / max(nnz_full, 1)))
        keep = X.row < rows
        return coo_matrix((X.data[keep], (X.row[keep], X.col[keep])), shape=(rows, G)), rows

    # team size of (i), by measurement on ~1/16 of C3 (nnz * K = 1.25e8 elements)
    Xs, _ = sample(1.25e8)
    bp, dp, st = _oracle_state(orc, Xs, K, dtype)
    trial_i = {}
    for n in _team_sizes(host):
        orc.cavi_iteration(Xs.data, Xs.row, Xs.col, st, 0.3, 0.3, bp, dp, nthreads=n)      # warm-up of this team
        t0 = time.perf_counter()
        orc.cavi_iteration(Xs.data, Xs.row, Xs.col, st, 0.3, 0.3, bp, dp, nthreads=n)
        trial_i[n] = time.perf_counter() - t0
    threads_i = min(trial_i, key=trial_i.get)
    trial_i_txt = ", ".join("%d: %.3f" % (k, v) for k, v in sorted(trial_i.items(), reverse=True))

    need_gb = (nnz_full * K * itemsize + nnz_full * (12 + itemsize) + 4.0 * (N + G) * K * itemsize) / 1e9
    whole = _mem_available_gb() >= need_gb * 1.25 + 4.0
    points = []
    if whole:
        bp, dp, st = _oracle_state(orc, X, K, dtype)
        x, row, col = X.data, X.row, X.col
        dt, iters = _time_iterations(
            lambda: orc.cavi_iteration(x, row, col, st, 0.3, 0.3, bp, dp, nthreads=threads_i), 8.0, 3)
        full_s, spread = dt, 0.0
        points.append({"cells": N, "nnz": int(nnz_full), "s_per_iter": dt, "iterations": iters,
                       "scaled_by_nnz_it_per_s": 1.0 / dt})
        how = ("the WHOLE matrix (nnz %d; Xphi of %.1f GB materialised every iteration like the reference's), one "
               "untimed and %d timed iterations, %.3f s/iter" % (nnz_full, nnz_full * K * itemsize / 1e9, iters, dt))
    else:
        for frac_target in (2.5e8, 5.0e8):             # nnz*K element budget of a sample (1/8 and 1/4 of C3)
            Xs, rows = sample(frac_target)
            bp, dp, st = _oracle_state(orc, Xs, K, dtype)
            x, row, col = Xs.data, Xs.row, Xs.col
            dt, iters = _time_iterations(
                lambda: orc.cavi_iteration(x, row, col, st, 0.3, 0.3, bp, dp, nthreads=threads_i), 6.0, 3)
            points.append({"cells": rows, "nnz": int(Xs.nnz), "s_per_iter": dt, "iterations": iters,
                           "scaled_by_nnz_it_per_s": (1.0 / dt) * Xs.nnz / float(nnz_full)})
            if Xs.nnz == nnz_full:
                break
        if len(points) == 2 and points[1]["nnz"] > points[0]["nnz"]:
            alpha = (points[1]["s_per_iter"] - points[0]["s_per_iter"]) / float(points[1]["nnz"] - points[0]["nnz"])
            gamma = points[1]["s_per_iter"] - alpha * points[1]["nnz"]
            if alpha <= 0:                              # timing noise: fall back to plain scaling
                alpha, gamma = points[1]["s_per_iter"] / points[1]["nnz"], 0.0
            full_s = alpha * nnz_full + max(gamma, 0.0)
        else:
            full_s = points[-1]["s_per_iter"] * nnz_full / float(points[-1]["nnz"])
        scaled = [p["scaled_by_nnz_it_per_s"] for p in points]
        spread = (max(scaled + [1.0 / full_s]) - min(scaled + [1.0 / full_s])) * full_s
        how = ("EXTRAPOLATED (host has %.0f GB available, the whole matrix needs %.0f): timed on the first %d and %d "
               "of %d cells (nnz %d and %d of %d; %.3f and %.3f s/iter), linear fit in nnz to %.2f s/iter; plain "
               "nnz-scaling of the two samples gives %.4f and %.4f it/s (spread %.0f %% of the value)"
               % (_mem_available_gb(), need_gb, points[0]["cells"], points[-1]["cells"], N, points[0]["nnz"],
                  points[-1]["nnz"], nnz_full, points[0]["s_per_iter"], points[-1]["s_per_iter"], full_s,
                  scaled[0], scaled[-1], 100 * spread))
    value = 1.0 / full_s

    # (ii) fused OpenMP on the whole matrix, its team size chosen the same way
    M = orc.FusedMatrix(X, dtype)
    bp, dp, st = _oracle_state(orc, X, K, dtype)
    trial = {}
    for n in _team_sizes(host):
        orc.fused_iteration(M, st, 0.3, 0.3, bp, dp, nthreads=n)          # warm-up of this team size
        t0 = time.perf_counter()
        orc.fused_iteration(M, st, 0.3, 0.3, bp, dp, nthreads=n)
        trial[n] = time.perf_counter() - t0
    fused_threads = min(trial, key=trial.get)
    dt_f, it_f = _time_iterations(lambda: orc.fused_iteration(M, st, 0.3, 0.3, bp, dp, nthreads=fused_threads),
                                  6.0, 10)
    return {
        "value": value, "unit": "iterations/s", "cores": host["usable"], "threads": threads_i, "kind": "port",
        "host": host,
        "sample": "variant (i) numba-structure C oracle (parallel Xphi + serial scatter-adds, the reference's "
                  "execution shape) at %d threads -- the fastest of the team sizes tried on a 1/16 row sample "
                  "(s per iteration: %s); the box shows %d logical CPUs, affinity %d, cgroup quota %s => %d usable -- "
                  "on %s" % (threads_i, trial_i_txt, host["logical"], host["affinity"],
                             "%.1f CPUs" % host["cgroup_cpu_max"] if host["cgroup_cpu_max"] else "none",
                             host["usable"], how),
        "whole_matrix": bool(whole), "extrapolation_spread": spread, "sample_points": points,
        "fused_openmp": {
            "value": 1.0 / dt_f, "unit": "iterations/s", "cores": host["usable"], "threads": fused_threads, "kind": "port",
            "sample": "variant (ii) fused OpenMP restatement (exp hoisted, no Xphi, parallel CSR + CSC "
                      "passes, AVX2), WHOLE matrix (nnz %d), %d timed iterations at %d threads -- the fastest of "
                      "the team sizes tried (s per iteration: %s) --, %.3f s/iter"
                      % (nnz_full, it_f, fused_threads,
                         ", ".join("%d: %.2f" % (k, v) for k, v in sorted(trial.items(), reverse=True)), dt_f),
        },
        "note": "CPU baseline = this build's restatements of the reference's path; numba itself cannot be "
                "installed here (SURVEY.md 8c)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--prewarm-s", type=float, default=0.3,
                    help="seconds of loss evaluations (read-only sweeps) before the warm-up steps (GPU clock ramp; 0 = none)")
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-sharded", action="store_true",
                    help="use the sharded driver (process group, exchange all-reduce) even with one rank")
    ap.add_argument("--eager", action="store_true",
                    help="single GPU: one library call (three launches) per timed iteration instead of ONE "
                         "schpf_steps call for all of them.  The default is the schpf_steps call -- a hipGraph "
                         "replay, what scHPF.fit issues between two loss checks; the kernel times for the "
                         "roofline then come from a second, eager pass.")
    ap.add_argument("--comm", default="library", choices=["library", "torch"],
                    help="sharded runs: all-reduce issued by the library (RCCL bound at run time, one call per "
                         "stretch of iterations) or by torch.distributed from Python (ShardedCAVI)")
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip the two rocprofv3 --pmc child passes that measure roofline.traffic (then the committed "
                         "static record is attached, if it is of this plan)")
    ap.add_argument("--pmc-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--pmc-k", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of sharded runs (nccl = RCCL; gloo only with --comm torch: the "
                         "exchange buffer is then staged through the host -- a plumbing check, not a measurement)")
    ap.add_argument("--same-gpu", action="store_true",
                    help="every rank on device 0 (one-GPU boxes: exercises the N > 1 code path of this script with "
                         "--comm torch --backend gloo; RCCL refuses two ranks on one device)")
    ap.add_argument("--fail-library-comm", action="store_true", help=argparse.SUPPRESS)   # tests: exercise the fall-back below
    ap.add_argument("--no-converge", action="store_true",
                    help="skip the wall-clock-to-convergence fits (second half of BASELINE.json's metric)")
    args = ap.parse_args()
    if args.pmc_child:
        _pmc_child(args.pmc_child, args.pmc_k, args.dtype, args.steps)
        return

    # the host driver of these boxes only supports dmabuf IPC: without this RCCL fails at the first
    # cross-process buffer exchange (hipIpcGetMemHandle: invalid argument)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher.  One rank per GPU under
        # torch.distributed.run on this node (127.0.0.1, a free port); rank 0 prints the one JSON line.
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        sys.stdout.flush()
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node",
                                  str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
                                  os.path.abspath(__file__)] + sys.argv[1:])
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    if args.same_gpu:
        local_rank = 0
    if args.backend == "gloo" and args.comm != "torch":
        raise SystemExit("--backend gloo needs --comm torch (the library's collective is RCCL)")
    torch.cuda.set_device(local_rank)
    sharded = world > 1 or args.force_sharded
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if args.backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))

    from schpf_amd import DeviceCAVI
    from schpf_amd.sharded import ShardedCAVI, exchange_tensor_of, row_partition, take_rows

    N, G, density, K = CONFIGS[args.config]
    dtype = np.float64 if args.dtype == "f64" else np.float32
    itemsize = np.dtype(dtype).itemsize
    # ONE matrix for every N: generator A with seed 42 (what BENCH's N = 1 line times).  With N > 1 every rank draws
    # it (deterministic; nothing but the communicator id travels between the ranks) and keeps its block of the
    # product's nnz-balanced row partition -- the split scHPF.fit(X, devices=[...]) makes.
    # The slab-drawn configurations (C5: 5e8 draws, > 12 GB of working set) are drawn PER RANK: no rank holds the whole
    # matrix, the partition and the hyperparameters' marginals come from all-reduced per-row / per-column sums
    # (synthetic_slabs_of_rank) -- the same matrix and the same row blocks as the whole-matrix path would give.
    import resource
    slab_rows = SLAB_CONFIGS.get(args.config)
    whole, my_rows, gen_facts = None, None, None
    t_gen = time.perf_counter()
    if slab_rows and world > 1:
        def all_reduce_np(a):
            t = torch.from_numpy(np.ascontiguousarray(a))
            if args.backend == "nccl":
                t = t.to("cuda:%d" % local_rank)
            dist.all_reduce(t)
            return t.cpu().numpy()
        threads = max(1, len(os.sched_getaffinity(0)) // world)      # the ranks share the box's cores
        X, bounds, gen_facts = synthetic_slabs_of_rank(N, G, density, 42, world, rank, all_reduce_np, slab_rows, threads)
        nnz_total = gen_facts["nnz_total"]
        my_rows = (int(bounds[rank]), int(bounds[rank + 1]))
    else:
        if slab_rows:                       # all of C5: the threaded slab generator (seconds, not minutes)
            X = synthetic_slabs(N, G, density, seed=42, slab_rows=slab_rows)
        else:
            X = synthetic_block(N, G, density, seed=42)
        nnz_total = int(X.nnz)
        bounds = row_partition(X, world)
        if world > 1:
            my_rows = (int(bounds[rank]), int(bounds[rank + 1]))
            whole = X
            X, _ = take_rows(whole, *my_rows)
    generation = {"seconds": time.perf_counter() - t_gen,
                  "host_peak_rss_gb": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6,
                  "how": ("per rank: this rank drew %d of %d slabs (%.3f of the whole matrix's %d draws)"
                          % (gen_facts["slabs_drawn"], gen_facts["slabs_total"],
                             gen_facts["draws"] / float(max(gen_facts["draws_whole_matrix"], 1)),
                             gen_facts["draws_whole_matrix"])) if gen_facts
                         else "the whole matrix on every rank"}
    if gen_facts:
        generation.update({k: gen_facts[k] for k in ("slabs_drawn", "slabs_total", "draws", "draws_whole_matrix")})
    n_local = X.shape[0]

    # the engine enqueues on a stream of its own; ShardedCAVI orders the collective with it
    eng = DeviceCAVI(n_local, G, K, dtype=dtype, device=local_rank)
    if sharded:
        eng.hint_sharded()
    t_up = time.perf_counter()
    if gen_facts:
        init_engine_of_rank(eng, X, K, dtype, gen_facts["row_sums"], gen_facts["col_sums"], rank)
    else:
        init_engine(eng, X, K, dtype, whole=whole, rows=my_rows)
    upload_s = time.perf_counter() - t_up
    del whole
    nnz_local = X.nnz
    # one library call for all K timed iterations, as scHPF.fit issues them between two loss checks: a
    # hipGraph replay on one GPU (an odd K: K - 1 iterations replayed + one eager), and for sharded runs
    # with the library's collective one call that keeps the launch queue full (-12 % per iteration against
    # one call per iteration; SCHPF_GRAPH_SHARDED=1 also captures those in a graph)
    use_graph = (not args.eager and not sharded) or (sharded and args.comm == "library")
    comm_fallback = None
    if sharded and args.comm == "library":
        from schpf_amd.sharded import NativeShard
        # rank 0's communicator id reaches the other ranks through the process group that is there anyway.  The
        # library's RCCL path with more than one rank has never met hardware: if its communicator cannot be set up on
        # ANY rank (agreed through the process group), every rank takes the torch.distributed driver of the same
        # protocol and kernels instead of leaving the run without a line, and the line says so
        drv, why = None, ""
        try:
            uid = [DeviceCAVI.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            if args.fail_library_comm:
                raise RuntimeError("--fail-library-comm")
            # ncclCommInitRank is collective: if ONE rank fails before it, the others would wait in it forever and the
            # run would end without a line -- a watchdog turns that into an error exit after three minutes
            import threading
            made = {}
            def make_shard():
                try:
                    made["drv"] = NativeShard(eng, uid[0], rank, world)
                except Exception as e:      # noqa: BLE001 -- re-raised on the main thread below
                    made["error"] = e
            th = threading.Thread(target=make_shard, daemon=True)
            th.start()
            th.join(180.0)
            if th.is_alive():
                sys.stderr.write("rank %d: the library's communicator did not come up within 180 s (another rank failed "
                                 "before ncclCommInitRank?); giving up\n" % rank)
                sys.stderr.flush()
                os._exit(3)
            if "error" in made:
                raise made["error"]
            drv = made["drv"]
        except Exception as e:   # noqa: BLE001 -- whatever it is, the other driver is the answer
            why = "%s: %s" % (type(e).__name__, e)
        ok = torch.tensor([1 if drv is not None else 0], dtype=torch.int32,
                          device=("cuda:%d" % local_rank) if args.backend == "nccl" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if drv is not None:
                eng.comm_destroy()
            comm_fallback = why or "another rank could not set up the library's communicator"
            args.comm = "torch"
            use_graph = False
            drv = ShardedCAVI(eng, exchange_tensor_of(eng, local_rank))
        step = drv.step
        loss_fn = drv.mean_negative_pois_llh
    elif sharded:
        drv = ShardedCAVI(eng, exchange_tensor_of(eng, local_rank))
        step = drv.step
        loss_fn = drv.mean_negative_pois_llh
    else:
        step = eng.step
        loss_fn = eng.mean_negative_pois_llh

    def fence():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    # t = 0 responsibilities (device generator, keyed by (seed, LOCAL cell, gene)): every rank its own stream of it
    eng.init_phi_device(12345 + 0x9E3779B97F4A7C15 * rank if world > 1 else 12345)
    # Device warm-up, untimed and outside the W warm-up steps: a fresh process reaches the timed region after seconds of
    # host-side work (matrix generation, plan build) with the GPU in a low power state, and a 15 ms timed region (the
    # driver's --steps 20) then measures the clock ramp -- round 3: 0.77 ms per iteration there against 0.70 in a
    # 100-step run of the same build on the same box class.  The warm-up work is LOSS EVALUATIONS (the same sweep
    # kernel in its read-only mode): they do not advance the model, so the W warm-up steps and the K timed steps still
    # start from the t = 0 state and `loss_after_warmup` / `loss_after_steps` still show the fit moving.
    prewarm_evals = 0
    if args.prewarm_s > 0 and world > 1:
        prewarm_evals = 200          # ranks must issue the same number of collectives: a count, not a clock
        for _ in range(prewarm_evals):
            loss_fn()
    elif args.prewarm_s > 0:
        t_pw = time.perf_counter()
        while time.perf_counter() - t_pw < args.prewarm_s:
            loss_fn()
            prewarm_evals += 1
    for _ in range(args.warmup):
        step()
    many = getattr(drv, "steps", None) if sharded else eng.steps   # sharded: a graph only with SCHPF_GRAPH_SHARDED=1
    if use_graph:
        many(args.steps)                # untimed: captures the K-iteration graph the timed call replays
    loss_start = loss_fn()
    # shader clock under the sweep launches (schpf_profile_clock: workgroup 0 of every launch stamps the shader-cycle
    # and the constant-rate counters; it works inside a replayed graph too): once over the timed call itself, once
    # over the eager pass the kernel times come from
    if use_graph:
        fence()
        eng.profile_clock()             # reset
        t0 = time.perf_counter()
        many(args.steps)                # EXACTLY K iterations, one library call (one hipGraph launch)
        fence()
        elapsed = time.perf_counter() - t0
        sclk_timed = eng.profile_clock()
        eng.profile(True)               # kernel times for the roofline: a second, eager pass of K iterations
        eng.profile_read()
        for _ in range(args.steps):
            step()
        prof = eng.profile_read()
        sclk_eager = eng.profile_clock()
        eng.profile(False)
    else:
        eng.profile(True)
        eng.profile_read()
        fence()
        eng.profile_clock()             # reset
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
        prof = eng.profile_read()
        sclk_timed = sclk_eager = eng.profile_clock()
        eng.profile(False)
    if sharded:
        el = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())

    # loss evaluation (every check_freq = 10 iterations in a default fit), outside the timed region: the median of five,
    # each right behind an iteration as in a fit (a single sample of a 0.4 ms call read 0.38-0.46 on one build);
    # loss_after_steps is the first one's value, i.e. the loss after exactly the K timed iterations
    loss_samples = []
    loss_end = None
    for i in range(5):
        if i:
            step()
        fence()
        t1 = time.perf_counter()
        val = loss_fn()
        torch.cuda.synchronize()
        loss_samples.append((time.perf_counter() - t1) * 1e3)
        if loss_end is None:
            loss_end = val
    loss_ms = float(np.median(loss_samples))

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
    # The compute roof of the same launch.  Essential arithmetic of the two-pass form: per stored nonzero and
    # orientation K FMAs for the normaliser s = sum_k Et[i,k] Eb[g,k] and K FMAs for acc_k += (x / s) Eb[g,k]
    # (sweep_impl.h pipe_step) = 4 K FMAs = 8 K flop per nonzero over both orientations (the reciprocal, the
    # cross-lane sum, decode and addressing are overhead, not counted).  Vector FP peak of the dtype, no MFMA.
    flops_iter = 8.0 * K * nnz_local
    flops_launch = flops_iter / per_iter
    valu_peak = VALU_PEAK_TFLOPS[args.dtype]
    valu_achieved = flops_launch / (sweep_ms * 1e-3) / 1e12 if sweep_ms > 0 else 0.0
    hbm_frac, valu_frac = achieved / HBM_PEAK_GBS, valu_achieved / valu_peak
    # The sustained clock scales the two on-chip roofs (the HBM roof does not move with it): the vector-FP peak is
    # quoted at the 2.4 GHz maximum, the LDS delivers 256 B per clock and CU (MI355X_MICROARCH.md, LDS section).
    sclk_mhz = sclk_eager[0] if sclk_eager[1] > 0 else None       # over the launches the kernel times come from
    clock_ratio = (sclk_mhz / MAX_CLOCK_MHZ) if sclk_mhz else 1.0
    valu_frac_at_sclk = valu_frac / clock_ratio
    n_cu = torch.cuda.get_device_properties(local_rank).multi_processor_count
    sb = eng.sweep_bytes()
    lds_alg = (sb["lds_read_nonzeros"] + sb["lds_staged_cell"] + sb["lds_staged_gene"]) / per_iter
    lds_exec = (sb["lds_read_stored_slots"] + sb["lds_staged_cell"] + sb["lds_staged_gene"]) / per_iter
    lds_peak = n_cu * LDS_BYTES_PER_CLK_PER_CU * (sclk_mhz or MAX_CLOCK_MHZ) * 1e6 / 1e9       # GB/s at the sustained clock
    lds_achieved = lds_alg / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0
    lds_frac = lds_achieved / lds_peak if lds_alg else 0.0
    bound = max((("hbm", hbm_frac), ("valu_fp%d" % (8 * itemsize), valu_frac_at_sclk), ("lds", lds_frac)),
                key=lambda kv: kv[1])[0]
    # what the kernel issues for it (tile plans): FMA wave-instructions = 2 sides x nnz x 2 KL / (64 / LPC) lane
    # groups per wave; the nonzero slots the plan stores say how many of the executed step halves carry
    # nonzeros; per wave step the f64 paired loop issues ~2 x 2 KL FMAs + 24 other VALU instructions
    # (rocprofv3 SQ_INSTS_VALU in profiles/)
    slots = info["entry_slots_cell"] + info["entry_slots_gene"]
    tile = info["chunk_len"] < 0
    # entry_slots_* count stored NONZERO slots (a step slot holds two); both orientations store every nonzero
    slot_fill = (2.0 * nnz_local / float(slots)) if (tile and slots) else None
    kl, lpc = info["KL"], info["LPC"]
    fma_wave_insts = 4.0 * kl * nnz_local / (64.0 / lpc)
    essential_over_issued = (4.0 * kl) / (4.0 * kl + 24.0) if (tile and args.dtype == "f64") else None
    # the counters were collected on the one-launch (dual) iteration of a single GPU
    traffic, traffic_src = static_traffic(args.config, args.dtype, info) if not sharded else (None, None)

    out = {
        "metric": "CAVI iterations/sec, 100kx20k K=20" if args.config == "c3"
                  else "CAVI iterations/sec (%s)" % args.config,
        "value": value, "unit": "iterations/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {
            "workload": "%s: synthetic %d cells x %d genes, density %.3f (negative-binomial counts, "
                        "RandomState(42%s), the same matrix for every N), nnz %d after summing duplicates, "
                        "K=%d, one CAVI iteration per step (no loss evaluation inside the step)"
                        % (args.config.upper(), N, G, density, " + slab: slabs of %d cells" % slab_rows if slab_rows else "",
                           nnz_total, K),
            "parallelism": ("cells row-sharded x%d by schpf_amd.sharded.row_partition (nnz-balanced contiguous row "
                            "blocks; this rank: rows %d..%d, nnz %d), one RCCL all-reduce of G*K+K per iteration (%s)"
                            % (world, my_rows[0], my_rows[1], nnz_local,
                               "issued by the library" if args.comm == "library" else "torch.distributed"))
                           if world > 1 else "single GPU",
            "launch": ("one library call for the %d timed iterations (%s), after %d untimed iterations of the same call"
                       % (args.steps, "schpf_steps_sharded; a hipGraph with one rank or SCHPF_GRAPH_SHARDED=1" if sharded
                          else "schpf_steps: one hipGraph replay", args.steps)
                       + "; before the %d warm-up steps %d loss evaluations (read-only sweeps, %.1f s) bring the GPU out of its idle clocks"
                       % (args.warmup, prewarm_evals, args.prewarm_s))
                      if use_graph else "one library call per iteration, eager launches",
            "plan": info,
            "generation": generation,
            **({"comm_fallback": "the library's RCCL communicator could not be set up (%s): torch.distributed "
                                 "issues the all-reduce instead, one call per iteration" % comm_fallback}
               if comm_fallback else {}),
        },
        "roofline": {
            # the nearer roof of the two below; `achieved` / `peak` / `frac` keep SURVEY 8(d)'s definition
            # (algorithmic bytes per launch / launch time against the HBM peak) whichever one binds
            "bound": bound,
            "bound_how": "the largest of hbm_frac, fp%d_valu_frac_at_sclk and lds.frac (the two on-chip roofs at the "
                         "shader clock measured under the kernel)" % (8 * itemsize),
            "sclk_mhz": sclk_mhz, "sclk_mhz_timed_call": sclk_timed[0] if sclk_timed[1] > 0 else None,
            "sclk_how": "schpf_profile_clock: workgroup 0 of every sweep launch reads s_memtime (shader cycles) and "
                        "s_memrealtime (constant rate) on entry and exit; sclk_mhz averages the %d launches of the eager "
                        "pass the kernel times come from, sclk_mhz_timed_call the %d launches inside the timed call; "
                        "max clock %d MHz" % (sclk_eager[1], sclk_timed[1], MAX_CLOCK_MHZ),
            "kernel": ("tile_sweep_dual_kernel (cell-side + gene-side sweep in one launch; algorithmic "
                       "bytes per launch = B_iter" if per_iter == 1 else
                       "tile_sweep_kernel (cell + gene launches; algorithmic bytes per launch = B_iter/2")
                      + ", B_iter = 12*nnz + 4*K*s*(N+G) + 2*s*(N+G))",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": hbm_frac, "traffic": traffic, "traffic_source": traffic_src,
            "hbm_frac": hbm_frac,
            "valu": {
                "what": "essential FMAs of the launch against the vector FP%d peak (no MFMA on this path): 4 K FMAs = "
                        "8 K flop per nonzero over both orientations (K for the normaliser + K for the accumulation, "
                        "per side)" % (8 * itemsize),
                "flops_per_launch": flops_launch, "achieved": valu_achieved, "peak": valu_peak, "unit": "TFLOP/s",
                "frac": valu_frac,
                "fma_wave_instructions_per_launch": fma_wave_insts / per_iter,
                "essential_over_issued_valu": essential_over_issued,
                "slot_fill": slot_fill,
                "note": "essential_over_issued_valu = FMAs / (FMAs + the ~24 other VALU instructions of a wave step of "
                        "the f64 paired loop); slot_fill = share of the stored step slots that carry a nonzero (a padding "
                        "slot executes every instruction of a nonzero); measured SQ_INSTS_VALU per launch: profiles/",
            },
            "fp%d_valu_frac" % (8 * itemsize): valu_frac,
            "fp%d_valu_frac_at_sclk" % (8 * itemsize): valu_frac_at_sclk,
            "lds": {
                "what": "LDS bytes of the launch against %d CUs x %d B/clk x the measured shader clock: reads = one table "
                        "row of KP = %d values per nonzero and orientation (2 * nnz * KP * %d B), + the window stagings "
                        "(LDS writes) the plans schedule (schpf_sweep_bytes)" % (n_cu, LDS_BYTES_PER_CLK_PER_CU,
                                                                              info["KP"], itemsize),
                "bytes_per_launch": lds_alg, "read_bytes_per_launch": sb["lds_read_nonzeros"] / per_iter,
                "staged_bytes_per_launch": (sb["lds_staged_cell"] + sb["lds_staged_gene"]) / per_iter,
                "bytes_per_launch_incl_padding_slots": lds_exec,
                "achieved": lds_achieved, "peak": lds_peak, "unit": "GB/s", "frac": lds_frac,
                "frac_at_max_clock": lds_frac * clock_ratio,
            },
            "algorithmic_gb_per_launch": b_launch / 1e9, "sweep_launches_per_iteration": per_iter,
            "avg_launch_ms": sweep_ms, "launches": sweeps,
            "timed_with": ("HIP events on the engine's stream around every launch, over a second pass of the same %d "
                           "iterations issued eagerly right after the timed call (events cannot sit inside the "
                           "replayed hipGraph / the single library call)" % args.steps) if use_graph
                          else "HIP events on the engine's stream around every launch of the timed iterations",
            "cell_sweep_ms": prof["cell_sweep"]["ms"] / max(prof["cell_sweep"]["launches"], 1),
            "gene_sweep_ms": prof["gene_sweep"]["ms"] / max(prof["gene_sweep"]["launches"], 1),
            "gamma_updates_ms": prof["gamma_updates"]["ms"] / max(prof["gamma_updates"]["launches"], 1),
            "iteration_frac_of_hbm_peak": (b_iter / (ms_per_step * 1e-3) / 1e9) / HBM_PEAK_GBS,
        },
        "loss_eval_ms": loss_ms, "loss_eval_samples_ms": [round(v, 4) for v in loss_samples],
        "loss_pass": {"plan": "gene-major" if sb["loss_side"] else "cell-major", "tasks": sb["loss_tasks"],
                      "how": "chosen by the library's list-schedule model of the pass (capi.hip loss_tasks / loss_side)"},
        "loss_after_warmup": loss_start, "loss_after_steps": loss_end,
        "iterations_per_s_with_loss_every_10": 10.0 / (10 * ms_per_step * 1e-3 + loss_ms * 1e-3),
        "upload_and_plan_s": upload_s,
    }
    if rank == 0 and world == 1 and sharded and args.config in SHARD_OF and SHARD_OF[args.config][0] == "c3":
        # a rank's iteration beside its ideal: the WHOLE matrix on this GPU in the same process, divided by the ranks
        # (what perfect strong scaling without any exchange would give the rank)
        eng.close()
        whole_cfg, ranks = SHARD_OF[args.config]
        Nw, Gw, dw, Kw = CONFIGS[whole_cfg]
        Xw = synthetic_block(Nw, Gw, dw, seed=42)
        with DeviceCAVI(Nw, Gw, Kw, dtype=dtype, device=local_rank) as whole_eng:
            init_engine(whole_eng, Xw, Kw, dtype)
            whole_eng.init_phi_device(12345)
            for _ in range(args.warmup):
                whole_eng.step()
            whole_eng.steps(args.steps)                 # captures the graph
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            whole_eng.steps(args.steps)
            whole_eng.synchronize()
            whole_ms = (time.perf_counter() - t0) / args.steps * 1e3
        out["per_rank"] = {"ms": ms_per_step, "ideal_ms": whole_ms / ranks, "whole_matrix_ms": whole_ms, "ranks": ranks,
                           "speedup_bound_before_the_exchange": whole_ms / ms_per_step,
                           "how": "this line's ms_per_step (a 1/%d row shard of %s through the sharded driver, one-rank "
                                  "all-reduce) beside 1/%d of the whole %s matrix's iteration on this GPU in the same "
                                  "process" % (ranks, whole_cfg.upper(), ranks, whole_cfg.upper())}
        del Xw
    if rank == 0 and world == 1 and not sharded and not args.no_traffic:
        eng.close()
        live, how = live_traffic(X, K, args.dtype)
        if live is not None:
            out["roofline"]["traffic"], out["roofline"]["traffic_source"] = live, how
        else:
            out["roofline"]["traffic_note"] = "live PMC pass unavailable (%s); static record used if any" % how
    if rank == 0 and world == 1 and not args.no_converge:
        eng.close()
        out["convergence"] = convergence_run(N, G, K, dtype, density)
        out["metric"] += " + wall-clock to convergence (see 'convergence')"
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(X, K, dtype)
        if "convergence" in out:     # the same fit at the CPU rates (iterations only, scaled by nnz)
            scale = out["convergence"]["nnz"] / float(nnz_total)
            its = out["convergence"]["iterations"]
            out["convergence"]["cpu_estimate_s"] = {
                "numba_structure": its * scale / out["cpu_baseline"]["value"],
                "fused_openmp": its * scale / out["cpu_baseline"]["fused_openmp"]["value"],
                "how": "iterations of the median GPU fit x (nnz of the planted matrix / nnz of the "
                       "throughput matrix) / CPU iterations per second; loss checks and set-up not counted"}
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
