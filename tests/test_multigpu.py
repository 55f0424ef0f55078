"""Multi-GPU parity, armed by the hardware: every test here enables itself when the box has at
least two (or four / eight) HIP devices and is skipped on the one-GPU boxes -- BASELINE.json
configs[3] (100k x 20k, K=20, cells sharded over 2/4/8 GPUs with an RCCL all-reduce of the
gene-side sums) and the sharded half of configs[4].  The protocol is the one that the CPU tests
(tests/test_sharded_cpu.py, gloo) and the one-GPU emulation (test_engine_gpu.py) cover; here it
runs over real RCCL ranks:

  * `scHPF.fit(X, devices=[0, 1])` (one process, one host thread per GPU, the collective inside the
    library) against the reference's golden fit traces;
  * two / four / eight PROCESSES under torch.distributed.run, NativeShard, against the oracle at
    BASELINE C2 size (tests/_multigpu_worker.py), eager and with the stretch captured as a hipGraph;
  * `bench.py --gpus N` as a smoke of the driver's own launch line.

One two-process case needs no second GPU and runs everywhere: both ranks on cuda:0, the all-reduce of the
device-resident exchange buffer carried by gloo (tests/_gloo_gpu_worker.py).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
from numpy.testing import assert_allclose

from conftest import ROOT, load_golden, golden_coo

pytestmark = pytest.mark.gpu


def _device_count():
    from schpf_amd import _lib
    return _lib.device_count()


def need_gpus(n):
    if _device_count() < n:
        pytest.skip("needs %d HIP devices, this box has %d" % (n, _device_count()))


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(world, script_args, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


FITS = [("fit_data_k5_s0_f64.npz", np.float64, {}),
        ("fit_conf_k4_s0_f32.npz", np.float32, {}),
        ("fit_data_k5_s0_f64_simul.npz", np.float64, {"beta_theta_simultaneous": True})]


@pytest.mark.parametrize("fname,dtype,kw", FITS)
@pytest.mark.parametrize("ndev", [2, 4, 8])
def test_fit_over_real_devices_reproduces_reference_trace(fname, dtype, kw, ndev):
    """scHPF.fit(X, devices=[0..ndev-1]) with the library's RCCL communicator (ThreadedShards,
    comm="rccl"): same stop iteration and losses as the reference's single-process run; parameters to
    the sharded summation order."""
    need_gpus(ndev)
    from schpf import scHPF
    g = load_golden(fname)
    X = golden_coo(g)
    np.random.seed(int(g["seed"]))
    model = scHPF(int(g["nfactors"]), dtype=dtype, max_iter=int(g["max_iter"]), verbose=False)
    model.fit(X, devices=list(range(ndev)), **kw)
    f32 = np.dtype(dtype) == np.float32
    assert len(model.loss) == len(g["loss"])
    assert_allclose(model.loss, g["loss"], rtol=1e-4 if f32 else 1e-9)
    for name in ("theta", "beta"):
        got = getattr(model, name)
        want = g[name + "_shape"].astype(np.float64) / g[name + "_rate"].astype(np.float64)
        assert_allclose(got.vi_shape.astype(np.float64) / got.vi_rate.astype(np.float64), want,
                        rtol=1e-3 if f32 else 1e-6, err_msg=name)


@pytest.mark.parametrize("graph", [0, 1])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_native_shards_under_torchrun_match_oracle(world, dtype, graph):
    """One process per GPU (the layout bench.py --gpus N and a production launch use): every rank
    checks its rows against the oracle's iteration on the whole C2-size matrix (worker docstring).
    world = 1 runs on every box and keeps the worker itself honest."""
    need_gpus(world)
    r = _torchrun(world, ["tests/_multigpu_worker.py", "--dtype", dtype, "--graph", str(graph)])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("parity ok") == world, r.stdout[-2000:]


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_two_processes_share_one_gpu_with_a_gloo_exchange(dtype):
    """Runs on EVERY box: two processes, each a rank with its own DeviceCAVI on cuda:0, the product's packing
    and update-from-exchange kernels and ShardedCAVI, the all-reduce of the device-resident exchange buffer
    carried by gloo (RCCL refuses two ranks on one device) -- against the oracle (tests/_gloo_gpu_worker.py)."""
    r = _torchrun(2, ["tests/_gloo_gpu_worker.py", "--dtype", dtype])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("parity ok") == 2, r.stdout[-2000:]


def test_bench_line_of_two_ranks_on_one_gpu():
    """bench.py's N = 2 code path on a one-GPU box: two ranks on device 0, torch-driven exchange over gloo --
    row blocks per rank, the rank-summed nnz, the MAX-over-ranks timing and rank 0's JSON line (the numbers
    are not a measurement: gloo stages the exchange buffer through the host)."""
    r = _torchrun(2, ["bench.py", "--gpus", "2", "--config", "c2", "--steps", "10", "--warmup", "3", "--comm", "torch",
                      "--backend", "gloo", "--same-gpu", "--no-cpu-baseline", "--no-converge", "--no-traffic"])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 10 and line["scaling"] == "strong"
    assert np.isfinite(line["value"]) and line["value"] > 0
    assert np.isfinite(line["loss_after_steps"]) and line["loss_after_steps"] < line["loss_after_warmup"]


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2 ...` with NO launcher around it (no WORLD_SIZE in the environment): the script
    re-executes itself under torch.distributed.run and rank 0 prints exactly one JSON line.  Both ranks on
    device 0 over gloo, as above.  The matrix is the N = 1 matrix (seed 42) split by row_partition: the line's
    nnz is the unsharded nnz."""
    from bench import synthetic_block, CONFIGS
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--config", "c2", "--steps", "10", "--warmup", "3",
                        "--comm", "torch", "--backend", "gloo", "--same-gpu", "--no-cpu-baseline", "--no-converge",
                        "--no-traffic"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 10 and line["scaling"] == "strong"
    N, G, dens, K = CONFIGS["c2"]
    nnz = synthetic_block(N, G, dens, seed=42).nnz
    assert "nnz %d " % nnz in line["config"]["workload"]
    assert "row_partition" in line["config"]["parallelism"]
    assert np.isfinite(line["loss_after_steps"]) and line["loss_after_steps"] < line["loss_after_warmup"]


def test_bench_draws_c5_per_rank():
    """`bench.py --gpus 2 --config c5-small` (C5 at 1/20 of its cells, drawn slab by slab like C5): no rank draws the
    whole matrix -- each about half of it plus a boundary slab -- yet the line's nnz is the whole matrix's and the
    row blocks are the product's partition (synthetic_slabs_of_rank; tests/test_bench_host.py checks the blocks entry
    for entry).  Both ranks on device 0 over gloo, self-launched."""
    from bench import synthetic_slabs, CONFIGS, SLAB_CONFIGS
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--config", "c5-small", "--steps", "6", "--warmup", "2",
                        "--comm", "torch", "--backend", "gloo", "--same-gpu", "--no-cpu-baseline", "--no-converge",
                        "--no-traffic"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    N, G, dens, K = CONFIGS["c5-small"]
    whole = synthetic_slabs(N, G, dens, seed=42, slab_rows=SLAB_CONFIGS["c5-small"])
    assert "nnz %d " % whole.nnz in line["config"]["workload"]
    gen = line["config"]["generation"]
    assert gen["slabs_total"] == 40 and gen["slabs_drawn"] <= 22
    assert gen["draws"] <= 0.56 * gen["draws_whole_matrix"]
    assert line["n_gpus"] == 2 and np.isfinite(line["value"]) and line["value"] > 0
    assert np.isfinite(line["loss_after_steps"]) and line["loss_after_steps"] < line["loss_after_warmup"]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_self_launch_over_real_devices(world):
    """The driver's N > 1 line if it is NOT wrapped in a launcher: `python bench.py --gpus N` over RCCL."""
    need_gpus(world)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", str(world), "--config", "c2", "--steps", "20", "--warmup", "5",
                        "--no-cpu-baseline", "--no-converge", "--no-traffic"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == world and np.isfinite(line["value"]) and line["value"] > 0


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_bench_line_over_ranks(world):
    """The driver's launch line for N > 1 at C2 size: one JSON line from rank 0 with n_gpus = N, a
    finite rate and a loss that moved (world = 1: the same sharded driver with one rank)."""
    need_gpus(world)
    r = _torchrun(world, ["bench.py", "--gpus", str(world), "--config", "c2", "--steps", "20", "--warmup", "5",
                          "--no-cpu-baseline", "--no-converge", "--no-traffic"] + (["--force-sharded"] if world == 1 else []))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == world and line["steps"] == 20
    assert np.isfinite(line["value"]) and line["value"] > 0
    assert np.isfinite(line["loss_after_steps"]) and line["loss_after_steps"] < line["loss_after_warmup"]


@pytest.mark.parametrize("world", [1, 2])
def test_bench_falls_back_to_the_torch_collective(world):
    """`bench.py --gpus N` with the library's communicator failing on every rank (the one thing of that path that has
    never met hardware with N > 1): all ranks agree through the process group, take the torch.distributed driver of the
    same protocol, and the line says so instead of the run ending without one."""
    need_gpus(world)
    r = _torchrun(world, ["bench.py", "--gpus", str(world), "--config", "c2", "--steps", "10", "--warmup", "2",
                          "--no-cpu-baseline", "--no-converge", "--no-traffic", "--fail-library-comm"]
                  + (["--force-sharded"] if world == 1 else []))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == world and "comm_fallback" in line["config"]
    assert np.isfinite(line["value"]) and line["value"] > 0
    assert np.isfinite(line["loss_after_steps"]) and line["loss_after_steps"] < line["loss_after_warmup"]
