"""The cells-sharded iteration (schpf_amd/sharded.py) on CPU: world_size 2, gloo backend, an
oracle-backed engine per rank.  Two ranks holding row blocks of X must reproduce the
unsharded oracle (up to summation order: the gene-side sums are added per shard first)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from numpy.testing import assert_allclose

from conftest import synthetic_counts
from schpf_amd.sharded import ShardedCAVI, row_partition, take_rows


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_row_partition_balances_nonzeros():
    X = synthetic_counts(500, 80, 0.1, seed=1)
    for world in (1, 2, 3, 8):
        b = row_partition(X, world)
        assert b[0] == 0 and b[-1] == 500 and np.all(np.diff(b) >= 0) and len(b) == world + 1
        counts = [take_rows(X, b[r], b[r + 1])[0].nnz for r in range(world)]
        assert sum(counts) == X.nnz
        assert max(counts) - min(counts) <= 2 * X.nnz / 500 * 3 + 40       # within a few rows' worth
    sub, keep = take_rows(X, 10, 20)
    assert sub.shape == (10, 80) and np.array_equal(X.row[keep] - 10, sub.row)


def _worker(rank, world, port, flags, dtype_name, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _oracle_engine import OracleEngine
    from oracle import hpf_oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dtype = np.dtype(dtype_name)
    X = synthetic_counts(240, 150, 0.12, seed=7)
    K, a, c = 6, 0.3, 0.3
    np.random.seed(3)                       # every rank draws the same global initialisation
    bp, dp, st = orc.setup_state(X, K, dtype, a, 1.0, c, 1.0)
    st.xi_shape[:] = 1.0 + K * a
    st.eta_shape[:] = 1.0 + K * c
    xphi0 = X.data[:, None] * np.random.dirichlet(np.ones(K), X.nnz)
    bounds = row_partition(X, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    Xl, keep = take_rows(X, lo, hi)
    eng = OracleEngine(Xl, K, dtype)
    eng.set_hypers(a, c, bp, dp)
    eng.set_gamma("xi", st.xi_shape[lo:hi], st.xi_rate[lo:hi])
    eng.set_gamma("eta", st.eta_shape, st.eta_rate)
    eng.set_gamma("beta", st.beta_shape, st.beta_rate)
    eng.set_gamma("theta", st.theta_shape[lo:hi], st.theta_rate[lo:hi])
    drv = ShardedCAVI(eng, eng.exchange)
    eng.init_phi_host(xphi0[keep])          # t = 0: the caller's Dirichlet draws, local rows
    losses = []
    for t in range(4):
        drv.step(**flags)
        losses.append(drv.mean_negative_pois_llh())
    ths, thr = eng.get_gamma("theta")
    bes, ber = eng.get_gamma("beta")
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), lo=lo, hi=hi, ths=ths, thr=thr, bes=bes, ber=ber,
             eta_r=eng.get_gamma("eta")[1], xi_r=eng.get_gamma("xi")[1], losses=np.array(losses))
    dist.destroy_process_group()


@pytest.mark.parametrize("flags", [{}, {"simultaneous": True}, {"freeze_genes": True}])
@pytest.mark.parametrize("dtype_name", ["float64", "float32"])
def test_two_ranks_reproduce_the_unsharded_oracle(tmp_path, oracle, flags, dtype_name):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), flags, dtype_name, str(tmp_path)), nprocs=world, join=True)
    dtype = np.dtype(dtype_name)
    X = synthetic_counts(240, 150, 0.12, seed=7)
    K, a, c = 6, 0.3, 0.3
    np.random.seed(3)
    bp, dp, st = oracle.setup_state(X, K, dtype, a, 1.0, c, 1.0)
    st.xi_shape[:] = 1.0 + K * a
    st.eta_shape[:] = 1.0 + K * c
    xphi0 = X.data[:, None] * np.random.dirichlet(np.ones(K), X.nnz)
    want_losses = []
    for t in range(4):
        oracle.cavi_iteration(X.data, X.row, X.col, st, a, c, bp, dp, xphi=xphi0 if t == 0 else None, **flags)
        want_losses.append(oracle.mean_negative_pois_llh(X.data, X.row, X.col, st.theta_shape, st.theta_rate,
                                                         st.beta_shape, st.beta_rate))
    tol = 2e-4 if dtype == np.float32 else 1e-11
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    assert parts[0]["hi"] == parts[1]["lo"] and parts[1]["hi"] == 240
    ths = np.concatenate([p["ths"] for p in parts]); thr = np.concatenate([p["thr"] for p in parts])
    assert_allclose(ths, st.theta_shape, rtol=tol)
    assert_allclose(thr, st.theta_rate, rtol=tol)
    assert_allclose(np.concatenate([p["xi_r"] for p in parts]), st.xi_rate, rtol=tol)
    for p in parts:                         # the gene side is replicated, identically
        assert_allclose(p["bes"], st.beta_shape, rtol=tol)
        assert_allclose(p["ber"], st.beta_rate, rtol=tol)
        assert_allclose(p["eta_r"], st.eta_rate, rtol=tol)
        assert_allclose(p["losses"], want_losses, rtol=1e-5 if dtype == np.float32 else 1e-11)
    assert np.array_equal(parts[0]["bes"], parts[1]["bes"])


@pytest.mark.parametrize("kw", [{}, {"beta_theta_simultaneous": True}])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_fit_over_devices_through_the_estimator(oracle, kw, dtype):
    """scHPF.fit over two shards end to end on CPU (fit(X, devices=[0, 1]) builds the same ThreadedShards itself): the estimator's loop, ThreadedShards'
    partition / scatter / gather of the four Gammas, the per-stretch calls and the loss -- with
    oracle-backed stand-in engines in place of DeviceCAVI and the all-reduce summed on the host.
    Must reproduce the unsharded restatement of the reference's fit (same seed, same stop)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _oracle_engine import OracleShardEngine
    import schpf_amd.sharded as sharded
    from schpf import scHPF
    X = synthetic_counts(120, 90, 0.15, seed=9)
    K = 4
    np.random.seed(5)
    want = oracle.oracle_fit(X, K, dtype=dtype, max_iter=25, simultaneous=bool(kw))
    np.random.seed(5)
    model = scHPF(K, dtype=dtype, max_iter=25, verbose=False)
    with sharded.ThreadedShards(X, K, dtype, [0, 1], comm="emulated", engine_factory=OracleShardEngine) as shards:
        model.fit(X, engine=shards, **kw)
    f32 = np.dtype(dtype) == np.float32
    assert model.bp == want["bp"] and model.dp == want["dp"]
    assert len(model.loss) == len(want["loss"])
    assert_allclose(model.loss, want["loss"], rtol=1e-4 if f32 else 1e-10)
    st = want["state"]
    for name in ("xi", "theta", "eta", "beta"):
        got = getattr(model, name)
        assert got.vi_shape.shape == getattr(st, name + "_shape").shape
        assert_allclose(got.vi_shape, getattr(st, name + "_shape"), rtol=5e-3 if f32 else 1e-8, err_msg=name)
        assert_allclose(got.vi_rate, getattr(st, name + "_rate"), rtol=5e-3 if f32 else 1e-8, err_msg=name)


def _draw_worker(rank, world, port, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from bench import synthetic_slabs_of_rank
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def all_reduce(a):                       # what bench.py's main() passes in (gloo: host tensors)
        t = torch.from_numpy(np.ascontiguousarray(a))
        dist.all_reduce(t)
        return t.numpy()

    X, bounds, facts = synthetic_slabs_of_rank(6000, 900, 0.03, 42, world, rank, all_reduce, slab_rows=250, threads=1)
    np.savez(os.path.join(out_dir, "draw%d.npz" % rank), row=X.row, col=X.col, data=X.data, shape=np.array(X.shape),
             bounds=bounds, nnz_total=facts["nnz_total"], draws=facts["draws"], whole=facts["draws_whole_matrix"],
             row_sums=facts["row_sums"], col_sums=facts["col_sums"])
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_draw_their_own_rows_over_a_process_group(tmp_path, world):
    """bench.py --gpus N --config c5 on CPU: `world` gloo processes, each drawing only its part of the slab-drawn
    matrix (benchlib/data.py synthetic_slabs_of_rank, the two all-reduces through the real process group) -- together
    they hold exactly the whole-matrix draw split by the product's row_partition, with the global marginals."""
    from bench import synthetic_slabs
    mp.spawn(_draw_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    X = synthetic_slabs(6000, 900, 0.03, seed=42, slab_rows=250)
    want_bounds = row_partition(X, world)
    parts = [np.load(os.path.join(str(tmp_path), "draw%d.npz" % r)) for r in range(world)]
    for r, p in enumerate(parts):
        want, _ = take_rows(X, int(want_bounds[r]), int(want_bounds[r + 1]))
        assert np.array_equal(p["bounds"], want_bounds) and tuple(p["shape"]) == want.shape
        assert np.array_equal(p["row"], want.row) and np.array_equal(p["col"], want.col) and np.array_equal(p["data"], want.data)
        assert int(p["nnz_total"]) == X.nnz
        assert np.array_equal(p["row_sums"], np.asarray(X.sum(1)).ravel())
        assert np.array_equal(p["col_sums"], np.asarray(X.sum(0)).ravel())
        assert int(p["draws"]) <= int(p["whole"]) * (1.0 / world + 3.0 / 24) + 1
