"""One rank of the two-process multi-GPU parity test (launched by tests/test_multigpu.py through
`python -m torch.distributed.run --nproc-per-node W`).  Test infrastructure, not product.

Every rank draws the SAME synthetic matrix (BASELINE C2 size by default), keeps its nnz-balanced row
block (schpf_amd.sharded.row_partition), runs the sharded iteration with the collective inside the
library (NativeShard: schpf_comm_init + schpf_steps_sharded over RCCL) and checks ITS rows of
theta/xi and the replicated beta/eta against the oracle's iteration on the whole matrix, then the
all-reduced loss.  Exit status 0 = parity on this rank.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=10000)
    ap.add_argument("--genes", type=int, default=5000)
    ap.add_argument("--density", type=float, default=0.03)
    ap.add_argument("--nfactors", type=int, default=10)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--graph", type=int, default=0)
    args = ap.parse_args()
    if args.graph:
        os.environ["SCHPF_GRAPH_SHARDED"] = "1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from numpy.testing import assert_allclose
    from conftest import synthetic_counts
    from oracle import hpf_oracle as orc
    from schpf_amd import DeviceCAVI
    from schpf_amd.sharded import NativeShard, row_partition, take_rows

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    orc.build()
    dtype = np.dtype(np.float64 if args.dtype == "f64" else np.float32)
    f32 = dtype == np.float32
    X = synthetic_counts(args.cells, args.genes, args.density, seed=42)
    K, a, c = args.nfactors, 0.3, 0.3
    np.random.seed(5)
    bp, dp, st = orc.setup_state(X, K, dtype, a, 1.0, c, 1.0)
    st.xi_shape[:] = 1.0 + K * a
    st.eta_shape[:] = 1.0 + K * c
    bounds = row_partition(X, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    sub, _ = take_rows(X, lo, hi)

    uid = [DeviceCAVI.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    with DeviceCAVI(hi - lo, args.genes, K, dtype=dtype, device=local) as eng:
        eng.hint_sharded()
        eng.upload(sub)
        eng.set_hypers(a, c, bp, dp)
        eng.set_gamma("xi", st.xi_shape[lo:hi], st.xi_rate[lo:hi])
        eng.set_gamma("theta", st.theta_shape[lo:hi], st.theta_rate[lo:hi])
        eng.set_gamma("eta", st.eta_shape, st.eta_rate)
        eng.set_gamma("beta", st.beta_shape, st.beta_rate)
        shard = NativeShard(eng, uid[0], rank, world)
        plan = [(1, {}), (4, {}), (4, {}), (1, {"simultaneous": True}), (2, {"freeze_genes": True}), (3, {})]
        done = 0
        for n, flags in plan:
            shard.steps(n, **flags)
            for _ in range(n):
                orc.cavi_iteration(X.data, X.row, X.col, st, a, c, bp, dp, **flags)
            done += n
            tol = 2e-5 * done if f32 else 1e-10
            got = {name: eng.get_gamma(name) for name in ("xi", "theta", "eta", "beta")}
            assert_allclose(got["theta"][0], st.theta_shape[lo:hi], rtol=tol, err_msg="theta.shape after %d" % done)
            assert_allclose(got["theta"][1], st.theta_rate[lo:hi], rtol=tol, err_msg="theta.rate after %d" % done)
            assert_allclose(got["xi"][1], st.xi_rate[lo:hi], rtol=tol, err_msg="xi.rate after %d" % done)
            assert_allclose(got["beta"][0], st.beta_shape, rtol=tol, err_msg="beta.shape after %d" % done)
            assert_allclose(got["beta"][1], st.beta_rate, rtol=tol, err_msg="beta.rate after %d" % done)
            assert_allclose(got["eta"][1], st.eta_rate, rtol=tol, err_msg="eta.rate after %d" % done)
        want = orc.mean_negative_pois_llh(X.data, X.row, X.col, st.theta_shape, st.theta_rate,
                                          st.beta_shape, st.beta_rate)
        assert_allclose(shard.mean_negative_pois_llh(), want, rtol=1e-5 if f32 else 1e-10)
        # the replicas must agree BITWISE: every rank applies the same update to the same all-reduced sums
        mine = torch.from_numpy(np.ascontiguousarray(eng.get_gamma("beta")[1])).cuda()
        ref = mine.clone()
        dist.broadcast(ref, src=0)
        assert bool(torch.equal(mine, ref)), "beta replicas differ between ranks"
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d/%d: parity ok over %d sharded iterations (%s, graph=%d)" % (rank, world, done, args.dtype, args.graph))


if __name__ == "__main__":
    main()
