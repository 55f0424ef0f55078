"""One rank of the two-process, ONE-GPU exchange test (launched by tests/test_multigpu.py through
`python -m torch.distributed.run --nproc-per-node 2`).  Test infrastructure, not product.

RCCL refuses two ranks on one device, so the one-GPU boxes cannot run the product's collective between two
processes.  What they can run is everything else of the two-rank protocol for real: two PROCESSES, each with
its own DeviceCAVI on cuda:0 holding its nnz-balanced row block, the HIP packing / update-from-exchange
kernels, and ShardedCAVI driving them -- with torch.distributed's gloo backend carrying the all-reduce of the
device-resident exchange buffer (staged through the host by gloo).  Every rank checks its cells' theta / xi
and the replicated beta / eta against the oracle's iteration on the whole matrix, then the all-reduced loss.
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
    ap.add_argument("--dtype", default="f64")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from numpy.testing import assert_allclose
    from conftest import synthetic_counts
    from oracle import hpf_oracle as orc
    from schpf_amd import DeviceCAVI
    from schpf_amd.sharded import ShardedCAVI, exchange_tensor_of, row_partition, take_rows

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc.build()
    dtype = np.dtype(np.float64 if args.dtype == "f64" else np.float32)
    f32 = dtype == np.float32
    X = synthetic_counts(4000, 2500, 0.03, seed=23)
    K, a, c = 12, 0.3, 0.3
    np.random.seed(7)
    bp, dp, st = orc.setup_state(X, K, dtype, a, 1.0, c, 1.0)
    st.xi_shape[:] = 1.0 + K * a
    st.eta_shape[:] = 1.0 + K * c
    bounds = row_partition(X, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    sub, _ = take_rows(X, lo, hi)
    with DeviceCAVI(hi - lo, X.shape[1], K, dtype=dtype, device=0) as eng:
        eng.hint_sharded()
        eng.upload(sub)
        eng.set_hypers(a, c, bp, dp)
        eng.set_gamma("xi", st.xi_shape[lo:hi], st.xi_rate[lo:hi])
        eng.set_gamma("theta", st.theta_shape[lo:hi], st.theta_rate[lo:hi])
        eng.set_gamma("eta", st.eta_shape, st.eta_rate)
        eng.set_gamma("beta", st.beta_shape, st.beta_rate)
        drv = ShardedCAVI(eng, exchange_tensor_of(eng, 0))
        plan = [{}, {}, {"simultaneous": True}, {"freeze_genes": True}, {}, {}]
        for done, flags in enumerate(plan, 1):
            drv.step(**flags)
            orc.cavi_iteration(X.data, X.row, X.col, st, a, c, bp, dp, **flags)
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
        assert_allclose(drv.mean_negative_pois_llh(), want, rtol=1e-5 if f32 else 1e-10)
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d/%d: parity ok over %d sharded iterations on one GPU (gloo transport, %s)" % (rank, world, len(plan), args.dtype))


if __name__ == "__main__":
    main()
