"""Cells sharded over the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference has no distributed layer; what it offers to shard on is
the model itself: given beta/eta, cells are independent (schpf/scHPF_.py:706-714),
and genes only need sums over cells (:699-703).  So rank p keeps a contiguous block
of rows of X with its xi/theta slices and a replica of eta/beta, and one iteration is

    step_local   gene-side sweep on the local rows; its sums (G*K) and the local
                 sum_i E[theta_ik] (K) are packed into one device buffer
    all_reduce   ONE sum all-reduce of that buffer (torch.distributed; backend "nccl"
                 is RCCL on ROCm), started asynchronously ...
    step_local   ... so that the cell-side sweep runs underneath it
    step_finish  every rank applies the identical beta/eta update to its replica,
                 then its own theta/xi update

With freeze_genes (project) there is nothing to exchange.  The loss needs a second,
3-scalar all-reduce on check iterations only.  The engine is duck-typed
(step_local / step_finish / exchange_tensor / loss_terms) so the protocol is testable
on CPU with the gloo backend and an oracle-backed engine (tests/test_sharded_cpu.py).
"""
import numpy as np

__all__ = ["row_partition", "take_rows", "ShardedCAVI", "exchange_tensor_of"]


def row_partition(X, world_size):
    """Contiguous row ranges balanced by stored nonzeros.  Returns world_size+1 bounds."""
    counts = np.bincount(X.row, minlength=X.shape[0]).astype(np.int64)
    cum = np.concatenate([[0], np.cumsum(counts)])
    targets = cum[-1] * np.arange(1, world_size) / world_size
    inner = np.searchsorted(cum, targets, side="left")
    bounds = np.concatenate([[0], inner, [X.shape[0]]]).astype(np.int64)
    return np.maximum.accumulate(bounds)


def take_rows(X, lo, hi):
    """Rows [lo, hi) of a COO matrix as a COO matrix with local row indices, stored order
    preserved; also returns the positions of the kept nonzeros in X."""
    from scipy.sparse import coo_matrix
    keep = np.flatnonzero((X.row >= lo) & (X.row < hi))
    sub = coo_matrix((X.data[keep], (X.row[keep] - lo, X.col[keep])), shape=(int(hi - lo), X.shape[1]))
    return sub, keep


class _DevicePointer(object):
    """Minimal __cuda_array_interface__ carrier so torch can view library-owned HBM."""

    def __init__(self, ptr, count, dtype):
        self.__cuda_array_interface__ = {
            "shape": (int(count),), "typestr": np.dtype(dtype).str, "data": (int(ptr), False),
            "version": 2, "strides": None}


def exchange_tensor_of(engine, device_index):
    """torch view (no copy) of a DeviceCAVI's exchange buffer."""
    import torch
    ptr, count = engine.exchange_buffer()
    return torch.as_tensor(_DevicePointer(ptr, count, engine.dtype), device="cuda:%d" % device_index)


class ShardedCAVI(object):
    """Drives one rank's engine through the sharded iteration."""

    def __init__(self, engine, exchange_tensor, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.engine = engine
        self.exchange = exchange_tensor
        self.group = group
        # Stream ordering.  torch.distributed orders a collective with torch's CURRENT stream only
        # (the process group's own stream waits for an event recorded there, and work.wait() makes
        # the current stream wait for the collective).  The engine's kernels run on the engine's
        # stream, so that stream is made torch's current stream around every collective: the
        # all-reduce then starts after the packing kernel and step_finish after the all-reduce.
        self._stream_ctx = None
        if getattr(exchange_tensor, "is_cuda", False) and hasattr(engine, "stream_handle"):
            import torch
            h = engine.stream_handle()
            dev = exchange_tensor.device
            stream = torch.cuda.default_stream(dev) if h == 0 else torch.cuda.ExternalStream(h, device=dev)
            self._stream_ctx = lambda: torch.cuda.stream(stream)

    def _on_engine_stream(self):
        import contextlib
        return self._stream_ctx() if self._stream_ctx is not None else contextlib.nullcontext()

    def step(self, freeze_genes=False, simultaneous=False):
        if freeze_genes:                       # nothing to exchange
            self.engine.step_local(freeze_genes=True, simultaneous=simultaneous)
            self.engine.step_finish(freeze_genes=True, simultaneous=simultaneous)
            return
        # gene-side sweep -> start the all-reduce of its sums -> cell-side sweep runs under it
        self.engine.step_local(simultaneous=simultaneous, side="gene")
        with self._on_engine_stream():
            work = self.dist.all_reduce(self.exchange, op=self.dist.ReduceOp.SUM, group=self.group,
                                        async_op=True)
            self.engine.step_local(simultaneous=simultaneous, side="cell")
            work.wait()
        self.engine.step_finish(simultaneous=simultaneous)

    def mean_negative_pois_llh(self):
        import torch
        llh, gl, nnz = self.engine.loss_terms()
        t = torch.tensor([llh, gl, float(nnz)], dtype=torch.float64, device=self.exchange.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        llh, gl, nnz = t.tolist()
        return -(llh - gl) / nnz
