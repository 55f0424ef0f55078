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
import inspect

import numpy as np

from .engine import mean_negative

__all__ = ["row_partition", "take_rows", "ShardedCAVI", "exchange_tensor_of", "ThreadedShards", "NativeShard"]


def row_partition(X, world_size):
    """Contiguous row ranges balanced by stored nonzeros.  Returns world_size+1 bounds."""
    return row_partition_from_counts(np.bincount(X.row, minlength=X.shape[0]), world_size)


def row_partition_from_counts(counts, world_size):
    """The same partition from the stored nonzeros per row alone (a rank that holds only part of the matrix can
    take part in computing it: bench.py draws C5 per rank)."""
    counts = np.asarray(counts).astype(np.int64)
    cum = np.concatenate([[0], np.cumsum(counts)])
    targets = cum[-1] * np.arange(1, world_size) / world_size
    inner = np.searchsorted(cum, targets, side="left")
    bounds = np.concatenate([[0], inner, [counts.shape[0]]]).astype(np.int64)
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
        return mean_negative(llh, gl, nnz)


class ThreadedShards(object):
    """One process, one host thread per GPU: the cells of X row-sharded over `devices`, every
    shard a DeviceCAVI on its device, all of them ranks of one RCCL communicator that lives
    inside the library (schpf_comm_init / schpf_steps_sharded).  This is what
    `scHPF.fit(X, devices=[...])` drives; it presents the single-engine interface `_fit` uses
    (set_hypers / set_gamma / get_gamma / init_phi_* / steps / mean_negative_pois_llh), with
    cell-side arrays split by the row partition and gene-side arrays replicated.

    ctypes releases the GIL during library calls, so the shards' launches really are issued
    concurrently -- which the collective needs: every rank must enqueue its all-reduce.
    """

    def __init__(self, X, nfactors, dtype, devices, comm="rccl", engine_factory=None):
        """comm="rccl": the product path.  comm="emulated" (tests on a one-GPU box, where RCCL
        refuses two ranks on one device): no communicator; the all-reduce is played by adding the
        shards' exchange buffers through torch views -- same packing, same update kernels.
        engine_factory: what builds a shard's engine (default DeviceCAVI; the CPU test of this class
        passes a stand-in with the same surface and hands the finished object to fit(engine=...))."""
        from concurrent.futures import ThreadPoolExecutor
        from .engine import DeviceCAVI
        make_engine = engine_factory or DeviceCAVI
        if not hasattr(X, "row"):
            X = X.tocoo()
        self.devices = list(devices)
        self.world = len(self.devices)
        if self.world < 2:
            raise ValueError("ThreadedShards needs at least two devices")
        self.dtype = np.dtype(dtype)
        self.ncells, self.ngenes, self.nfactors = X.shape[0], X.shape[1], int(nfactors)
        self.nnz = int(X.data.shape[0])
        self.bounds = row_partition(X, self.world)
        self._pool = ThreadPoolExecutor(max_workers=self.world)
        self.keep = [None] * self.world          # positions of each shard's nonzeros in X (for init_phi_host)
        self.comm = comm
        uid = DeviceCAVI.comm_unique_id() if comm == "rccl" else None

        # Two phases.  comm_init is a collective: a rank that raised before reaching it (an upload
        # validation error, out of memory, an empty shard) would leave the others blocked inside
        # ncclCommInitRank for ever.  So every shard is built and uploaded first, the errors are
        # gathered, and the communicator is only joined when all shards exist.
        def build(rank):
            lo, hi = int(self.bounds[rank]), int(self.bounds[rank + 1])
            if hi <= lo:
                raise ValueError("shard %d of %d would hold no cells (%d cells for %d devices)"
                                 % (rank, self.world, self.ncells, self.world))
            sub, keep = take_rows(X, lo, hi)
            self.keep[rank] = keep
            eng = make_engine(hi - lo, self.ngenes, self.nfactors, dtype=self.dtype, device=self.devices[rank])
            try:
                eng.hint_sharded()
                if "warn" in inspect.signature(eng.upload).parameters:
                    eng.upload(sub, warn=False)    # a pool thread must not touch the warnings machinery
                else:                              # stand-in engines of the CPU tests take no `warn`
                    eng.upload(sub)
            except BaseException:
                eng.close()
                raise
            return eng

        def attempt(fn):
            def run(rank):
                try:
                    return fn(rank), None
                except BaseException as exc:
                    return None, exc
            return run
        self.engines = []
        try:
            built = self._each(attempt(build))
            self.engines = [e for e, _ in built if e is not None]
            errors = [exc for _, exc in built if exc is not None]
            if errors:
                raise errors[0]
            if uid is not None:   # collective: all threads arrive, every shard exists
                joined = self._each(attempt(lambda r: self.engines[r].comm_init(uid, r, self.world)))
                errors = [exc for _, exc in joined if exc is not None]
                if errors:
                    raise errors[0]
        except BaseException:
            self.close()
            raise
        if self.engines and hasattr(self.engines[0], "upload_info"):
            # values rounded to float32 in ANY shard, reported once and on the calling thread; the facts are read by the
            # shard's own pool thread (every library call makes the context's device the thread's current one: read
            # from here, the caller's current HIP device would be left at the last shard's)
            infos = self._each(lambda r: self.engines[r].upload_info())
            rounded = sum(i["rounded"] for i in infos)
            if rounded:
                import warnings
                warnings.warn("%d of %d values of X.data are not exactly representable in float32 and were "
                              "rounded (relative error <= 6e-8); counts are stored as float32 on the device"
                              % (rounded, sum(i["nnz"] for i in infos)), RuntimeWarning, stacklevel=3)
        if comm != "rccl":
            self._views = [e.exchange if hasattr(e, "exchange") else exchange_tensor_of(e, d)
                           for e, d in zip(self.engines, self.devices)]

    def _each(self, fn):
        return list(self._pool.map(fn, range(self.world)))

    def _rows(self, rank):
        return slice(int(self.bounds[rank]), int(self.bounds[rank + 1]))

    # ---- the DeviceCAVI interface used by scHPF._fit
    def set_hypers(self, a, c, bp, dp):
        self._each(lambda r: self.engines[r].set_hypers(a, c, bp, dp))

    def set_gamma(self, name, vi_shape, vi_rate):
        if name in ("xi", "theta"):
            self._each(lambda r: self.engines[r].set_gamma(name, vi_shape[self._rows(r)], vi_rate[self._rows(r)]))
        else:
            self._each(lambda r: self.engines[r].set_gamma(name, vi_shape, vi_rate))

    def get_gamma(self, name):
        if name in ("eta", "beta"):             # identical on every rank
            return self.engines[0].get_gamma(name)
        parts = self._each(lambda r: self.engines[r].get_gamma(name))
        return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])

    def init_phi_host(self, Xphi_data):
        self._each(lambda r: self.engines[r].init_phi_host(Xphi_data[self.keep[r]]))

    def init_phi_device(self, seed):
        # the device generator is keyed by (seed, LOCAL cell, gene) and every shard numbers its cells
        # from 0: one seed for all would give local cell i of every shard the same responsibilities for
        # a gene.  Each rank gets its own stream of the generator (seed mixed with the rank).
        def rank_seed(r):
            return (int(seed) + 0x9E3779B97F4A7C15 * (r + 1)) & (2 ** 64 - 1)
        self._each(lambda r: self.engines[r].init_phi_device(rank_seed(r)))

    def steps(self, n, freeze_genes=False, simultaneous=False, cells_first=False):
        if cells_first:
            raise ValueError("the minibatch order is not available for sharded fits")
        if self.comm == "rccl":
            self._each(lambda r: self.engines[r].steps_sharded(n, freeze_genes=freeze_genes,
                                                               simultaneous=simultaneous))
            return
        for _ in range(n):                       # emulated all-reduce (tests): same protocol, summed here
            if not freeze_genes:
                for e in self.engines:
                    e.step_local(simultaneous=simultaneous, side="gene")
                    e.synchronize()
                total = self._views[0].clone()
                for v in self._views[1:]:
                    total += v.to(total.device)
                for v in self._views:
                    v.copy_(total.to(v.device))
                if total.is_cuda:
                    import torch
                    torch.cuda.synchronize()
                for e in self.engines:
                    e.step_local(simultaneous=simultaneous, side="cell")
            else:
                for e in self.engines:
                    e.step_local(freeze_genes=True, simultaneous=simultaneous)
            for e in self.engines:
                e.step_finish(freeze_genes=freeze_genes, simultaneous=simultaneous)

    def mean_negative_pois_llh(self):
        terms = self._each(lambda r: self.engines[r].loss_terms())   # same process: sum on the host
        llh = sum(t[0] for t in terms); gl = sum(t[1] for t in terms); nnz = sum(t[2] for t in terms)
        return mean_negative(llh, gl, nnz)

    def close(self):
        engines, self.engines = getattr(self, "engines", []), []
        for e in engines:
            e.close()
        self._pool.shutdown(wait=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class NativeShard(object):
    """One rank of a multi-process sharded fit (one process per GPU, e.g. under
    torch.distributed.run) with the collective inside the library: same role as ShardedCAVI, one
    library call per stretch of iterations instead of three calls and a torch collective per
    iteration.  `unique_id` is DeviceCAVI.comm_unique_id() of rank 0, delivered by the caller."""

    def __init__(self, engine, unique_id, rank, world):
        self.engine = engine
        engine.comm_init(unique_id, rank, world)

    def step(self, freeze_genes=False, simultaneous=False):
        self.engine.steps_sharded(1, freeze_genes=freeze_genes, simultaneous=simultaneous)

    def steps(self, n, freeze_genes=False, simultaneous=False):
        self.engine.steps_sharded(n, freeze_genes=freeze_genes, simultaneous=simultaneous)

    def mean_negative_pois_llh(self):
        llh, gl, nnz = self.engine.loss_terms_all()
        return mean_negative(llh, gl, nnz)
