"""Device-resident CAVI engine: the state of scHPF._fit's loop kept in HBM.

One `DeviceCAVI` = one GPU = the count matrix (both sweep plans), the four
variational Gammas and their derived tables.  `step()` is the body of the loop at
/root/reference/schpf/scHPF_.py:657-714; `mean_negative_pois_llh()` is the default
loss (schpf/loss.py:142-168).  Nothing here computes on the CPU: it is a thin
object around the C ABI in include/schpf_hip.h.
"""
import ctypes

import numpy as np

from . import _lib

_NAMES = {"xi": _lib.XI, "theta": _lib.THETA, "eta": _lib.ETA, "beta": _lib.BETA}
_VAL_KINDS = {np.dtype(np.int32): _lib.VAL_I32, np.dtype(np.int64): _lib.VAL_I64,
              np.dtype(np.float32): _lib.VAL_F32, np.dtype(np.float64): _lib.VAL_F64}


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def mean_negative(llh, gl, nnz):
    """-(sum llh) / nnz.  A matrix without stored entries has no mean: NaN, as the reference's np.mean over an empty
    array (loss.py:167) gives -- not a ZeroDivisionError."""
    return -(llh - gl) / nnz if nnz else float("nan")


class DeviceCAVI(object):
    """CAVI state on one MI355X.

    Parameters
    ----------
    ncells, ngenes, nfactors : int
        ncells is the number of local cells when cells are sharded over GPUs.
    dtype : np.float64 or np.float32
        model precision (scHPF(dtype=...), scHPF_.py:239).
    device : int
        HIP device ordinal.
    stream : int or None
        a hipStream_t handle (e.g. torch.cuda.current_stream().cuda_stream) to
        enqueue on; None lets the library create its own stream.  0 -- what torch
        reports for its default stream -- means the device's null stream, so that
        work torch enqueues on its current stream (collectives) is ordered with the
        engine's kernels.
    """

    def __init__(self, ncells, ngenes, nfactors, dtype=np.float64, device=0, stream=None):
        self._lib = _lib.load()
        _lib.require_gpu()
        self.dtype = np.dtype(dtype)
        if self.dtype == np.float64:
            code = _lib.F64
        elif self.dtype == np.float32:
            code = _lib.F32
        else:
            raise TypeError("dtype must be float64 or float32")
        self.ncells, self.ngenes, self.nfactors = int(ncells), int(ngenes), int(nfactors)
        self.nnz = 0
        handle = ctypes.c_void_p()
        if stream is None:
            stream_arg = None
        else:
            stream_arg = int(stream) or _lib.STREAM_DEFAULT
        _lib.check(self._lib.schpf_create(ctypes.byref(handle), int(device),
                                          ctypes.c_void_p(stream_arg), code,
                                          self.ncells, self.ngenes, self.nfactors))
        self._h = handle

    # ------------------------------------------------------------------ lifetime
    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.schpf_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -------------------------------------------------------------------- inputs
    def upload(self, X, warn=True):
        """X: scipy coo_matrix-like with .row, .col, .data (ncells x ngenes).  warn=False: the caller
        issues rounding_warning() itself (a helper thread must not touch the warnings machinery)."""
        if tuple(X.shape) != (self.ncells, self.ngenes):
            raise ValueError("X has shape %s, engine was created for %s"
                             % (tuple(X.shape), (self.ncells, self.ngenes)))
        data = np.ascontiguousarray(X.data)
        if data.dtype not in _VAL_KINDS:
            data = data.astype(np.float64)
        row = np.ascontiguousarray(X.row, dtype=np.int32)
        col = np.ascontiguousarray(X.col, dtype=np.int32)
        _lib.check(self._lib.schpf_upload_coo(self._h, data.shape[0], _p(row), _p(col), _p(data),
                                              _VAL_KINDS[data.dtype]))
        self.nnz = int(data.shape[0])
        if warn:
            self.rounding_warning(stacklevel=3)

    def rounding_warning(self, stacklevel=2):
        """Warn (on the calling thread) if the last upload rounded values of X.data to float32."""
        info = self.upload_info()
        if info["rounded"]:
            import warnings
            warnings.warn("%d of %d values of X.data are not exactly representable in float32 and were "
                          "rounded (relative error <= 6e-8); counts are stored as float32 on the device"
                          % (info["rounded"], info["nnz"]), RuntimeWarning, stacklevel=stacklevel)

    def keep_rows(self, on=True):
        """Call before upload(): also keep a row-sorted copy of the matrix in HBM, from which batch
        engines gather their rows (upload_rows) -- minibatch CAVI without host slicing or re-uploads."""
        _lib.check(self._lib.schpf_keep_rows(self._h, int(bool(on))))

    def upload_rows(self, source, rows):
        """This engine's matrix := rows `rows` (in that order) of `source`'s matrix (a DeviceCAVI on the
        same device that was told to keep_rows()); len(rows) must be this engine's ncells."""
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        _lib.check(self._lib.schpf_upload_rows(self._h, source._h, _p(rows), int(rows.shape[0])))
        self.nnz = self.upload_info()["nnz"]

    def upload_info(self):
        """{'nnz', 'rounded' (values rounded to float32), 'zeros' (explicitly stored), 'packed', 'rows'
        (a row-sorted copy is resident: keep_rows() took effect -- device-built tile plans only)}."""
        info = (ctypes.c_int64 * 4)()
        _lib.check(self._lib.schpf_upload_info(self._h, info))
        out = dict(zip(("nnz", "rounded", "zeros"), [int(v) for v in info[:3]]))
        out["packed"], out["rows"] = int(info[3]) & 1, bool(int(info[3]) & 2)
        return out

    def set_hypers(self, a, c, bp, dp):
        _lib.check(self._lib.schpf_set_hypers(self._h, float(a), float(c), float(bp), float(dp)))

    def _dims(self, name):
        n = self.ncells if name in ("xi", "theta") else self.ngenes
        return (n, self.nfactors) if name in ("theta", "beta") else (n,)

    def set_gamma(self, name, vi_shape, vi_rate):
        dims = self._dims(name)
        s = np.ascontiguousarray(vi_shape, dtype=self.dtype)
        r = np.ascontiguousarray(vi_rate, dtype=self.dtype)
        if s.shape != dims or r.shape != dims:
            raise ValueError("%s must have shape %s, got %s / %s" % (name, dims, s.shape, r.shape))
        _lib.check(self._lib.schpf_set_state(self._h, _NAMES[name], _p(s), _p(r)))

    def get_gamma(self, name):
        dims = self._dims(name)
        s = np.empty(dims, dtype=self.dtype)
        r = np.empty(dims, dtype=self.dtype)
        _lib.check(self._lib.schpf_get_state(self._h, _NAMES[name], _p(s), _p(r)))
        return s, r

    # ----------------------------------------------------------------- iteration
    def init_phi_host(self, Xphi_data):
        """t == 0 responsibilities drawn by the caller (scHPF_.py:652-655)."""
        x = np.ascontiguousarray(Xphi_data, dtype=np.float64)
        if x.shape != (self.nnz, self.nfactors):
            raise ValueError("Xphi_data must be (nnz, nfactors)")
        _lib.check(self._lib.schpf_init_phi_host(self._h, _p(x)))

    def init_phi_device(self, seed):
        _lib.check(self._lib.schpf_init_phi_device(self._h, ctypes.c_uint64(int(seed) & (2 ** 64 - 1))))

    @staticmethod
    def _flags(freeze_genes, simultaneous, sharded=False, cells_first=False):
        return ((_lib.FREEZE_GENES if freeze_genes else 0) | (_lib.SIMULTANEOUS if simultaneous else 0)
                | (_lib.SHARDED if sharded else 0) | (_lib.CELLS_FIRST if cells_first else 0))

    def step(self, freeze_genes=False, simultaneous=False, cells_first=False):
        """One CAVI iteration.  cells_first=True is the reference's minibatch order
        (scHPF_.py:688-704): cell block from the current beta, then gene block from the new theta."""
        _lib.check(self._lib.schpf_step(self._h, self._flags(freeze_genes, simultaneous,
                                                             cells_first=cells_first)))

    def steps(self, n, freeze_genes=False, simultaneous=False, cells_first=False):
        """n CAVI iterations in one call (the stretch between two loss checks); replayed as one
        hipGraph from the second call with the same arguments."""
        _lib.check(self._lib.schpf_steps(self._h, self._flags(freeze_genes, simultaneous,
                                                              cells_first=cells_first), int(n)))

    def step_local(self, freeze_genes=False, simultaneous=False, side=None):
        """Sweeps of the sharded iteration.  side=None: both; 'gene': the gene-side sweep and the
        packing of its sums into the exchange buffer; 'cell': the cell-side sweep (so that the
        caller can overlap it with the all-reduce of the gene-side sums)."""
        extra = {None: 0, "gene": _lib.LOCAL_GENE, "cell": _lib.LOCAL_CELL}[side]
        _lib.check(self._lib.schpf_step_local(self._h, self._flags(freeze_genes, simultaneous, True) | extra))

    def step_finish(self, freeze_genes=False, simultaneous=False):
        _lib.check(self._lib.schpf_step_finish(self._h, self._flags(freeze_genes, simultaneous, True)))

    # ------------------------------------------------- sharded, collective inside the library
    @staticmethod
    def comm_unique_id():
        """128 bytes identifying a new RCCL communicator: generate on ONE rank, give to all."""
        buf = ctypes.create_string_buffer(128)
        _lib.check(_lib.load().schpf_comm_unique_id(buf))
        return buf.raw

    def hint_sharded(self, on=True):
        """Call before upload() when the engine will run sharded iterations (two sweep launches)."""
        _lib.check(self._lib.schpf_hint_sharded(self._h, int(bool(on))))

    def hint_transient(self, on=True):
        """Call before upload() when the engine's matrix will be replaced every iteration (minibatches sliced on the
        host): its plans are then built the cheapest way (no balancing pass)."""
        _lib.check(self._lib.schpf_hint_transient(self._h, int(bool(on))))

    def comm_init(self, unique_id, rank, world):
        """Join the communicator (collective: returns once all `world` ranks have called it)."""
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of comm_unique_id()")
        try:
            _lib.check(self._lib.schpf_comm_init(self._h, ctypes.c_char_p(bytes(unique_id)), int(rank), int(world)))
        except _lib.SchpfHipError as e:
            raise _lib.SchpfHipError(str(e) + _lib.ipc_hint())
        self.comm_world = int(world)

    def comm_destroy(self):
        """Leave the communicator (captured stretches that hold its all-reduce are dropped first)."""
        _lib.check(self._lib.schpf_comm_destroy(self._h))
        self.comm_world = 0

    def steps_sharded(self, n, freeze_genes=False, simultaneous=False):
        """n iterations of the sharded protocol with the all-reduce issued by the library."""
        _lib.check(self._lib.schpf_steps_sharded(self._h, self._flags(freeze_genes, simultaneous), int(n)))

    def loss_terms_all(self):
        llh, gl, nnz = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        _lib.check(self._lib.schpf_loss_terms_all(self._h, ctypes.byref(llh), ctypes.byref(gl), ctypes.byref(nnz)))
        return llh.value, gl.value, nnz.value

    def exchange_buffer(self):
        """(device pointer, element count) of the buffer to all-reduce between
        step_local and step_finish; elements have the model dtype."""
        ptr, count = ctypes.c_void_p(), ctypes.c_int64()
        _lib.check(self._lib.schpf_exchange_buffer(self._h, ctypes.byref(ptr), ctypes.byref(count)))
        return ptr.value, count.value

    def loss_terms(self):
        """(sum x log r - r, sum lgamma(x+1), nnz) over the local nonzeros."""
        llh, gl, nnz = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        _lib.check(self._lib.schpf_loss_terms(self._h, ctypes.byref(llh), ctypes.byref(gl), ctypes.byref(nnz)))
        return llh.value, gl.value, nnz.value

    def mean_negative_pois_llh(self):
        llh, gl, nnz = self.loss_terms()
        return mean_negative(llh, gl, nnz)

    def synchronize(self):
        _lib.check(self._lib.schpf_synchronize(self._h))

    def stream_handle(self):
        """The hipStream_t (as an int; 0 = null stream) the engine enqueues on."""
        h = ctypes.c_void_p()
        _lib.check(self._lib.schpf_stream_handle(self._h, ctypes.byref(h)))
        return h.value or 0

    # ----------------------------------------------------------------- reporting
    def profile(self, enable=True):
        _lib.check(self._lib.schpf_profile_enable(self._h, int(bool(enable))))

    def profile_read(self):
        ms = (ctypes.c_double * 4)()
        n = (ctypes.c_int64 * 4)()
        _lib.check(self._lib.schpf_profile_read(self._h, ms, n))
        keys = ("cell_sweep", "gene_sweep", "loss_sweep", "gamma_updates")
        return {k: {"ms": ms[i], "launches": n[i]} for i, k in enumerate(keys)}

    def profile_clock(self):
        """(MHz, launches): the shader clock the device sustained under the sweep launches since the last call."""
        mhz, n = ctypes.c_double(), ctypes.c_int64()
        _lib.check(self._lib.schpf_profile_clock(self._h, ctypes.byref(mhz), ctypes.byref(n)))
        return mhz.value, n.value

    def sweep_bytes(self):
        """Bytes one iteration moves through the LDS / streams from HBM, from the tile plans (zeros otherwise)."""
        info = (ctypes.c_int64 * 8)()
        _lib.check(self._lib.schpf_sweep_bytes(self._h, info))
        keys = ("lds_read_nonzeros", "lds_read_stored_slots", "lds_staged_cell", "lds_staged_gene",
                "hbm_entry_stream", "partial_rows", "loss_side", "loss_tasks")
        return dict(zip(keys, [int(v) for v in info]))

    def plan_info(self):
        info = (ctypes.c_int64 * 16)()
        _lib.check(self._lib.schpf_plan_info(self._h, info))
        keys = ("KP", "KL", "LPC", "chunk_len", "windows_cell", "windows_gene", "n_chunks_cell",
                "n_chunks_gene", "n_waves_cell", "n_waves_gene", "entry_slots_cell", "entry_slots_gene",
                "ring_cell", "ring_gene", "ring_slot_bytes", "waves_per_block")
        return dict(zip(keys, [int(v) for v in info]))
