"""ctypes binding of libschpf_hip.so (the C ABI declared in include/schpf_hip.h).

There is no CPU fallback: if the shared library is missing or cannot be loaded the
import of anything that computes raises, loudly.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SCHPF_LIB_PATH") or os.path.join(_HERE, "libschpf_hip.so")   # override: A/B of two builds

F32, F64 = 0, 1
XI, THETA, ETA, BETA = 0, 1, 2, 3
VAL_I32, VAL_I64, VAL_F32, VAL_F64 = 0, 1, 2, 3
STREAM_DEFAULT = 1   # SCHPF_STREAM_DEFAULT: the device's null stream
FREEZE_GENES, SIMULTANEOUS, SHARDED, CELLS_FIRST, LOCAL_GENE, LOCAL_CELL = 1, 2, 4, 8, 16, 32

_vp = ctypes.c_void_p
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64 = ctypes.c_int64
_int = ctypes.c_int
_dbl = ctypes.c_double
_dblp = ctypes.POINTER(ctypes.c_double)
_i64p = ctypes.POINTER(ctypes.c_int64)

# name -> argtypes; every function returns int status (0 = ok)
SIGNATURES = {
    "schpf_device_count": [ctypes.POINTER(_int)],
    "schpf_digamma": [_i64, _vp, _vp],
    "schpf_gammaln": [_i64, _vp, _vp],
    "schpf_xphi": [_int, _i64, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "schpf_pois_llh_pointwise": [_int, _i64, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "schpf_shape_update": [_int, _i64, _int, _vp, _vp, _int, _dbl, _vp],
    "schpf_rate_update": [_int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp],
    "schpf_capacity_rate_update": [_int, _int, _int, _vp, _vp, _dbl, _vp],
    "schpf_create": [ctypes.POINTER(_vp), _int, _vp, _int, _int, _int, _int],
    "schpf_destroy": [_vp],
    "schpf_upload_coo": [_vp, _i64, _vp, _vp, _vp, _int],
    "schpf_set_hypers": [_vp, _dbl, _dbl, _dbl, _dbl],
    "schpf_set_state": [_vp, _int, _vp, _vp],
    "schpf_get_state": [_vp, _int, _vp, _vp],
    "schpf_init_phi_host": [_vp, _vp],
    "schpf_init_phi_device": [_vp, ctypes.c_uint64],
    "schpf_step": [_vp, ctypes.c_uint],
    "schpf_steps": [_vp, ctypes.c_uint, _int],
    "schpf_step_local": [_vp, ctypes.c_uint],
    "schpf_exchange_buffer": [_vp, ctypes.POINTER(_vp), _i64p],
    "schpf_step_finish": [_vp, ctypes.c_uint],
    "schpf_loss_terms": [_vp, _dblp, _dblp, _i64p],
    "schpf_synchronize": [_vp],
    "schpf_hint_sharded": [_vp, _int],
    "schpf_hint_transient": [_vp, _int],
    "schpf_keep_rows": [_vp, _int],
    "schpf_upload_rows": [_vp, _vp, _vp, _int],
    "schpf_comm_unique_id": [_vp],
    "schpf_comm_init": [_vp, _vp, _int, _int],
    "schpf_comm_destroy": [_vp],
    "schpf_steps_sharded": [_vp, ctypes.c_uint, _int],
    "schpf_loss_terms_all": [_vp, _dblp, _dblp, _i64p],
    "schpf_stream_handle": [_vp, ctypes.POINTER(_vp)],
    "schpf_profile_enable": [_vp, _int],
    "schpf_profile_read": [_vp, _dblp, _i64p],
    "schpf_profile_clock": [_vp, _dblp, _i64p],
    "schpf_sweep_bytes": [_vp, _i64p],
    "schpf_plan_info": [_vp, _i64p],
    "schpf_upload_info": [_vp, _i64p],
    "schpf_coo_marginals": [_i64, _vp, _vp, _vp, _int, _int, _int, _vp, _vp],
    "schpf_debug_plan_expand": [_i64, _vp, _vp, _vp, _int, _int, _int, _int, _int,
                                _vp, _vp, _vp, _vp, _vp, _vp, _i64p],
    "schpf_debug_tile_expand": [_i64, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _int,
                                _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64p],
}
STRING_FUNCS = ("schpf_last_error", "schpf_version")

_lib = None
HIP_RUNTIME = None
IPC_NOTE = ""   # set by load() when dmabuf IPC may not be in effect (see there)


ERR_NO_MEMORY = 2      # include/schpf_hip.h SCHPF_ERR_NO_MEMORY


class SchpfHipError(RuntimeError):
    """A failed library call; `status` is the C function's return value."""

    def __init__(self, msg, status=1):
        RuntimeError.__init__(self, msg)
        self.status = status


def is_out_of_memory(exc):
    """True if a library error is "no memory" -- hipErrorOutOfMemory on the device or std::bad_alloc in the host-side
    plan builders, which the C ABI reports with a status of its own (SCHPF_ERR_NO_MEMORY) -- the only failure the
    callers' "does not fit, try a smaller layout" fallbacks are meant for."""
    return getattr(exc, "status", None) == ERR_NO_MEMORY


def build(force=False):
    """Compile libschpf_hip.so for gfx950 with hipcc (schpf_amd/csrc/Makefile)."""
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc, "-s", "-j4"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return LIB_PATH


def _hip_runtime_path():
    """The ONE HIP runtime this process should use.

    PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so (torch/lib);
    loading a second copy from /opt/rocm next to it gives two HSA runtimes in one process:
    whichever initialises second may see no GPU, and memory allocated by one is unknown to the
    other (RCCL on our exchange buffer).  libschpf_hip.so therefore has no DT_NEEDED on the
    runtime; we pick it here: $SCHPF_HIP_RUNTIME, else torch's bundled copy when torch is
    installed (found without importing torch), else the system ROCm.
    """
    override = os.environ.get("SCHPF_HIP_RUNTIME")
    if override:
        return override
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is not None and spec.origin:
            cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
            if os.path.exists(cand):
                return cand
    except (ImportError, ValueError):
        pass
    for cand in ("/opt/rocm/lib/libamdhip64.so", "libamdhip64.so"):
        if cand.startswith("/") and not os.path.exists(cand):
            continue
        return cand
    return "libamdhip64.so"


def load():
    global _lib, HIP_RUNTIME
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SchpfHipError(
            "libschpf_hip.so not found at %s: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    # Multi-process / multi-GPU work (RCCL, sharing device memory between processes) needs dmabuf IPC
    # on this driver stack: without HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment BEFORE the HSA runtime
    # initialises, RCCL fails with `hipIpcGetMemHandle: invalid argument`.  Set here, ahead of the dlopen
    # below, so that fit(devices=[...]), run_trials_pool(devices=...), the CLI and bench.py all get it;
    # a value the caller exported wins.
    # It has to be set here and not on the multi-device paths only: the first HIP call of ANY engine initialises
    # the runtime, and a later fit(devices=[...]) in the same process could no longer change it.  What this
    # cannot fix is a runtime that was initialised before this module was loaded (torch.cuda touched first
    # without the variable): IPC_NOTE records that, and comm_init failures quote it (ipc_hint()).
    global IPC_NOTE
    if os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") is None:
        if _hsa_runtime_mapped():
            IPC_NOTE = ("the HSA runtime was already loaded in this process without HSA_ENABLE_IPC_MODE_LEGACY=0 "
                        "(e.g. torch.cuda was used before schpf_amd was imported): export it before starting Python")
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    elif os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] != "0":
        IPC_NOTE = ("HSA_ENABLE_IPC_MODE_LEGACY=%s is exported; multi-GPU work on this driver stack needs 0"
                    % os.environ["HSA_ENABLE_IPC_MODE_LEGACY"])
    HIP_RUNTIME = _hip_runtime_path()
    # RCCL follows the HIP runtime: the copy next to it (torch/lib or /opt/rocm/lib) unless overridden
    if not os.environ.get("SCHPF_RCCL_PATH"):
        cand = os.path.join(os.path.dirname(HIP_RUNTIME), "librccl.so")
        if os.path.exists(cand):
            os.environ["SCHPF_RCCL_PATH"] = cand
    try:
        ctypes.CDLL(HIP_RUNTIME, mode=ctypes.RTLD_GLOBAL)
    except OSError as e:
        raise SchpfHipError("cannot load the HIP runtime %s: %s" % (HIP_RUNTIME, e))
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    for name in STRING_FUNCS:
        getattr(lib, name).restype = ctypes.c_char_p
        getattr(lib, name).argtypes = []
    _lib = lib
    return lib


def _hsa_runtime_mapped():
    try:
        with open("/proc/self/maps") as f:
            return "libhsa-runtime64" in f.read()
    except OSError:
        return False


def ipc_hint():
    """Appended to communicator errors: why device memory may not be shareable between ranks."""
    return (" [" + IPC_NOTE + "]") if IPC_NOTE else ""


def check(status):
    if status != 0:
        msg = load().schpf_last_error().decode("utf-8", "replace")
        if status != ERR_NO_MEMORY and ("must be" in msg or "out of range" in msg or "unknown" in msg):
            raise ValueError(msg)
        raise SchpfHipError(msg, status)


def device_count():
    n = _int(0)
    check(load().schpf_device_count(ctypes.byref(n)))
    return n.value


def require_gpu():
    if device_count() < 1:
        raise SchpfHipError("no HIP device visible: schpf_amd computes only on an AMD GPU "
                            "(MI355X / gfx950); there is no CPU fallback")
