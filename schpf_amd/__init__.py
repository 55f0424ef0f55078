"""schpf_amd -- MI355X-native engine for the scHPF CAVI hot path.

Host side: the reference's Python API (scHPF estimator, HPF_Gamma, loss functions,
the hpf_numba operator names) re-implemented over a C-ABI HIP library
(include/schpf_hip.h, schpf_amd/csrc/).  There is no CPU compute path.
"""
from ._version import __version__
from . import hpf_hip, loss, preprocessing
from .engine import DeviceCAVI
from .scHPF_ import HPF_Gamma, scHPF, load_model, save_model, combine_across_cells
from .trials import run_trials, run_trials_pool

# make the pickle module path of the classes (schpf.scHPF_) resolvable
import schpf.scHPF_  # noqa: E402,F401

__all__ = ["__version__", "hpf_hip", "loss", "preprocessing", "DeviceCAVI", "HPF_Gamma", "scHPF", "load_model",
           "save_model", "combine_across_cells", "run_trials", "run_trials_pool"]
