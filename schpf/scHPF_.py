"""Alias of schpf_amd.scHPF_ under the reference's module path (pickle compatibility)."""
from schpf_amd.scHPF_ import *  # noqa: F401,F403
from schpf_amd.scHPF_ import HPF_Gamma, scHPF, load_model, save_model, combine_across_cells  # noqa: F401
from schpf_amd.trials import run_trials, run_trials_pool  # noqa: F401
