"""Import-name alias so that this build is a drop-in for `import schpf`.

joblib model files written by the reference pickle `schpf.scHPF_.scHPF` and
`schpf.scHPF_.HPF_Gamma`; this package makes those paths resolve to the
MI355X-native implementation in schpf_amd (and files written here load in the
reference).  Nothing is implemented in this package.
"""
from schpf_amd.scHPF_ import *  # noqa: F401,F403
from schpf_amd.trials import run_trials, run_trials_pool  # noqa: F401
from schpf_amd.util import *  # noqa: F401,F403
from schpf_amd._version import __version__  # noqa: F401
from schpf_amd import loss, hpf_hip  # noqa: F401
from . import scHPF_  # noqa: F401
