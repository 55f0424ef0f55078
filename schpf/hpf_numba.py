"""The reference's operator module name, served by the HIP implementation (schpf_amd.hpf_hip)."""
from schpf_amd.hpf_hip import *  # noqa: F401,F403
