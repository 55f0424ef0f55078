"""Alias of schpf_amd.loss under the reference's module path."""
from schpf_amd.loss import *  # noqa: F401,F403
