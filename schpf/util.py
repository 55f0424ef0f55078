"""Alias of schpf_amd.util under the reference's module path."""
from schpf_amd.util import *  # noqa: F401,F403
