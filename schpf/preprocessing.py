"""Alias of schpf_amd.preprocessing (loaders only) under the reference's module path."""
from schpf_amd.preprocessing import *  # noqa: F401,F403
