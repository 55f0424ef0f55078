"""Alias of schpf_amd.preprocessing (loaders and the prep pipeline) under the reference's module path."""
from schpf_amd.preprocessing import *  # noqa: F401,F403
