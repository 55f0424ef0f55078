"""Import-only stand-in for the `numba` package (TEST TOOL, this container only).

numba is not installable here (no wheel, no network).  The reference package
(`/root/reference/schpf`) imports numba at module import time, so without this
it cannot be imported to generate golden vectors.  This module contains no
reference code: `njit` is the identity decorator (the decorated function bodies
then run as ordinary Python over NumPy scalars -- the reference's own
arithmetic, evaluated strictly left to right), `prange` is `range`.

Used only by tests/golden/make_golden.py.  Never imported by the product.
"""
__version__ = "standin"


def njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda fn: fn


jit = njit
prange = range
