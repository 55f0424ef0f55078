"""The special functions of the fused Gamma update (schpf_amd/csrc/special.h: psi, log, exp, reciprocal -- written for
the update kernel's instruction count) compiled for the HOST with g++ and checked against SciPy / NumPy: the header is
plain C++ over fma / frexp / ldexp, the only device-specific piece (the hardware reciprocal seed) is replaced by a seed
of the same 24-bit precision.  The same functions on the GPU are pinned by tests/test_ops_gpu.py (schpf_digamma) and by
every engine parity test."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
from scipy.special import digamma

from conftest import ROOT, load_golden

SRC = r"""
#include "special.h"
extern "C" {
void h_psi(long n, const double *x, double *o) { for (long i = 0; i < n; ++i) o[i] = schpf::digamma(x[i]); }
void h_psi_less_log(long n, const double *x, const double *rate, double *o)
{ for (long i = 0; i < n; ++i) o[i] = schpf::digamma_less_log(x[i], schpf::fast_rcp(rate[i])); }
void h_log(long n, const double *x, double *o) { for (long i = 0; i < n; ++i) o[i] = schpf::fast_log(x[i]); }
void h_exp(long n, const double *x, double *o) { for (long i = 0; i < n; ++i) o[i] = schpf::fast_exp(x[i]); }
void h_rcp(long n, const double *x, double *o) { for (long i = 0; i < n; ++i) o[i] = schpf::fast_rcp(x[i]); }
}
"""


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    d = tmp_path_factory.mktemp("special")
    src = d / "h.cpp"
    src.write_text(SRC)
    so = d / "libspecial_host.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-mfma", "-fPIC", "-shared",
                           "-I", os.path.join(ROOT, "schpf_amd", "csrc"), str(src), "-o", str(so)])
    lib = ctypes.CDLL(str(so))

    def call(name, *arrays):
        arrays = [np.ascontiguousarray(a, dtype=np.float64) for a in arrays]
        out = np.empty(arrays[0].shape[0])
        getattr(lib, name)(ctypes.c_long(out.shape[0]), *[a.ctypes.data_as(ctypes.c_void_p) for a in arrays],
                           out.ctypes.data_as(ctypes.c_void_p))
        return out
    return call


def _abs_or_rel(got, want):
    err = np.abs(got - want)
    return np.minimum(err, err / np.maximum(np.abs(want), 1e-300))


def test_psi_on_the_golden_grid_and_densely(host):
    """Within 4e-15 (relative or absolute) of SciPy's psi -- the bound tests/test_ops_gpu.py holds the device to -- on
    the committed 1 948-point grid and on 3e5 more points of [1e-4, 1e6]; the un-shifted branch above 1e8 too."""
    g = load_golden("psi_gammaln.npz")
    assert _abs_or_rel(host("h_psi", g["x"]), g["psi"]).max() <= 4e-15
    x = np.concatenate([np.logspace(-4, 6, 200001), np.linspace(0.3, 30.0, 100001),
                        [9.99e7, 1e8, 1.01e8, 1e12, 1e15]])
    assert _abs_or_rel(host("h_psi", x), digamma(x)).max() <= 4e-15


def test_psi_minus_log_rate(host):
    """E[log x] = psi(shape) - log(rate) (hpf_numba.py:83-94) with ONE logarithm and a shared reciprocal: as good as
    the difference of two rounded library values (whose own rounding is ~1 ulp of the larger term)."""
    rng = np.random.RandomState(0)
    shape = np.exp(rng.uniform(np.log(0.05), np.log(1e5), 300000))
    rate = np.exp(rng.uniform(-12, 12, 300000))
    got = host("h_psi_less_log", shape, rate)
    want = digamma(shape) - np.log(rate)
    scale = np.maximum(np.abs(digamma(shape)), np.abs(np.log(rate)))
    assert (np.abs(got - want) / np.maximum(scale, 1.0)).max() <= 4e-15


def test_log_exp_reciprocal(host):
    rng = np.random.RandomState(1)
    z = np.concatenate([np.exp(rng.uniform(-700, 700, 300000)), rng.uniform(0.5, 2.0, 300000),
                        [1.0, np.sqrt(0.5), np.nextafter(np.sqrt(0.5), 1.0), np.sqrt(2.0), 5e-324, 1e-310, 1.7e308]])
    want = np.log(z)
    err = np.abs(host("h_log", z) - want)
    assert (err / np.maximum(np.spacing(np.abs(want)), 5e-324)).max() <= 2.0 or err.max() == 0.0     # 2 ulp
    with np.errstate(all="ignore"):
        ends = host("h_log", np.array([0.0, np.inf, np.nan]))
    assert ends[0] == -np.inf and ends[1] == np.inf and np.isnan(ends[2])
    d = np.concatenate([-rng.uniform(0, 700, 300000), rng.uniform(-1, 1, 100000), [0.0]])
    got, want = host("h_exp", d), np.exp(d)
    assert (np.abs(got - want) / want).max() <= 4.5e-16                                               # 2 ulp
    tail = host("h_exp", np.array([-745.0, -746.0, -800.0, -1e5, -1e300]))
    assert np.array_equal(tail, np.exp(np.array([-745.0, -746.0, -800.0, -1e5, -1e300])))
    r = np.exp(rng.uniform(-600, 600, 200000))
    assert np.abs(host("h_rcp", r) * r - 1.0).max() <= 2.3e-16
