"""pytest configuration: markers, paths, shared fixtures."""
import functools
import os
import sys

import numpy as np
import pytest
from scipy.sparse import coo_matrix

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def golden_coo(g):
    shape = tuple(int(v) for v in g["shape"])
    return coo_matrix((g["x"], (g["row"], g["col"])), shape=shape)


def synthetic_counts(ncells, ngenes, frac, seed=42):
    """The reference's fixture recipe (tests/conftest.py:14-25 of the reference):
    negative-binomial counts at uniformly random positions, duplicates summed."""
    rng = np.random.RandomState(seed)
    nnz = int(ncells * ngenes * frac)
    x = rng.negative_binomial(2, 0.5, nnz)
    x[x == 0] = 1
    ci = rng.randint(0, ncells, nnz).astype(np.int32)
    gi = rng.randint(0, ngenes, nnz).astype(np.int32)
    X = coo_matrix((x, (ci, gi)), (ncells, ngenes), dtype=np.int32)
    X.sum_duplicates()
    return X


@functools.lru_cache(maxsize=2)
def bench_matrix(N, G, dens):
    """The matrices bench.py times (generator A of SURVEY.md 8(d), seed 42), drawn once per session."""
    if N * G * dens > 2e8:      # all of C5: bench.py's threaded slab generator (5e8 draws in well under a minute)
        from bench import synthetic_slabs
        return synthetic_slabs(N, G, dens, seed=42)
    from bench import synthetic_block               # bench.py's generator A, same seed: entry for entry
    return synthetic_block(N, G, dens, seed=42)     # synthetic_counts(...) (tests/test_bench_host.py), faster


@pytest.fixture(scope="session")
def oracle():
    from oracle import hpf_oracle
    hpf_oracle.build()
    return hpf_oracle


@pytest.fixture(params=["f64", "f32"])
def ops(request):
    return load_golden("ops_%s.npz" % request.param)
