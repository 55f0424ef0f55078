"""Host-side pieces of bench.py that run without a GPU: the synthetic generator (generator A)
and the row split of the N > 1 bench -- every rank draws the N = 1 matrix and keeps its block of the product's
nnz-balanced partition."""
import numpy as np
import pytest
from scipy.sparse import coo_matrix

from conftest import synthetic_counts


@pytest.mark.parametrize("shape", [(300, 1000, 0.03, 42), (1000, 500, 0.05, 3), (7, 3, 0.5, 1), (2000, 1500, 0.08, 9)])
def test_bench_generator_equals_the_fixture_recipe(shape):
    """bench.synthetic_block sums duplicates by sorting packed keys; it must return, entry for entry, what the
    reference's fixture recipe (draws, then coo_matrix.sum_duplicates) returns."""
    from bench import synthetic_block
    n, g, dens, seed = shape
    A = synthetic_block(n, g, dens, seed)
    B = synthetic_counts(n, g, dens, seed=seed)
    assert A.shape == B.shape and A.nnz == B.nnz
    assert A.row.dtype == np.int32 and A.col.dtype == np.int32 and A.data.dtype == np.int32
    assert np.array_equal(A.row, B.row) and np.array_equal(A.col, B.col) and np.array_equal(A.data, B.data)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_rank_blocks_tile_the_unsharded_matrix(world):
    from bench import synthetic_block
    from schpf_amd.sharded import row_partition, take_rows
    X = synthetic_block(5000, 800, 0.04, 42)
    bounds = row_partition(X, world)
    assert bounds[0] == 0 and bounds[-1] == X.shape[0] and np.all(np.diff(bounds) >= 0)
    parts = [take_rows(X, int(bounds[r]), int(bounds[r + 1])) for r in range(world)]
    assert sum(p[0].nnz for p in parts) == X.nnz
    back = coo_matrix((np.concatenate([p[0].data for p in parts]),
                       (np.concatenate([p[0].row + int(bounds[r]) for r, p in enumerate(parts)]),
                        np.concatenate([p[0].col for p in parts]))), shape=X.shape)
    assert (back != X).nnz == 0
    nnz = np.array([p[0].nnz for p in parts])
    assert nnz.max() - nnz.min() <= 2 * np.bincount(X.row, minlength=X.shape[0]).max()
