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


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_per_rank_draw_gives_the_blocks_of_the_whole_matrix(world):
    """bench.py --gpus N --config c5: no rank draws the whole matrix (synthetic_slabs_of_rank).  Ranks are played by
    threads, the all-reduce by a barrier: every rank ends up with exactly its block of the product's row partition
    of the whole-matrix draw, the global nnz and marginals, having drawn about 1 / world of it."""
    import threading
    from bench import synthetic_slabs, synthetic_slabs_of_rank
    from schpf_amd.sharded import row_partition, take_rows
    N, G, dens, slab = 20000, 3000, 0.02, 500
    X = synthetic_slabs(N, G, dens, seed=42, slab_rows=slab)
    bounds_want = row_partition(X, world)
    barrier = threading.Barrier(world)
    box, out, errors = {}, {}, []

    def all_reduce_of(rank):
        def all_reduce(a):
            box[rank] = a
            barrier.wait()
            total = sum(box[r] for r in range(world))
            barrier.wait()
            return total
        return all_reduce

    def run(rank):
        try:
            out[rank] = synthetic_slabs_of_rank(N, G, dens, 42, world, rank, all_reduce_of(rank), slab_rows=slab, threads=1)
        except Exception as e:      # noqa: BLE001
            errors.append(e)
            barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors
    for rank in range(world):
        Xl, bounds, facts = out[rank]
        want, _ = take_rows(X, int(bounds_want[rank]), int(bounds_want[rank + 1]))
        assert np.array_equal(bounds, bounds_want)
        assert Xl.shape == want.shape
        assert np.array_equal(Xl.row, want.row) and np.array_equal(Xl.col, want.col) and np.array_equal(Xl.data, want.data)
        assert facts["nnz_total"] == X.nnz
        assert np.array_equal(facts["row_sums"], np.asarray(X.sum(1)).ravel())
        assert np.array_equal(facts["col_sums"], np.asarray(X.sum(0)).ravel())
        # 1 / world of the draws and at most three boundary slabs more
        assert facts["draws"] <= facts["draws_whole_matrix"] * (1.0 / world + 3.0 / facts["slabs_total"]) + 1
