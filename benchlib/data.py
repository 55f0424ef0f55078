"""Synthetic count matrices of the benchmark: generator A (the reference's fixture recipe) whole,
slab by slab and per rank, and generator B (planted Gamma-Poisson factors) for the convergence fits."""
import os

import numpy as np
from scipy.sparse import coo_matrix


def synthetic_block(ncells, ngenes, density, seed):
    """Generator A of SURVEY.md 8(d) = the reference's test-fixture recipe
    (tests/conftest.py:14-25): negative-binomial counts at uniform positions, dups summed.
    Same draws in the same order as the fixture; the duplicates are summed by sorting packed
    (row, col, count) keys and adding up runs -- entry for entry what coo_matrix.sum_duplicates
    returns (canonical row-major order), in less than half the time at 1e8 draws (its lexsort)."""
    rng = np.random.RandomState(seed)
    nnz = int(round(ncells * ngenes * density))
    x = rng.negative_binomial(2, 0.5, nnz)
    x[x == 0] = 1
    if nnz == 0 or int(x.max()) > 255 or ncells * ngenes >= 2 ** 54:   # the count must fit 8 key bits
        row = rng.randint(0, ncells, nnz).astype(np.int32)
        col = rng.randint(0, ngenes, nnz).astype(np.int32)
        X = coo_matrix((x.astype(np.int32), (row, col)), shape=(ncells, ngenes), dtype=np.int32)
        X.sum_duplicates()
        return X
    bits = max(1, int(ngenes - 1).bit_length())
    key = rng.randint(0, ncells, nnz).astype(np.int64)
    key <<= bits
    key |= rng.randint(0, ngenes, nnz)
    key <<= 8
    key |= x
    del x
    key.sort()
    pos = key >> 8
    first = np.empty(nnz, dtype=bool)
    first[:1] = True
    np.not_equal(pos[1:], pos[:-1], out=first[1:])
    idx = np.flatnonzero(first)
    counts = np.add.reduceat(key & 255, idx).astype(np.int32)
    pos = pos[idx]
    X = coo_matrix((counts, ((pos >> bits).astype(np.int32), (pos & ((1 << bits) - 1)).astype(np.int32))),
                   shape=(ncells, ngenes), dtype=np.int32)
    X.has_canonical_format = True
    return X


SLAB_ROWS = 25000       # cells per slab of the slab generator (one RandomState(seed + slab) each)
SLAB_CONFIGS = {"c5": SLAB_ROWS, "c5-small": 1250}     # configs drawn slab by slab (40 slabs each): ranks draw their own rows


def _slab_draw(ncells, ngenes, density, seed, slab_rows, i):
    """Slab i of generator A drawn slab by slab: rows [i * slab_rows, ...), canonical (row-major, unique),
    duplicates summed by sorting packed (row, col, count) keys and adding up runs.  Returns (row, col, count, draws)."""
    bits = max(1, int(ngenes - 1).bit_length())
    r0 = i * slab_rows
    nr = min(slab_rows, ncells - r0)
    rng = np.random.RandomState(seed + i)
    n = int(round(nr * ngenes * density))
    x = rng.negative_binomial(2, 0.5, n)
    x[x == 0] = 1
    np.minimum(x, 255, out=x)                      # P(count > 255) is 2^-250; keeps the count in 8 key bits
    key = rng.randint(0, nr, n).astype(np.int64)
    key <<= bits
    key |= rng.randint(0, ngenes, n)
    key <<= 8
    key |= x
    del x
    key.sort()
    pos = key >> 8
    first = np.empty(n, dtype=bool)
    first[:1] = True
    np.not_equal(pos[1:], pos[:-1], out=first[1:])
    idx = np.flatnonzero(first)
    counts = np.add.reduceat(key & 255, idx).astype(np.int32) if n else np.zeros(0, np.int32)
    pos = pos[idx]
    return (pos >> bits).astype(np.int32) + np.int32(r0), (pos & ((1 << bits) - 1)).astype(np.int32), counts, n


def _draw_slabs(ncells, ngenes, density, seed, slab_rows, which, threads=None):
    from concurrent.futures import ThreadPoolExecutor
    which = list(which)
    workers = threads or max(1, min(len(which), len(os.sched_getaffinity(0)), 32))
    if not which:
        return {}
    with ThreadPoolExecutor(max_workers=workers) as pool:
        parts = list(pool.map(lambda i: _slab_draw(ncells, ngenes, density, seed, slab_rows, i), which))
    return dict(zip(which, parts))


def _coo_of_parts(parts, shape, row_offset=0):
    order = sorted(parts)
    cat = lambda j, dt: (np.concatenate([parts[i][j] for i in order]) if order else np.zeros(0, dt))   # noqa: E731
    row = cat(0, np.int32)
    if row_offset:
        row = row - np.int32(row_offset)
    X = coo_matrix((cat(2, np.int32), (row, cat(1, np.int32))), shape=shape, dtype=np.int32)
    X.has_canonical_format = True
    return X


def synthetic_slabs(ncells, ngenes, density, seed, slab_rows=SLAB_ROWS, threads=None):
    """Generator A for matrices of several 1e8 draws (all of C5: 5e8): the same recipe drawn slab by
    slab of `slab_rows` cells, one RandomState(seed + slab) and one thread per slab (NumPy releases the
    GIL in the draws and in sort) -- coo_matrix.sum_duplicates lexsorts 5e8 entries on one core for minutes.  The
    result is canonical (row-major, unique) and does not depend on the number of threads."""
    n_slabs = (ncells + slab_rows - 1) // slab_rows
    return _coo_of_parts(_draw_slabs(ncells, ngenes, density, seed, slab_rows, range(n_slabs), threads),
                         (ncells, ngenes))


def synthetic_slabs_of_rank(ncells, ngenes, density, seed, world, rank, all_reduce, slab_rows=SLAB_ROWS, threads=None):
    """Rank `rank`'s block of synthetic_slabs(...) under the product's nnz-balanced row partition WITHOUT any rank
    drawing the whole matrix ("generate per-shard on each GPU's host slice").  Pass 1: rank r draws the
    r-th of `world` contiguous runs of slabs and contributes their per-row nonzero counts, row sums and column sums; `all_reduce` (a
    sum over the ranks of a NumPy array) makes them global: the partition (schpf_amd.sharded.row_partition_from_counts)
    and the marginals the empirical hyperparameters need.  Pass 2: the rank draws the slabs that overlap its rows and
    that it does not hold yet (the partition is balanced by nonzeros, the runs by rows: a slab or two at the ends),
    and drops the others.  Per rank: 1/world of the draws plus a few boundary slabs -- not the whole matrix.
    Returns (X_local, bounds, facts) with facts = {nnz_total, row_sums, col_sums, slabs_drawn, slabs_total, draws}."""
    from schpf_amd.sharded import row_partition_from_counts
    n_slabs = (ncells + slab_rows - 1) // slab_rows
    mine = list(range(n_slabs * rank // world, n_slabs * (rank + 1) // world))   # contiguous: mostly the rank's own rows
    parts = _draw_slabs(ncells, ngenes, density, seed, slab_rows, mine, threads)
    drawn, draws = set(mine), sum(p[3] for p in parts.values())
    row_nnz = np.zeros(ncells, dtype=np.int64)
    row_sum = np.zeros(ncells, dtype=np.float64)
    col_sum = np.zeros(ngenes, dtype=np.float64)
    for r, c, v, _ in parts.values():
        row_nnz += np.bincount(r, minlength=ncells)
        row_sum += np.bincount(r, weights=v, minlength=ncells)
        col_sum += np.bincount(c, weights=v, minlength=ngenes)
    row_nnz, row_sum, col_sum = all_reduce(row_nnz), all_reduce(row_sum), all_reduce(col_sum)
    bounds = row_partition_from_counts(row_nnz, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    need = [i for i in range(n_slabs) if i * slab_rows < hi and min((i + 1) * slab_rows, ncells) > lo] if hi > lo else []
    for i in list(parts):
        if i not in need:
            del parts[i]
    more = _draw_slabs(ncells, ngenes, density, seed, slab_rows, [i for i in need if i not in parts], threads)
    drawn |= set(more)
    draws += sum(p[3] for p in more.values())
    parts.update(more)
    for i in list(parts):                       # the two boundary slabs: keep the rank's rows only
        r, c, v, n = parts[i]
        if r.size and (r[0] < lo or r[-1] >= hi):
            keep = (r >= lo) & (r < hi)
            parts[i] = (r[keep], c[keep], v[keep], n)
    X = _coo_of_parts(parts, (hi - lo, ngenes), row_offset=lo)
    facts = {"nnz_total": int(row_nnz.sum()), "row_sums": row_sum, "col_sums": col_sum, "slabs_drawn": len(drawn),
             "slabs_total": n_slabs, "draws": int(draws),
             "draws_whole_matrix": int(sum(int(round(min(slab_rows, ncells - i * slab_rows) * ngenes * density))
                                           for i in range(n_slabs)))}
    return X, bounds, facts


def planted_block(ncells, ngenes, K, target_events, seed):
    """Generator B of SURVEY.md 8(d): counts from a planted Gamma-Poisson factor model, so that
    the reference's stop rule has something to converge to.  x_ig ~ Poisson(sum_k theta_ik
    beta_gk) is sampled factor by factor: the events of factor k are Poisson(S_theta_k *
    S_beta_k) many, each landing on cell i with probability theta_ik / S_theta_k and gene g with
    probability beta_gk / S_beta_k (independent because the rate factorises)."""
    rng = np.random.RandomState(seed)
    theta = rng.gamma(0.3, 1.0, (ncells, K)) * rng.gamma(2.0, 0.5, (ncells, 1))
    beta = rng.gamma(0.3, 1.0, (ngenes, K)) * rng.gamma(2.0, 0.5, (ngenes, 1))
    st, sb = theta.sum(0), beta.sum(0)
    scale = target_events / float((st * sb).sum())
    # one independent stream per factor so that the factors can be drawn by a thread pool (NumPy
    # releases the GIL in random_sample / searchsorted) and the matrix does not depend on the pool
    counts_k = rng.poisson(st * sb * scale)
    seeds = rng.randint(0, 2 ** 31 - 1, K)

    def draw(k):
        r = np.random.RandomState(seeds[k])
        n_k = int(counts_k[k])
        rr = np.searchsorted(np.cumsum(theta[:, k]) / st[k], r.random_sample(n_k)).astype(np.int32)
        cc = np.searchsorted(np.cumsum(beta[:, k]) / sb[k], r.random_sample(n_k)).astype(np.int32)
        return rr, cc

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, min(K, (os.cpu_count() or 1)))) as pool:
        drawn = list(pool.map(draw, range(K)))
    rows = [d[0] for d in drawn]
    cols = [d[1] for d in drawn]
    del drawn
    row = np.minimum(np.concatenate(rows), ncells - 1)
    col = np.minimum(np.concatenate(cols), ngenes - 1)
    # events -> counts: sort (row, col) keys and count runs (what coo_matrix.sum_duplicates does
    # through a lexsort, several times slower at 1.6e8 events); the result is canonical row-major
    bits = max(1, int(ngenes - 1).bit_length())
    key = (row.astype(np.int64) << bits) | col
    del row, col
    key.sort()
    first = np.empty(key.shape[0], dtype=bool)
    first[:1] = True
    np.not_equal(key[1:], key[:-1], out=first[1:])
    idx = np.flatnonzero(first)
    del first
    counts = np.diff(idx, append=key.shape[0]).astype(np.int32)
    key = key[idx]
    X = coo_matrix((counts, ((key >> bits).astype(np.int32), (key & ((1 << bits) - 1)).astype(np.int32))),
                   shape=(ncells, ngenes), dtype=np.int32)
    X.has_canonical_format = True
    return X
