"""Host-side helpers of the reference's schpf/util.py: the minibatch index stream the CAVI loop
uses, the factor-quality summaries `scHPF score` writes, and the COO row split / collapse / insert
helpers around the validation-cell split (util.py:116-215)."""
import numpy as np
from scipy.sparse import coo_matrix

__all__ = ["minibatch_ix_generator", "mean_cellscore_fraction", "mean_cellscore_fraction_list",
           "max_pairwise", "max_pairwise_table", "split_coo_rows", "collapse_coo_rows", "insert_coo_rows"]


def minibatch_ix_generator(ncells, batchsize):
    """Endless stream of cell-index batches: one shuffle of 0..ncells-1 (NumPy global RNG, drawn
    at the first `next`), walked cyclically in strides of `batchsize`; a stride that runs off the
    end wraps around to the front (reference util.py:218-231)."""
    assert ncells >= batchsize
    order = np.arange(ncells)
    np.random.shuffle(order)
    start = 0
    while True:
        end = start + batchsize
        if end > ncells:
            end %= ncells
            batch = np.hstack([order[start:], order[:end]])
        else:
            batch = order[start:end]
        start = end % ncells
        yield batch


def mean_cellscore_fraction(cell_scores, ntop_factors=1):
    """Mean over cells of the share of a cell's total score held by its `ntop_factors`
    highest-scoring factors (reference util.py:11-33; `scHPF score` writes the curve)."""
    top = np.partition(cell_scores, cell_scores.shape[1] - ntop_factors, axis=1)[:, -ntop_factors:]
    return float(np.mean(top.sum(axis=1) / cell_scores.sum(axis=1)))


def mean_cellscore_fraction_list(cell_scores):
    """mean_cellscore_fraction for ntop_factors = 1 .. nfactors (reference util.py:36-41)."""
    ordered = np.sort(cell_scores, axis=1)[:, ::-1]
    shares = np.cumsum(ordered, axis=1) / cell_scores.sum(axis=1, keepdims=True)
    return [float(v) for v in shares.mean(axis=0)]


def max_pairwise(gene_scores, ntop=200, second_greatest=False):
    """Largest (or second largest) overlap between the `ntop` top genes of any two factors and
    its hypergeometric tail probability P(overlap >= observed) (reference util.py:44-85).
    Returns (overlap, p)."""
    from collections import namedtuple
    from scipy.stats import hypergeom
    ngenes, nfactors = gene_scores.shape
    member = np.zeros((ngenes, nfactors), dtype=np.int32)
    tops = np.argsort(gene_scores, axis=0)[-ntop:]
    member[tops, np.arange(nfactors)[None, :]] = 1
    overlaps = (member.T @ member)[np.triu_indices(nfactors, k=1)]
    # the reference walks the pairs in order and keeps a running (max, runner-up): the runner-up only
    # moves when a pair beats it, so equal maxima leave it behind -- reproduced
    best = last = 0
    for o in overlaps:
        if o > best:
            best, last = int(o), best
        elif o > last:
            last = int(o)
    overlap = last if second_greatest else best
    p = float(hypergeom.pmf(k=overlap, M=ngenes, N=ntop, n=ntop) + hypergeom.sf(k=overlap, M=ngenes, N=ntop, n=ntop))
    return namedtuple("Overlap", ["overlap", "p"])(overlap, p)


def max_pairwise_table(gene_scores, ntop_list=(50, 100, 150, 200, 250, 300)):
    """DataFrame of max_pairwise for several `ntop` (reference util.py:88-113)."""
    import pandas as pd
    first = [max_pairwise(gene_scores, n, False) for n in ntop_list]
    second = [max_pairwise(gene_scores, n, True) for n in ntop_list]
    return pd.DataFrame({"ntop": list(ntop_list), "max_overlap": [o.overlap for o in first],
                         "p_max": [o.p for o in first], "max2_overlap": [o.overlap for o in second],
                         "p_max2": [o.p for o in second]})


def split_coo_rows(X, split_indices):
    """(a, b): the rows of X listed in `split_indices` (in that order), and the others in their
    original order, both as COO (reference util.py:116-140)."""
    rest = np.setdiff1d(np.arange(X.shape[0]), split_indices)
    csr = X.tocsr()
    return csr[split_indices, :].tocoo(), csr[rest, :].tocoo()


def collapse_coo_rows(coo):
    """X without its empty rows, and the original indices of the rows kept (reference
    util.py:143-160)."""
    kept = np.flatnonzero(coo.getnnz(1) > 0)
    return coo.tocsr()[kept].tocoo(), kept


def insert_coo_rows(a, b, b_indices):
    """The (a.rows + b.rows) x cols COO matrix whose rows `b_indices` (ascending, unique) are b's
    rows and whose other rows are a's, each set in its own order -- the inverse of split_coo_rows
    (reference util.py:163-215; same errors).  Built from the two index maps instead of a Python
    loop over rows."""
    if a.shape[1] != b.shape[1]:
        raise ValueError("a.shape[1] must equal b.shape[1], received a with shape {} and b with shape {}".format(
            a.shape, b.shape))
    n_out = a.shape[0] + b.shape[0]
    b_indices = np.asarray(b_indices)
    if np.max(b_indices) >= n_out:
        raise ValueError("Invalid row indices {} for array with a.shape[0] + b.shape[0] = {} + {} = {}".format(
            b_indices, a.shape[0], b.shape[0], n_out))
    if not np.all(np.diff(b_indices) > 0):
        raise ValueError("`b_indices` must be ordered without repeats. Received {}".format(b_indices))
    a, b = a.tocsr().tocoo(), b.tocsr().tocoo()                 # duplicates summed, row-major
    a_rows = np.setdiff1d(np.arange(n_out), b_indices)          # where a's rows land
    row = np.concatenate([a_rows[a.row], b_indices[b.row]])
    order = np.argsort(row, kind="stable")
    return coo_matrix((np.concatenate([a.data, b.data])[order], (row[order], np.concatenate([a.col, b.col])[order])),
                      shape=(n_out, a.shape[1]))
