"""The one helper of the reference's schpf/util.py that the CAVI loop uses."""
import numpy as np

__all__ = ["minibatch_ix_generator", "mean_cellscore_fraction", "mean_cellscore_fraction_list",
           "max_pairwise", "max_pairwise_table"]


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
