"""Count-matrix loaders: the data formats on the input side of the hot path.

Host-only mirrors of the reference's loaders (/root/reference/schpf/preprocessing.py:11-135
and the `.mtx` branch of its command line, bin/scHPF:370-374): same names, arguments, return
types, dtypes and COO entry order, so a matrix loaded here and handed to `scHPF.fit` is the
matrix the reference would have trained on.  Parsing is vectorised (pandas' C tokenizer) instead
of per-token Python loops; the gene filtering / `prep` pipeline is out of scope.
"""
import numpy as np
from scipy.sparse import coo_matrix

__all__ = ["load_coo", "load_txt", "load_loom", "load_mtx", "load_counts"]


def load_coo(filename):
    """Tab-separated `cell <TAB> gene <TAB> count` triples, 0-indexed, no header
    (preprocessing.py:11-29).  Shape is (max cell + 1, max gene + 1); values are int64."""
    import pandas as pd
    raw = pd.read_csv(filename, sep="\t", header=None, dtype=np.int64, comment="#").values
    if raw.ndim != 2 or raw.shape[1] < 3:
        raise ValueError("%s: expected three tab-separated integer columns" % filename)
    return coo_matrix((raw[:, 2], (raw[:, 0], raw[:, 1])))


def load_mtx(filename):
    """Matrix Market file, cells x genes (what the reference's CLI reads with scipy's mmread)."""
    from scipy.io import mmread
    return coo_matrix(mmread(filename))


def load_txt(filename, ngene_cols=2, verbose=True):
    """Whitespace-delimited genes x cells text matrix without header whose first `ngene_cols`
    columns are gene attributes (preprocessing.py:67-135).

    Returns (coo, genes): the cells x genes int32 COO matrix and an ngenes x ngene_cols
    DataFrame of the attribute columns (strings).  Entry order follows the reference: gene-major
    for plain files (it appends gene by gene), cell-major for .gz/.bz2 files (it transposes a
    dense array first).
    """
    import pandas as pd
    assert ngene_cols > 0
    compressed = filename.endswith(".gz") or filename.endswith(".bz2")
    if compressed:
        print(".....WARNING: Input file {} is compressed. It may be faster to manually "
              "decompress before loading.".format(filename))
    df = pd.read_csv(filename, header=None, sep=r"\s+", dtype={c: str for c in range(ngene_cols)})
    genes = df[list(range(ngene_cols))]
    counts = df.drop(columns=list(range(ngene_cols))).values        # genes x cells
    if not np.issubdtype(counts.dtype, np.integer):
        as_int = counts.astype(np.int64)
        if not np.array_equal(as_int, counts):
            raise ValueError("%s: counts must be integers" % filename)
        counts = as_int
    ngenes, ncells = counts.shape
    if compressed:
        dense = counts.T
        nz = np.nonzero(dense)
        coo = coo_matrix((dense[nz], nz), shape=dense.shape, dtype=np.int32)
    else:
        g, cell = np.nonzero(counts)                                 # gene-major, cells ascending
        coo = coo_matrix((counts[g, cell], (cell, g)), shape=(ncells, ngenes), dtype=np.int32)
        genes = pd.DataFrame(genes.values.tolist())
    if verbose and ngenes >= 10000:
        print("\tloaded {} genes for {} cells".format(ngenes, ncells))
    return coo, genes


def load_loom(filename):
    """Loom file -> (cells x genes COO, gene-attribute DataFrame with Accession and Gene first)
    (preprocessing.py:32-64).  Needs the optional `loompy` package."""
    import pandas as pd
    import loompy
    with loompy.connect(filename) as ds:
        genes = pd.DataFrame(dict(ds.ra.items()))
        coo = ds.sparse().T
    first = [c for c in ("Accession", "Gene") if c in genes.columns]
    return coo, genes[first + genes.columns.difference(first).tolist()]


def load_counts(filename):
    """Dispatch on the extension the way the reference's `train` command does
    (bin/scHPF:370-374): `.mtx` -> Matrix Market, anything else -> COO triples."""
    return load_mtx(filename) if filename.endswith(".mtx") else load_coo(filename)
