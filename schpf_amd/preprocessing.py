"""Count-matrix loaders: the data formats on the input side of the hot path.

Host-only mirrors of the reference's loaders (/root/reference/schpf/preprocessing.py:11-135
and the `.mtx` branch of its command line, bin/scHPF:370-374): same names, arguments, return
types, dtypes and COO entry order, so a matrix loaded here and handed to `scHPF.fit` is the
matrix the reference would have trained on.  Parsing is vectorised (pandas' C tokenizer) instead
of per-token Python loops.  The second half of the file is the `prep` / `prep-like` pipeline
(preprocessing.py:138-495): gene masks, the validation-cell split and the two composite loaders
the command line calls -- host work on either side of the hot path.
"""
import warnings

import numpy as np
from scipy.sparse import coo_matrix, issparse

__all__ = ["load_coo", "load_txt", "load_loom", "load_mtx", "load_counts",
           "min_cells_expressing_mask", "genelist_mask", "subsample_cell_ixs",
           "split_validation_cells", "load_and_filter", "load_like"]


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


# ------------------------------------------------------------------ prep / prep-like pipeline

def min_cells_expressing_mask(counts, min_cells, verbose=True):
    """Genes observed in at least `min_cells` cells (preprocessing.py:138-166).

    counts: cells x genes sparse matrix (or dense array).  `min_cells` strictly between 0 and 1
    is a proportion: the threshold becomes round(min_cells * ncells) and, as in the reference,
    that is printed whatever `verbose` says.  Returns a boolean array over genes."""
    if 0 < min_cells < 1:
        frac = min_cells
        min_cells = round(frac * counts.shape[0])
        print(".....requiring {}% of cells = {} cells observed expressing for gene inclusion".format(
            100 * frac, min_cells))
    if issparse(counts):
        coo = counts.tocoo()
        # one per stored nonzero value, like the reference's astype(bool).sum(axis=0)
        n_expressing = np.bincount(coo.col[coo.data != 0], minlength=coo.shape[1])
    else:
        n_expressing = np.count_nonzero(np.asarray(counts), axis=0)
    return n_expressing >= min_cells


def _stem(names):
    """Gene identifiers without what follows the first '.' (ENSEMBL version suffixes)."""
    return names.str.split(".").str[0]


def genelist_mask(candidates, genelist, whitelist=True, split_on_dot=True):
    """Which `candidates` (pd.Series of gene ids or names) are on (`whitelist=True`) or off
    (`False`) `genelist` (pd.Series); identifiers are compared without their '.version' suffix
    unless `split_on_dot` is false (preprocessing.py:169-200).  Returns a boolean ndarray."""
    if split_on_dot:
        candidates, genelist = _stem(candidates), _stem(genelist)
    on_list = candidates.isin(genelist).values
    return on_list if whitelist else ~on_list


def subsample_cell_ixs(choices, nselect, group_ids=None, max_group_frac=0.5):
    """`nselect` cell indices drawn without replacement from `choices` (an index array, or an int n
    for arange(n)), sorted (preprocessing.py:203-269).

    With `group_ids` (one label per choice) the draw is spread about evenly over the groups, no
    group giving more than floor(size * max_group_frac) cells; if that cannot yield `nselect`
    cells a UserWarning says so and fewer are returned.  Draws come from the global
    `np.random` state by the same sequence of calls as the reference (one `choice`; or per round
    one `multinomial` for the remainders, then one `choice` per group that still has room), so a
    seeded run selects the same cells."""
    if isinstance(choices, (int, np.integer)):
        choices = np.arange(choices)
    if group_ids is None:
        return np.sort(np.random.choice(choices, nselect, replace=False))
    if len(group_ids) != len(choices):
        raise AssertionError("group_ids must have one label per choice")
    labels, sizes = np.unique(group_ids, return_counts=True)
    room = np.floor(sizes * max_group_frac).astype(int)     # what each group may still give
    picked, wanted = [], nselect
    while room.sum() > 0 and wanted > 0:
        open_groups = room > 0
        share = open_groups / open_groups.sum()
        quota = np.floor(share * wanted).astype(int)
        leftover = np.sum(np.ceil(share * wanted) - quota).astype(int)
        quota = quota + np.random.multinomial(leftover, share)
        for gi in range(len(labels)):
            if room[gi] <= 0:
                continue
            take = min(quota[gi], room[gi])
            pool = np.setdiff1d(choices[group_ids == labels[gi]], picked)
            picked.extend(list(np.random.choice(pool, take, replace=False)))
            room[gi] -= take
            wanted -= take
    if wanted > 0:
        warnings.warn("Could not select {} cells with given group_ids under constraint max_group_frac={}. "
                      "{} cells selected.".format(nselect, max_group_frac, wanted), UserWarning)
    return np.sort(picked)


def split_validation_cells(X, nselect, group_id_file="", max_group_frac=0.5, verbose=True):
    """Hold out `nselect` random cells of X, optionally balanced over the groups listed (one id per
    cell, `np.loadtxt`) in `group_id_file` (preprocessing.py:272-325).

    Returns (Xtrain, Xvalidation, validation_ix): X without / with only the selected rows, and
    the selected row indices (sorted)."""
    from .util import split_coo_rows
    group_ids = np.loadtxt(group_id_file) if group_id_file is not None and len(group_id_file) else None
    chosen = subsample_cell_ixs(X.shape[0], nselect, group_ids, max_group_frac)
    if verbose:
        msg = ".....{} cells selected".format(len(chosen))
        if group_ids is not None:
            msg += " ~~evenly from groups in {} under constraint max_group_frac={}".format(
                group_id_file, max_group_frac)
            msg += "\n\tGroup counts:"
            for gid, n in zip(*np.unique(group_ids[chosen], return_counts=True)):
                msg += "\n\t\t[{}] {}".format(gid, n)
        print(msg)
    Xvalidation, Xtrain = split_coo_rows(X, chosen)
    return Xtrain, Xvalidation, chosen


def _load_with_gene_names(infile, by_gene_name):
    """(umis, genes, name column of the gene lists, candidate names) for a loom or text input:
    loom files are matched by `Accession` when they have it, else by `Gene`
    (preprocessing.py:375-391, :460-476)."""
    if infile.endswith(".loom"):
        umis, genes = load_loom(infile)
        if "Accession" in genes.columns:
            return umis, genes, 0, genes["Accession"]
        if "Gene" in genes.columns:
            return umis, genes, 1, genes["Gene"]
        raise ValueError("loom files must have at least one of the row attributes: `Gene` or `Accession`.")
    umis, genes = load_txt(infile)
    col = 1 if by_gene_name else 0
    return umis, genes, col, genes[col]


def _read_genelist(path):
    import pandas as pd
    return pd.read_csv(path, sep=r"\s+", header=None)


def load_and_filter(infile, min_cells, whitelist="", blacklist="", filter_by_gene_name=False,
                    no_split_on_dot=False, verbose=True):
    """Load a genes x cells text matrix (or a loom file) and keep the genes that are expressed in
    `min_cells` cells, on `whitelist` and not on `blacklist` (two-column id / name files; the
    blacklist wins) -- the body of `scHPF prep` (preprocessing.py:328-415).

    Returns (filtered, genes): the cells x kept-genes COO matrix and the kept rows of the gene
    table.  Raises ValueError for a negative `min_cells` or a loom file without gene names.

    Reference quirk kept on purpose: it passes `split_on_dot = ~no_split_on_dot` to the list masks,
    and `~False == -1`, `~True == -2` are both truthy -- identifiers are ALWAYS compared without
    their '.version' suffix here, whatever `no_split_on_dot` says (preprocessing.py:401-406)."""
    if verbose:
        print("Loading data.....")
    umis, genes, list_col, names = _load_with_gene_names(infile, filter_by_gene_name)
    if verbose:
        print(".....found {} cells and {} genes".format(*umis.shape))
        print("Generating masks for filtering.....")
    if min_cells < 0:
        raise ValueError("min_cells must be >= 0")
    keep = min_cells_expressing_mask(umis, min_cells)
    if whitelist is not None and len(whitelist):
        keep &= genelist_mask(names, _read_genelist(whitelist)[list_col], split_on_dot=True)
    if blacklist is not None and len(blacklist):
        keep &= genelist_mask(names, _read_genelist(blacklist)[list_col], whitelist=False, split_on_dot=True)
    if verbose:
        print("Filtering data.....")
    # the reference slices a LIL copy; a CSR column selection gives the same row-major COO
    filtered = umis.tocsr()[:, np.flatnonzero(keep)].tocoo()
    return filtered, genes.loc[keep]


def load_like(infile, reference, by_gene_name=False, no_split_on_dot=False):
    """Load a text / loom matrix with exactly the genes of the two-column file `reference`, in its
    order -- the body of `scHPF prep-like` (preprocessing.py:418-495).  A gene listed twice in
    `infile` resolves to its first row, as in the reference.

    Returns (umis, genes).  Raises ValueError when a reference gene is missing from `infile`."""
    umis, genes, list_col, names = _load_with_gene_names(infile, by_gene_name)
    wanted = _read_genelist(reference)[list_col]
    if not no_split_on_dot:
        wanted, names = _stem(wanted), _stem(names)
    first_row = {}
    for i, g in enumerate(names.values):
        first_row.setdefault(g, i)
    perm = []
    for g in wanted.values:
        if g not in first_row:
            raise ValueError("Reference gene `{}` in reference `{}` not found in infile `{}`".format(
                g, reference, infile))
        perm.append(first_row[g])
    return umis.tocsr()[:, perm].tocoo(), genes.loc[perm]
