"""Loaders on the input side of the path (schpf_amd/preprocessing.py) against the COO the
reference's own loader produced from its own test file (tests/golden/make_golden.py)."""
import gzip
import os
import shutil

import numpy as np
from numpy.testing import assert_array_equal
from scipy.io import mmwrite
from scipy.sparse import coo_matrix

from conftest import GOLDEN, load_golden
from schpf_amd import preprocessing as prep

TXT = os.path.join(GOLDEN, "PJ030merge.c300t400_g0t500.matrix.txt")


def test_load_txt_matches_reference_loader():
    g = load_golden("pbmc_like_data.npz")
    X, genes = prep.load_txt(TXT, verbose=False)
    assert X.shape == tuple(g["shape"]) and X.dtype == np.int32
    assert_array_equal(X.row, g["row"])          # same entries in the same (gene-major) order
    assert_array_equal(X.col, g["col"])
    assert_array_equal(X.data, g["x"])
    assert genes.shape == (X.shape[1], 2)
    assert genes.iloc[0, 0] == "ENSG00000000003.14" and genes.iloc[0, 1] == "TSPAN6"


def test_load_txt_compressed_is_cell_major(tmp_path):
    path = str(tmp_path / "m.txt.gz")
    with open(TXT, "rb") as src, gzip.open(path, "wb") as dst:
        shutil.copyfileobj(src, dst)
    g = load_golden("pbmc_like_data.npz")
    X, genes = prep.load_txt(path, verbose=False)
    want = coo_matrix((g["x"], (g["row"], g["col"])), shape=tuple(g["shape"]))
    assert (X != want).nnz == 0
    assert np.all(np.diff(X.row) >= 0)
    assert genes.shape == (X.shape[1], 2)


def test_load_txt_gene_columns(tmp_path):
    path = str(tmp_path / "m.txt")
    with open(path, "w") as fh:
        fh.write("g1 0 3 0\ng2 1 0 2\n")
    X, genes = prep.load_txt(path, ngene_cols=1, verbose=False)
    assert X.shape == (3, 2)
    assert_array_equal(X.toarray(), [[0, 1], [3, 0], [0, 2]])
    assert genes.values.tolist() == [["g1"], ["g2"]]


def test_load_coo_and_mtx_roundtrip(tmp_path):
    g = load_golden("pbmc_like_data.npz")
    want = coo_matrix((g["x"], (g["row"], g["col"])), shape=tuple(g["shape"]))
    tsv = str(tmp_path / "m.tsv")
    np.savetxt(tsv, np.stack([want.row, want.col, want.data], 1), fmt="%d", delimiter="\t")
    X = prep.load_coo(tsv)
    assert X.dtype == np.int64
    assert_array_equal(X.row, want.row)
    assert_array_equal(X.col, want.col)
    assert_array_equal(X.data, want.data)
    # shape is inferred from the largest indices, as in the reference
    assert X.shape == (want.row.max() + 1, want.col.max() + 1)
    mtx = str(tmp_path / "m.mtx")
    mmwrite(mtx, want, field="integer")
    Y = prep.load_counts(mtx)
    assert Y.shape == want.shape and (Y != want).nnz == 0
    assert (prep.load_counts(tsv) != X).nnz == 0
