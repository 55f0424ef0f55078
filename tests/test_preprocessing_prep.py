"""The `prep` / `prep-like` pipeline against vectors produced by the reference itself
(tests/golden/prep_data.npz, make_golden.py::make_prep) and the reference's own test cases
(/root/reference/tests/test_preprocessing.py:91-195, tests/test_util.py:66-132)."""
import os

import numpy as np
import pandas as pd
import pytest
from numpy.testing import assert_array_equal, assert_equal
from scipy.sparse import coo_matrix

from schpf_amd import preprocessing as prep
from schpf_amd.util import collapse_coo_rows, insert_coo_rows, split_coo_rows

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TXT = os.path.join(GOLD, "PJ030merge.c300t400_g0t500.matrix.txt")
WL = os.path.join(GOLD, "prep_whitelist.txt")
BL = os.path.join(GOLD, "sample_blacklist.txt")
LIKE = os.path.join(GOLD, "prep_like_reference.txt")
GROUPS = os.path.join(GOLD, "prep_group_ids.txt")
NCELLS, NGENES = 100, 500


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "prep_data.npz"))


@pytest.fixture(scope="module")
def data():
    return prep.load_txt(TXT, verbose=False)


def same_coo(m, gold, key):
    """Entry-for-entry: the same COO arrays in the same order."""
    assert_array_equal(m.row, gold[key + "_row"])
    assert_array_equal(m.col, gold[key + "_col"])
    assert_array_equal(m.data, gold[key + "_data"])
    assert m.data.dtype == gold[key + "_data"].dtype


@pytest.mark.parametrize("tag,kw", [
    ("m2", dict(min_cells=2, whitelist=WL, blacklist=BL)),
    ("m5name", dict(min_cells=5, whitelist=WL, blacklist=BL, filter_by_gene_name=True)),
    ("frac_nosplit", dict(min_cells=0.05, whitelist=WL, no_split_on_dot=True)),
    ("m0", dict(min_cells=0))])
def test_load_and_filter_matches_reference(gold, tag, kw):
    f, g = prep.load_and_filter(TXT, verbose=False, **kw)
    assert_array_equal(np.array(f.shape), gold["laf_%s_shape" % tag])
    same_coo(f, gold, "laf_" + tag)
    assert_array_equal(g.values.astype(str), gold["laf_%s_genes" % tag])
    assert_array_equal(g.index.values, gold["laf_%s_index" % tag])


def test_load_and_filter_properties():
    """The reference's own assertions (tests/test_preprocessing.py:172-195)."""
    wl, bl = (pd.read_csv(p, sep=r"\s+", header=None) for p in (WL, BL))
    stem = lambda s: s.str.split(".").str[0]
    f2, g2 = prep.load_and_filter(TXT, min_cells=2, whitelist=WL, blacklist=BL, verbose=False)
    assert f2.shape[0] == NCELLS and f2.shape[1] <= NGENES and f2.shape[1] == len(g2)
    assert stem(g2[0]).isin(stem(bl[0])).sum() == 0
    assert stem(g2[0]).isin(stem(wl[0])).sum() == len(g2)
    f5, g5 = prep.load_and_filter(TXT, min_cells=5, whitelist=WL, blacklist=BL, verbose=False)
    assert f5.shape[1] <= f2.shape[1]
    assert np.all(np.asarray(f5.astype(bool).sum(axis=0)) >= 5)
    with pytest.raises(ValueError):
        prep.load_and_filter(TXT, min_cells=-1, verbose=False)


@pytest.mark.parametrize("tag,kw", [("id", {}), ("name", dict(by_gene_name=True)), ("nosplit", dict(no_split_on_dot=True))])
def test_load_like_matches_reference(gold, data, tag, kw):
    f, g = prep.load_like(TXT, reference=LIKE, **kw)
    same_coo(f, gold, "like_" + tag)
    assert_array_equal(g.index.values, gold["like_%s_index" % tag])
    umis, genes = data
    perm = gold["like_perm"]
    assert_array_equal(f.toarray(), umis.toarray()[:, perm])          # tests/test_preprocessing.py:52-88
    assert len(g) == len(perm)


def test_load_like_missing_gene(tmp_path, data):
    _umis, genes = data
    bad = genes.loc[[3, 1, 4]].copy()
    bad.loc[1, 0] = "random"
    ref = str(tmp_path / "genes.txt")
    bad.to_csv(ref, header=None, sep="\t", index=None)
    with pytest.raises(ValueError, match="not found in infile"):
        prep.load_like(TXT, reference=ref)
    prep.load_like(TXT, reference=ref, by_gene_name=True)             # the names still match


def test_min_cells_expressing(gold, data):
    umis, _ = data
    ncells, ngenes = umis.shape
    assert_array_equal(prep.min_cells_expressing_mask(umis, 3), gold["mask_min3"])
    assert_array_equal(prep.min_cells_expressing_mask(umis, 0.1), gold["mask_frac"])
    # the reference's cases (tests/test_preprocessing.py:91-116)
    assert prep.min_cells_expressing_mask(umis, 0).sum() == ngenes
    assert prep.min_cells_expressing_mask(umis, ngenes + 1).sum() == 0
    assert prep.min_cells_expressing_mask(umis, 0.9999999).sum() == 0
    want = np.asarray(umis.astype(bool).sum(axis=0))[0] >= 5
    assert_array_equal(prep.min_cells_expressing_mask(umis, 5), want)
    assert_array_equal(prep.min_cells_expressing_mask(umis, 5 / ncells), want)
    assert_array_equal(prep.min_cells_expressing_mask(umis.toarray(), 5), want)   # dense input too
    assert_array_equal(prep.min_cells_expressing_mask(umis.tocsr(), 5), want)


def test_genelist_mask(data):
    _umis, genes = data
    wl = pd.read_csv(WL, sep=r"\s+", header=None)
    stem = lambda s: s.str.split(".").str[0]
    shared_id = stem(genes[0]).isin(stem(wl[0])).values
    shared_name = genes[1].isin(wl[1]).values
    assert_array_equal(prep.genelist_mask(genes[0], wl[0]), shared_id)
    assert_array_equal(prep.genelist_mask(genes[1], wl[1]), shared_name)
    assert_array_equal(prep.genelist_mask(genes[0], wl[0], whitelist=False), ~shared_id)
    assert_array_equal(prep.genelist_mask(genes[1], wl[1], whitelist=False), ~shared_name)
    exact = genes[0].isin(wl[0]).values
    assert_array_equal(prep.genelist_mask(genes[0], wl[0], split_on_dot=False), exact)
    assert exact.sum() < shared_id.sum()                                # the whitelist has altered version suffixes


def test_subsample_cell_ixs_matches_reference_draws(gold):
    np.random.seed(11)
    assert_array_equal(prep.subsample_cell_ixs(100, 17), gold["sub_plain"])
    groups = gold["groups"]
    np.random.seed(12)
    assert_array_equal(prep.subsample_cell_ixs(100, 30, group_ids=groups, max_group_frac=0.4), gold["sub_groups"])
    np.random.seed(13)
    assert_array_equal(prep.subsample_cell_ixs(np.arange(50, 150), 20, group_ids=groups, max_group_frac=0.5),
                       gold["sub_choices"])


def test_subsample_cell_ixs_constraints():
    """tests/test_preprocessing.py:139-169."""
    assert len(prep.subsample_cell_ixs(20, 10)) == 10
    assert len(prep.subsample_cell_ixs(np.arange(20), 10)) == 10
    idx = prep.subsample_cell_ixs(102, 10, group_ids=np.array([0] * 100 + [1, 1]), max_group_frac=0.5)
    assert (100 in idx) ^ (101 in idx) and len(idx) == 10
    small = np.array([0] * 18 + [1, 1])
    idx = prep.subsample_cell_ixs(20, 5, group_ids=small, max_group_frac=0.4)
    assert 18 not in idx and 19 not in idx and len(idx) == 5
    with pytest.warns(UserWarning) as record:
        idx = prep.subsample_cell_ixs(20, 5, group_ids=small, max_group_frac=0.25)
    assert len(record) == 1
    assert 18 not in idx and 19 not in idx and len(idx) == 4          # floor(0.25 * 18)


def test_split_validation_cells_and_back(gold, data):
    umis, _ = data
    np.random.seed(14)
    Xt, Xv, vix = prep.split_validation_cells(umis, 20, GROUPS, max_group_frac=0.5, verbose=False)
    assert_array_equal(vix, gold["split_vix"])
    same_coo(Xt, gold, "split_train")
    same_coo(Xv, gold, "split_valid")
    assert_array_equal(np.array(Xt.shape), gold["split_train_shape"])
    back = insert_coo_rows(Xt, Xv, vix)
    same_coo(back, gold, "insert")
    assert_array_equal(back.toarray(), umis.toarray())
    np.random.seed(3)
    Xt, Xv, vix = prep.split_validation_cells(umis, 7, verbose=False)  # no groups
    assert Xv.shape == (7, NGENES) and Xt.shape == (NCELLS - 7, NGENES)


def test_collapse_coo_rows(gold, data):
    umis, _ = data
    col, nz = collapse_coo_rows(umis.T.tocoo())
    assert_array_equal(nz, gold["collapse_nz"])
    same_coo(col, gold, "collapse")
    a = coo_matrix((np.array([1, 2, 3, 4, 5, 6]), (np.array([0, 0, 2, 3, 3, 3]), np.array([0, 2, 2, 0, 1, 2]))))
    collapsed, kept = collapse_coo_rows(a)                               # tests/test_util.py:81-89
    assert collapsed.shape[0] == a.shape[0] - 1
    assert_array_equal(kept, [0, 2, 3])


def test_split_coo_rows():
    X = coo_matrix((np.array([1, 2, 3, 4, 5, 6]), (np.array([0, 0, 2, 3, 3, 3]), np.array([0, 2, 2, 0, 1, 2]))))
    a, b = split_coo_rows(X, np.array([0, 2, 3]))                        # tests/test_util.py:66-78
    assert a.shape == (3, 3) and b.shape == (1, 3)
    assert_array_equal(b.toarray()[0], X.toarray()[1])
    assert_array_equal(a.toarray(), X.toarray()[[0, 2, 3]])


def test_insert_coo_rows():
    """tests/test_util.py:92-132."""
    a = coo_matrix((np.array([1, 2, 3, 4, 5, 6]), (np.array([0, 0, 1, 2, 2, 2]), np.array([0, 2, 2, 0, 1, 2]))))
    b_row, b_col, b_data = np.array([0, 1, 1]), np.array([2, 1, 2]), np.array([11, 12, 13])
    b = coo_matrix((b_data, (b_row, b_col)))
    ab = insert_coo_rows(a, b, [0, 1])
    assert ab.shape[0] == a.shape[0] + b.shape[0]
    assert_array_equal(ab.toarray()[:2], b.toarray())
    assert_array_equal(ab.toarray()[2:], a.toarray())
    ab = insert_coo_rows(a, b, [1, 4])
    assert_array_equal(ab.toarray()[[1, 4]], b.toarray())
    assert_array_equal(ab.toarray()[[0, 2, 3]], a.toarray())
    with pytest.raises(ValueError, match=r"a.shape\[1\] must equal b.shape\[1\]"):
        insert_coo_rows(a, coo_matrix((b_data, (b_row, b_col)), shape=[3, 5]), [1, 4])
    with pytest.raises(ValueError, match="Invalid row indices"):
        insert_coo_rows(a, b, [1, 7])
    for bad in ([2, 1], [1, 1]):
        with pytest.raises(ValueError, match="must be ordered"):
            insert_coo_rows(a, b, bad)
