"""The `scHPF` command line (schpf_amd/cli.py): same sub-commands, options and output file names
as the reference's bin/scHPF for train / score / project."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, load_golden, golden_coo


def run_cli(*argv):
    from schpf_amd import cli
    return cli.main(list(argv))


def test_parser_has_the_reference_options():
    from schpf_amd import cli
    p = cli._parser()
    a = p.parse_args(["train", "-i", "x.mtx", "-k", "7", "-t", "3", "-M", "50", "-m", "5", "-e", "0.1", "-f", "5",
                      "-a", "-2", "--float32", "-bs", "100", "-sl", "4", "-bts", "-sa", "-rp", "--quiet"])
    assert (a.nfactors, a.ntrials, a.max_iter, a.min_iter, a.check_freq, a.batchsize, a.smooth_loss) == \
        (7, 3, 50, 5, 5, 100, 4)
    assert a.float32 and a.beta_theta_simultaneous and a.save_all and a.reproject and not a.verbose
    assert p.parse_args(["train-pool", "-i", "x", "-k", "5", "6", "7"]).nfactors == [5, 6, 7]
    b = p.parse_args(["project", "-m", "m.joblib", "-i", "x.mtx"])
    assert (b.max_iter, b.min_iter, b.check_freq, b.recalc_bp) == (500, 10, 10, False)     # bin/scHPF:276-289


def test_score_writes_the_reference_files(tmp_path):
    """`scHPF score` needs no GPU: scores of a model file written by the REFERENCE."""
    model = os.path.join(GOLDEN, "ref_model_f64.joblib")
    genes = tmp_path / "genes.txt"
    import joblib
    m = joblib.load(model)
    G = m.ngenes
    genes.write_text("".join("ENSG%05d\tgene%d\n" % (i, i) for i in range(G)))
    assert run_cli("score", "-m", model, "-o", str(tmp_path / "out"), "-p", "run1", "-g", str(genes)) == 0
    out = tmp_path / "out"
    cs = np.loadtxt(out / "run1.cell_score.txt", delimiter="\t")
    gs = np.loadtxt(out / "run1.gene_score.txt", delimiter="\t")
    np.testing.assert_allclose(cs, m.cell_score())
    np.testing.assert_allclose(gs, m.gene_score())
    ranked = np.loadtxt(out / "run1.ranked_genes.txt", dtype=str, delimiter="\t")
    assert ranked.shape == gs.shape and ranked[0, 0] == "gene%d" % int(np.argmax(gs[:, 0]))
    frac = np.loadtxt(out / "run1.mean_cellscore_fraction.txt", skiprows=1)
    assert frac.shape == (cs.shape[1], 2) and abs(frac[-1, 1] - 1.0) < 1e-12 and np.all(np.diff(frac[:, 1]) >= 0)
    table = (out / "run1.maximum_overlaps.txt").read_text().splitlines()
    assert table[0].split("\t") == ["ntop", "max_overlap", "p_max", "max2_overlap", "p_max2"] and len(table) == 11
    assert (out / "run1.score_commandline_args.json").exists()


def test_score_statistics_follow_their_definitions():
    from schpf_amd.util import max_pairwise, mean_cellscore_fraction, mean_cellscore_fraction_list
    rng = np.random.RandomState(0)
    cs = rng.gamma(1.0, 1.0, (50, 6))
    want = [np.mean(np.sort(cs, 1)[:, -n:].sum(1) / cs.sum(1)) for n in range(1, 7)]
    np.testing.assert_allclose(mean_cellscore_fraction_list(cs), want)
    np.testing.assert_allclose(mean_cellscore_fraction(cs, 2), want[1])
    gs = rng.gamma(1.0, 1.0, (300, 5))
    tops = np.argsort(gs, axis=0)[-40:]
    pair = sorted(len(np.intersect1d(tops[:, i], tops[:, j])) for i in range(5) for j in range(i + 1, 5))
    assert max_pairwise(gs, 40).overlap == pair[-1]
    assert max_pairwise(gs, 40, second_greatest=True).overlap <= pair[-1]
    assert 0.0 <= max_pairwise(gs, 40).p <= 1.0


@pytest.mark.gpu
def test_train_then_project_from_the_shell(tmp_path):
    """bin/scHPF train -> model file with the reference's name -> bin/scHPF project on it."""
    from scipy.io import mmwrite
    g = load_golden("fit_data_k5_s0_f64.npz")
    X = golden_coo(g)
    mtx = tmp_path / "train.mtx"
    mmwrite(str(mtx), X, field="integer")
    exe = [sys.executable, os.path.join(ROOT, "bin", "scHPF")]
    subprocess.check_call(exe + ["train", "-i", str(mtx), "-o", str(tmp_path / "m"), "-k", "5", "-t", "2", "-M", "30",
                                 "--quiet"])
    # the reference's name (bin/scHPF:446-449): "_b{batchsize}" is appended whenever ncells > batchsize,
    # i.e. also for batchsize 0 -- a quirk scripts downstream rely on
    model_file = tmp_path / "m" / "scHPF_K5_b0_2trials.joblib"
    assert model_file.exists() and (tmp_path / "m" / "train_commandline_args.json").exists()
    subprocess.check_call(exe + ["project", "-m", str(model_file), "-i", str(mtx), "--max-iter", "10"])
    proj = tmp_path / "m" / "scHPF_K5_b0_2trials.proj.joblib"
    assert proj.exists()
    import joblib
    model, projected = joblib.load(model_file), joblib.load(proj)
    assert projected.beta == model.beta and projected.theta.dims == model.theta.dims
    assert len(model.loss) >= 2 and model.loss[-1] < model.loss[0]


def test_prep_and_prep_like_write_the_reference_files(tmp_path):
    """`scHPF prep` / `prep-like` (bin/scHPF:317-366): the files the reference writes, holding what the
    reference's load_and_filter / split_validation_cells / load_like return (tests/golden/prep_data.npz)."""
    from scipy.io import mmread
    gold = np.load(os.path.join(GOLDEN, "prep_data.npz"))
    txt = os.path.join(GOLDEN, "PJ030merge.c300t400_g0t500.matrix.txt")
    out = tmp_path / "prepped"
    p = __import__("schpf_amd.cli", fromlist=["_parser"])._parser()
    a = p.parse_args(["prep", "-i", txt])
    assert (a.min_cells, a.whitelist, a.blacklist, a.n_validation_cells, a.validation_max_group_frac) == \
        (0.01, "", "", 0, 0.5)                                                             # bin/scHPF:58-88
    np.random.seed(14)
    assert run_cli("prep", "-i", txt, "-o", str(out), "-p", "pj", "-m", "0", "-nvc", "20", "-vgid",
                   os.path.join(GOLDEN, "prep_group_ids.txt")) == 0
    names = sorted(os.listdir(out))
    assert names == ["pj.filtered.mtx", "pj.genes.txt", "pj.prep_commandline_args.json", "pj.train_cell_ix.txt",
                     "pj.train_cells.mtx", "pj.validation_cell_ix.txt", "pj.validation_cells.mtx"]
    f = mmread(str(out / "pj.filtered.mtx"))
    assert np.array_equal(f.row, gold["laf_m0_row"]) and np.array_equal(f.col, gold["laf_m0_col"])
    assert np.array_equal(f.data, gold["laf_m0_data"]) and f.data.dtype.kind == "i"
    assert open(out / "pj.filtered.mtx").readline().split()[3] == "integer"
    vix = np.loadtxt(out / "pj.validation_cell_ix.txt", dtype=int)
    assert np.array_equal(vix, gold["split_vix"])
    assert np.array_equal(np.loadtxt(out / "pj.train_cell_ix.txt", dtype=int), np.setdiff1d(np.arange(100), vix))
    v = mmread(str(out / "pj.validation_cells.mtx"))
    assert np.array_equal(v.row, gold["split_valid_row"]) and np.array_equal(v.data, gold["split_valid_data"])
    t = mmread(str(out / "pj.train_cells.mtx"))
    assert np.array_equal(t.col, gold["split_train_col"]) and t.shape == tuple(gold["split_train_shape"])
    genes = np.loadtxt(out / "pj.genes.txt", dtype=str, delimiter="\t")
    assert np.array_equal(genes, gold["laf_m0_genes"])

    # filtered with lists; then a second data set prepared like the first
    assert run_cli("prep", "-i", txt, "-o", str(out), "-p", "m2", "-m", "2", "-w", os.path.join(GOLDEN, "prep_whitelist.txt"),
                   "-b", os.path.join(GOLDEN, "sample_blacklist.txt")) == 0
    f = mmread(str(out / "m2.filtered.mtx"))
    assert np.array_equal(f.col, gold["laf_m2_col"]) and f.shape == tuple(gold["laf_m2_shape"])
    assert run_cli("prep-like", "-i", txt, "-r", str(out / "m2.genes.txt"), "-o", str(tmp_path / "like")) == 0
    g = mmread(str(tmp_path / "like" / "filtered.mtx"))
    assert np.array_equal(g.toarray(), f.toarray())
    assert (tmp_path / "like" / "genes.txt").read_text() == (out / "m2.genes.txt").read_text()
    assert (tmp_path / "like" / "prep-like_commandline_args.json").exists()
