#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Run in the build container only (the reference does not exist on the GPU box):

    python tests/golden/make_golden.py

It imports the reference package from /root/reference (with the import-only
`numba` stand-in in tests/golden/_numba_standin, because numba is not
installable here), executes the reference's own functions on seeded inputs and
stores inputs + outputs as small .npz files.  Nothing of the reference's source
is stored; the fixtures are data.

What each file pins (reference file:line):
  psi_gammaln.npz      hpf_numba.psi / cgammaln (hpf_numba.py:16-22) == SciPy's
                       C psi / gammaln, on the reference's own test points
                       (tests/test_inference.py:24-37) plus a dense grid
  ops_f64/f32.npz      compute_Xphi_data (:54-114), compute_Xphi_data_numpy
                       (:117-125), compute_loading_shape_update (:128-156),
                       compute_loading_rate_update (:159-177),
                       compute_capacity_rate_update (:180-188), compute_pois_llh
                       (:24-51) + the numpy llh (loss.py:132-134), on the
                       reference's conftest recipe (tests/conftest.py:10-38)
  fit_*.npz            whole scHPF.fit()/project() traces (scHPF_.py:425-503,
                       526-780): bp, dp, per-check loss, final xi/theta/eta/beta
  trials_data_k5_f64.npz  run_trials / run_trials_pool (scHPF_.py:968-1332): winner and losses
  trials_validation_k5_f64.npz  run_trials(vcells=...): the validation-loss trace of
                       projection_loss_function (loss.py:37-102)
  pbmc_like_data.npz   the COO the reference's loader makes of its own test data
                       file tests/_data/PJ030merge...matrix.txt (data only)
  ref_model_f64.joblib a model file written by the reference's save_model
  prep_data.npz        the `prep` / `prep-like` pipeline (preprocessing.py:138-495, util.py:116-215)
                       on the reference's test matrix: load_and_filter with the list files
                       prep_whitelist.txt (made here from the matrix's own gene table) and
                       sample_blacklist.txt (the reference's tests/_data file, data only),
                       load_like on a shuffled subset, seeded subsample_cell_ixs /
                       split_validation_cells draws, insert_coo_rows
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(HERE, "_numba_standin"))

import numpy as np  # noqa: E402
from scipy.sparse import coo_matrix  # noqa: E402

import schpf  # noqa: E402  (the reference)
from schpf import hpf_numba, scHPF  # noqa: E402
from schpf import preprocessing as prep  # noqa: E402
import schpf.loss as ls  # noqa: E402

assert schpf.__file__.startswith("/root/reference"), schpf.__file__


def conftest_data(seed=42, ncells=300, ngenes=1000, frac=0.03):
    """tests/conftest.py:14-25 recipe (module seed at :8)."""
    np.random.seed(seed)
    nnz = int(ncells * ngenes * frac)
    x = np.random.negative_binomial(2, 0.5, nnz)
    x[x == 0] = 1
    ci = np.random.randint(0, ncells, nnz, dtype=np.int32)
    gi = np.random.randint(0, ngenes, nnz, dtype=np.int32)
    X = coo_matrix((x, (ci, gi)), (ncells, ngenes), dtype=np.int32)
    X.sum_duplicates()
    return X


def gam(g):
    return g.vi_shape.copy(), g.vi_rate.copy()


def make_psi_gammaln():
    pts = np.array([0.0001, 0.001, 0.01, 0.1, 1, 10, 100, 1000], dtype=np.float64)
    grid = np.concatenate([
        pts,
        np.logspace(-4, 6, 1500),
        np.linspace(0.9, 2.1, 241),          # around the positive root 1.4616
        np.arange(1, 200, dtype=np.float64),  # integer counts + 1 for gammaln
    ])
    psi = np.array([hpf_numba.psi(float(v)) for v in grid])
    gln = np.array([hpf_numba.cgammaln(float(v)) for v in grid])
    np.savez_compressed(os.path.join(HERE, "psi_gammaln.npz"), x=grid, psi=psi,
                        gammaln=gln, n_test_points=len(pts))


def make_ops(dtype, tag):
    X = conftest_data()
    np.random.seed(1234)
    model = scHPF(4, dtype=dtype)
    model._initialize(X)
    # Xphi fixture as tests/test_inference.py:17-20
    random_phi = np.random.dirichlet(np.ones(model.nfactors), X.data.shape[0]).astype(dtype)
    xphi_in = X.data[:, None] * random_phi
    th, be, xi, eta = model.theta, model.beta, model.xi, model.eta
    out = dict(
        x=X.data, row=X.row, col=X.col, shape=np.array(X.shape),
        theta_shape=th.vi_shape, theta_rate=th.vi_rate,
        beta_shape=be.vi_shape, beta_rate=be.vi_rate,
        xi_shape=xi.vi_shape, xi_rate=xi.vi_rate,
        eta_shape=eta.vi_shape, eta_rate=eta.vi_rate,
        a=model.a, c=model.c, ap=model.ap, cp=model.cp, bp=model.bp, dp=model.dp,
        xphi_in=xphi_in,
    )
    out["xphi"] = hpf_numba.compute_Xphi_data(
        X.data, X.row, X.col, th.vi_shape, th.vi_rate, be.vi_shape, be.vi_rate)
    out["xphi_numpy"] = hpf_numba.compute_Xphi_data_numpy(X, th, be)
    out["theta_shape_upd"] = hpf_numba.compute_loading_shape_update(
        xphi_in, X.row, X.shape[0], model.a)
    out["beta_shape_upd"] = hpf_numba.compute_loading_shape_update(
        xphi_in, X.col, X.shape[1], model.c)
    out["theta_rate_upd"] = hpf_numba.compute_loading_rate_update(
        xi.vi_shape, xi.vi_rate, be.vi_shape, be.vi_rate)
    out["beta_rate_upd"] = hpf_numba.compute_loading_rate_update(
        eta.vi_shape, eta.vi_rate, th.vi_shape, th.vi_rate)
    out["eta_rate_upd"] = hpf_numba.compute_capacity_rate_update(
        be.vi_shape, be.vi_rate, model.dp)
    out["xi_rate_upd"] = hpf_numba.compute_capacity_rate_update(
        th.vi_shape, th.vi_rate, model.bp)
    out["llh"] = hpf_numba.compute_pois_llh(
        X.data, X.row, X.col, th.vi_shape, th.vi_rate, be.vi_shape, be.vi_rate)
    out["llh_numpy"] = ls.pois_llh_pointwise(X, theta=th, beta=be, single_process=True)
    out["mean_neg_llh"] = ls.mean_negative_pois_llh(X, theta=th, beta=be)
    out["cell_score"] = model.cell_score()
    out["gene_score"] = model.gene_score()
    out["cellmean_neg_llh"] = model.cellmean_negative_pois_llh(X)
    np.savez_compressed(os.path.join(HERE, "ops_%s.npz" % tag), **out)


def trace(model, X, losses, extra=None):
    d = dict(x=X.data, row=X.row, col=X.col, shape=np.array(X.shape),
             bp=model.bp, dp=model.dp, loss=np.array(losses, dtype=np.float64),
             nfactors=model.nfactors)
    for name in ("xi", "theta", "eta", "beta"):
        s, r = gam(getattr(model, name))
        d[name + "_shape"], d[name + "_rate"] = s, r
    if extra:
        d.update(extra)
    return d


def make_fit(X, K, seed, dtype, fname, max_iter=60, **fit_kw):
    np.random.seed(seed)
    model = scHPF(K, dtype=dtype, max_iter=max_iter, verbose=False)
    model.fit(X, **fit_kw)
    d = trace(model, X, model.loss, dict(seed=seed, max_iter=max_iter, niter_checks=len(model.loss)))
    np.savez_compressed(os.path.join(HERE, fname), **d)
    return model


def make_project(model, X, fname, seed=7):
    """project() on the first 40 cells (scHPF_.py:448-503), reinit default."""
    Xp = X.tocsr()[:40].tocoo()
    np.random.seed(seed)
    proj = model.project(Xp, max_iter=20)
    d = trace(proj, Xp, proj.loss, dict(seed=seed))
    np.savez_compressed(os.path.join(HERE, fname), **d)


def make_trials(X, fname):
    """run_trials (scHPF_.py:968-1147): 3 restarts, best final loss wins; also the reprojected
    variant.  Losses of every restart are kept so the selection can be checked."""
    import schpf as ref
    np.random.seed(11)
    best, rest = ref.run_trials(X, 5, ntrials=3, max_iter=30, verbose=False, return_all=True)
    d = trace(best, X, best.loss, dict(seed=11, rejected_final=np.array([m.loss[-1] for m in rest])))
    np.random.seed(12)
    bests = ref.run_trials_pool(X, [4, 6], ntrials=2, njobs=1, max_iter=20, verbose=False)
    d["pool_nfactors"] = np.array([m.nfactors for m in bests])
    d["pool_checks"] = np.array([len(m.loss) for m in bests])
    np.savez_compressed(os.path.join(HERE, fname), **d)


def make_validation(X, fname):
    """run_trials with held-out validation cells (scHPF_.py:1097-1106 -> loss.py:37-102
    projection_loss_function): the model trains on the first 70 cells; at every check the last
    30 are projected onto the current eta/beta (warm-started from the previous projection) and
    THEIR mean negative llh is the loss that drives the stop rule."""
    import schpf as ref
    csr = X.tocsr()
    Xt, Xv = csr[:70].tocoo(), csr[70:].tocoo()
    np.random.seed(21)
    model = ref.run_trials(Xt, 5, ntrials=1, max_iter=40, verbose=False, vcells=Xv)
    d = trace(model, Xt, model.loss, dict(seed=21, n_train=70))
    np.savez_compressed(os.path.join(HERE, fname), **d)


def make_prep(txt, fname):
    """Outputs of the reference's prep pipeline on its own test matrix."""
    import shutil
    import pandas as pd
    from schpf import util as ref_util
    umis, genes = prep.load_txt(txt, verbose=False)
    d = {}
    # list files: a whitelist of ~80 % of the matrix's genes (version suffixes altered for some,
    # so that the split-on-dot matters) plus strangers; the reference's own sample blacklist
    rng = np.random.RandomState(5)
    on = rng.rand(len(genes)) < 0.8
    wl = genes.loc[on].copy()
    bump = rng.rand(len(wl)) < 0.3
    wl.loc[bump, 0] = wl.loc[bump, 0].str.split(".").str[0] + ".99"
    wl = pd.concat([wl, pd.DataFrame({0: ["ENSG99999999999.1", "ENSG88888888888.2"], 1: ["NOPE1", "NOPE2"]})])
    wl_file = os.path.join(HERE, "prep_whitelist.txt")
    wl.to_csv(wl_file, sep="\t", header=None, index=None)
    bl_file = os.path.join(HERE, "sample_blacklist.txt")
    shutil.copyfile("/root/reference/tests/_data/sample_blacklist.txt", bl_file)
    os.chmod(bl_file, 0o644)
    for tag, kw in (("m2", dict(min_cells=2, whitelist=wl_file, blacklist=bl_file)),
                    ("m5name", dict(min_cells=5, whitelist=wl_file, blacklist=bl_file, filter_by_gene_name=True)),
                    ("frac_nosplit", dict(min_cells=0.05, whitelist=wl_file, no_split_on_dot=True)),
                    ("m0", dict(min_cells=0))):
        f, g = prep.load_and_filter(txt, verbose=False, **kw)
        d["laf_%s_row" % tag], d["laf_%s_col" % tag], d["laf_%s_data" % tag] = f.row, f.col, f.data
        d["laf_%s_shape" % tag] = np.array(f.shape)
        d["laf_%s_genes" % tag] = g.values.astype(str)
        d["laf_%s_index" % tag] = g.index.values
    # load_like on a shuffled subset of the genes
    perm = np.random.RandomState(9).choice(len(genes), len(genes) - 10, replace=False)
    like_file = os.path.join(HERE, "prep_like_reference.txt")
    genes.loc[perm].to_csv(like_file, sep="\t", header=None, index=None)
    for tag, kw in (("id", {}), ("name", dict(by_gene_name=True)), ("nosplit", dict(no_split_on_dot=True))):
        f, g = prep.load_like(txt, reference=like_file, **kw)
        d["like_%s_row" % tag], d["like_%s_col" % tag], d["like_%s_data" % tag] = f.row, f.col, f.data
        d["like_%s_index" % tag] = g.index.values
    d["like_perm"] = perm
    # masks
    d["mask_min3"] = prep.min_cells_expressing_mask(umis, 3)
    d["mask_frac"] = prep.min_cells_expressing_mask(umis, 0.1)
    # seeded draws
    np.random.seed(11)
    d["sub_plain"] = prep.subsample_cell_ixs(100, 17)
    groups = np.random.RandomState(2).randint(0, 4, 100)
    groups[:3] = 7                                           # a tiny group
    np.random.seed(12)
    d["sub_groups"] = prep.subsample_cell_ixs(100, 30, group_ids=groups, max_group_frac=0.4)
    np.random.seed(13)
    d["sub_choices"] = prep.subsample_cell_ixs(np.arange(50, 150), 20, group_ids=groups, max_group_frac=0.5)
    d["groups"] = groups
    gid_file = os.path.join(HERE, "prep_group_ids.txt")
    np.savetxt(gid_file, groups, fmt="%d")
    np.random.seed(14)
    Xt, Xv, vix = prep.split_validation_cells(umis, 20, gid_file, max_group_frac=0.5, verbose=False)
    d["split_vix"] = vix
    for nm, m in (("split_train", Xt), ("split_valid", Xv)):
        d[nm + "_row"], d[nm + "_col"], d[nm + "_data"], d[nm + "_shape"] = m.row, m.col, m.data, np.array(m.shape)
    back = ref_util.insert_coo_rows(Xt, Xv, vix)
    d["insert_row"], d["insert_col"], d["insert_data"] = back.row, back.col, back.data
    col, nz = ref_util.collapse_coo_rows(umis.T.tocoo())      # genes x cells: 77 empty gene rows
    d["collapse_nz"] = nz
    d["collapse_row"], d["collapse_col"], d["collapse_data"] = col.row, col.col, col.data
    np.savez_compressed(os.path.join(HERE, fname), **d)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "prep":
        make_prep("/root/reference/tests/_data/PJ030merge.c300t400_g0t500.matrix.txt", "prep_data.npz")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "validation":
        txt = "/root/reference/tests/_data/PJ030merge.c300t400_g0t500.matrix.txt"
        Xd, _genes = prep.load_txt(txt, verbose=False)
        make_validation(Xd, "trials_validation_k5_f64.npz")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "trials":
        txt = "/root/reference/tests/_data/PJ030merge.c300t400_g0t500.matrix.txt"
        Xd, _genes = prep.load_txt(txt, verbose=False)
        make_trials(Xd, "trials_data_k5_f64.npz")
        return
    make_psi_gammaln()
    make_ops(np.float64, "f64")
    make_ops(np.float32, "f32")

    txt = "/root/reference/tests/_data/PJ030merge.c300t400_g0t500.matrix.txt"
    Xd, _genes = prep.load_txt(txt, verbose=False)
    np.savez_compressed(os.path.join(HERE, "pbmc_like_data.npz"), x=Xd.data, row=Xd.row,
                        col=Xd.col, shape=np.array(Xd.shape))
    Xc = conftest_data()

    m64 = make_fit(Xd, 5, 0, np.float64, "fit_data_k5_s0_f64.npz")
    make_fit(Xd, 5, 1, np.float64, "fit_data_k5_s1_f64.npz")
    make_fit(Xd, 5, 0, np.float32, "fit_data_k5_s0_f32.npz")
    make_fit(Xc, 4, 0, np.float64, "fit_conf_k4_s0_f64.npz", max_iter=40)
    make_fit(Xc, 4, 0, np.float32, "fit_conf_k4_s0_f32.npz", max_iter=40)
    make_fit(Xd, 5, 0, np.float64, "fit_data_k5_s0_f64_simul.npz", max_iter=30,
             beta_theta_simultaneous=True)
    make_fit(Xd, 5, 0, np.float64, "fit_data_k5_s0_f64_single.npz", max_iter=30,
             single_process=True)
    make_fit(Xd, 5, 3, np.float64, "fit_data_k5_s3_f64_batch.npz", max_iter=30,
             batchsize=32)
    make_project(m64, Xd, "project_data_k5_f64.npz")
    make_trials(Xd, "trials_data_k5_f64.npz")
    make_validation(Xd, "trials_validation_k5_f64.npz")
    make_prep(txt, "prep_data.npz")
    schpf.save_model(m64, os.path.join(HERE, "ref_model_f64.joblib"))
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
