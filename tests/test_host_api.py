"""Host-side API semantics that need no GPU: HPF_Gamma, estimator parameters, _setup,
model files.  Modelled on the reference's tests/test_scHPF_model.py."""
import os

import numpy as np
import pytest
from numpy.testing import assert_equal, assert_array_equal

from conftest import GOLDEN, load_golden, golden_coo, synthetic_counts
import schpf
from schpf import HPF_Gamma, scHPF, combine_across_cells, load_model, save_model


@pytest.fixture()
def data():
    return synthetic_counts(300, 1000, 0.03)


@pytest.fixture(params=[np.float64, np.float32])
def model_uninit(request):
    return scHPF(4, dtype=request.param)


@pytest.fixture()
def model(model_uninit, data):
    np.random.seed(3)
    model_uninit._initialize(data)
    return model_uninit


def test_version():
    assert schpf.__version__ == "0.5.0"


def test_setup_meanvar(model_uninit, data):
    bp, dp, *_ = model_uninit._setup(X=data, freeze_genes=False, reinit=True)
    cell_sums, gene_sums = data.sum(axis=1), data.sum(axis=0)
    assert_equal(bp, np.mean(cell_sums) / np.var(cell_sums))
    assert_equal(dp, np.mean(gene_sums) / np.var(gene_sums))


def test_setup_dims_dtype(model_uninit, data):
    bp, dp, xi, eta, theta, beta = model_uninit._setup(X=data, freeze_genes=False, reinit=True)
    assert xi.dims == (300,) and eta.dims == (1000,)
    assert theta.dims == (300, 4) and beta.dims == (1000, 4)
    for g in (xi, eta, theta, beta):
        assert g.vi_shape.dtype == model_uninit.dtype and g.vi_rate.dtype == model_uninit.dtype


def test_setup_matches_reference_draws():
    """Equal seed -> the reference's initial distributions (golden ops fixtures were
    initialised with np.random.seed(1234) on the conftest matrix)."""
    g = load_golden("ops_f64.npz")
    X = golden_coo(g)
    np.random.seed(1234)
    m = scHPF(4)
    m._initialize(X)
    assert m.bp == float(g["bp"]) and m.dp == float(g["dp"])
    assert_array_equal(m.theta.vi_shape, g["theta_shape"])
    assert_array_equal(m.beta.vi_rate, g["beta_rate"])
    assert_array_equal(m.xi.vi_rate, g["xi_rate"])
    assert_array_equal(m.eta.vi_shape, g["eta_shape"])
    # scores are plain host arithmetic
    np.testing.assert_allclose(m.cell_score(), g["cell_score"], rtol=1e-12)
    np.testing.assert_allclose(m.gene_score(), g["gene_score"], rtol=1e-12)


def test_setup_freeze(model, data):
    my_data = data.tocsr()[:20].tocoo()
    bp, dp, xi, eta, theta, beta = (model.bp, model.dp, model.xi, model.eta, model.theta, model.beta)
    model.bp = None
    bp2, dp2, xi2, eta2, theta2, beta2 = model._setup(X=my_data, freeze_genes=True, reinit=True)
    assert_equal(dp2, dp)
    assert eta2 == eta and beta2 == beta
    assert bp2 != bp and xi2.dims != xi.dims and theta2.dims != theta.dims
    model.bp = bp
    bp3 = model._setup(X=my_data, freeze_genes=True, reinit=True)[0]
    assert bp3 == bp and bp3 != bp2


def test_setup_freeze_errors(data):
    m = scHPF(4)
    with pytest.raises(ValueError):
        m._setup(X=data, freeze_genes=True)          # dp is None
    m.dp = 1.0
    with pytest.raises(ValueError):
        m._setup(X=data, freeze_genes=True)          # eta/beta missing


def test_set_ac(model_uninit):
    model_uninit.nfactors = None
    with pytest.raises(ValueError):
        model_uninit.a = -2
    with pytest.raises(ValueError):
        model_uninit.c = -2
    model_uninit.nfactors = 15
    model_uninit.a = -2
    model_uninit.c = -2
    assert model_uninit.a == 1 / np.sqrt(15) and model_uninit.c == 1 / np.sqrt(15)
    assert model_uninit.get_params()["a"] == 1 / np.sqrt(15)


@pytest.mark.parametrize("a_dims", [[5], [5, 10]])
def test_gamma_combine(a_dims):
    b_dims = list(a_dims); b_dims[0] = 3
    np.random.seed(0)
    A = HPF_Gamma.random_gamma_factory(a_dims, 1.0, 1.0)
    B = HPF_Gamma.random_gamma_factory(b_dims, 1.0, 1.0)
    ixs = [0, 4, 6]
    AB = A.combine(B, ixs)
    assert AB.dims[0] == 8
    assert_array_equal(AB.vi_shape[ixs], B.vi_shape)
    rest = np.setdiff1d(np.arange(8), ixs)
    assert_array_equal(AB.vi_rate[rest], A.vi_rate)
    with pytest.raises(AssertionError):
        A.combine(B, [0, 0, 1])
    with pytest.raises(AssertionError):
        A.combine(B, [0, 1])
    with pytest.raises(AssertionError):
        A.combine(B, [0, 1, 8])


def test_gamma_ctor_asserts():
    with pytest.raises(AssertionError):
        HPF_Gamma(np.ones(3), np.ones(4))
    with pytest.raises(AssertionError):
        HPF_Gamma(np.ones(3), np.zeros(3))
    with pytest.raises(AssertionError):
        HPF_Gamma(np.ones(3, np.float32), np.ones(3, np.float64))
    g = HPF_Gamma(np.full(3, 2.0), np.full(3, 4.0))
    np.testing.assert_allclose(g.e_x, 0.5)
    np.testing.assert_allclose(g.e_logx, 0.42278433509846713 - np.log(4.0))
    assert g.sample(3).shape == (3, 3)


def test_combine_across_cells(model, data):
    other = scHPF(4, dtype=model.dtype)
    np.random.seed(4)
    sub = data.tocsr()[:10].tocoo()
    other.eta, other.beta, other.dp = model.eta, model.beta, model.dp
    other.bp = model.bp
    other.xi = HPF_Gamma.random_gamma_factory((10,), 1.0, 1.0, dtype=model.dtype)
    other.theta = HPF_Gamma.random_gamma_factory((10, 4), 1.0, 1.0, dtype=model.dtype)
    ixs = np.arange(0, 20, 2)
    xy = combine_across_cells(model, other, ixs)
    assert xy.ncells == 310 and xy.bp == model.bp
    assert_array_equal(xy.theta.vi_shape[ixs], other.theta.vi_shape)
    other.bp = model.bp + 1
    assert combine_across_cells(model, other, ixs).bp is None


def test_model_files_roundtrip_with_reference(tmp_path, model):
    """A joblib file written by the reference loads here, and what we write pickles the
    reference's class paths (so it loads there)."""
    ref = load_model(os.path.join(GOLDEN, "ref_model_f64.joblib"))
    g = load_golden("fit_data_k5_s0_f64.npz")
    assert isinstance(ref, scHPF) and isinstance(ref.theta, HPF_Gamma)
    assert ref.nfactors == 5 and ref.bp == float(g["bp"])
    assert_array_equal(ref.beta.vi_shape, g["beta_shape"])
    assert set(ref.__dict__) == set(scHPF(5).__dict__)       # same persisted attribute names
    fname = str(tmp_path / "m.joblib")
    save_model(model, fname)
    raw = open(fname, "rb").read()
    assert b"schpf.scHPF_" in raw and b"schpf_amd" not in raw
    back = load_model(fname)
    assert back.theta == model.theta and back.eta == model.eta and back.dtype == model.dtype


def test_project_argument_check(model, data):
    with pytest.raises(ValueError):
        model.project(data, recalc_bp=True, replace=True)


def test_minibatch_over_several_devices_is_refused(data):
    """A batch is a row subset one GPU holds whole: batchsize with devices=[0, 1] is an argument error, raised
    before anything touches a device."""
    from schpf import scHPF
    with pytest.raises(ValueError, match="one device"):
        scHPF(4, verbose=False).fit(data, batchsize=32, devices=[0, 1])


def test_minibatch_generator_cycles_and_wraps():
    from schpf.util import minibatch_ix_generator
    np.random.seed(5)
    gen = minibatch_ix_generator(10, 4)
    got = [next(gen) for _ in range(5)]
    np.random.seed(5)
    order = np.arange(10); np.random.shuffle(order)
    assert_array_equal(got[0], order[:4]); assert_array_equal(got[1], order[4:8])
    assert_array_equal(got[2], np.hstack([order[8:], order[:2]]))      # wraps around
    assert_array_equal(got[3], order[2:6])
    assert sorted(np.concatenate(got[:5])[:10]) == sorted(order)
    gen = minibatch_ix_generator(8, 8)
    assert len(next(gen)) == 8 and len(next(gen)) == 8


def test_trial_drivers_exported():
    from schpf import run_trials, run_trials_pool
    import inspect
    sig = inspect.signature(run_trials)
    for name in ("X", "nfactors", "ntrials", "min_iter", "max_iter", "check_freq", "epsilon",
                 "better_than_n_ago", "dtype", "verbose", "vcells", "vX", "loss_function", "model_kwargs",
                 "return_all", "reproject", "reproject_kwargs", "batchsize", "beta_theta_simultaneous",
                 "loss_smoothing"):
        assert name in sig.parameters
    assert "njobs" in inspect.signature(run_trials_pool).parameters


def test_coo_marginals_match_scipy_sums_exactly():
    """schpf_coo_marginals (threaded, in the library) feeds the empirical bp/dp instead of
    X.sum(1) / X.sum(0) (scHPF_.py:855-857): same integers, hence the same bp and dp bits."""
    from schpf_amd import hpf_hip
    X = golden_coo(load_golden("pbmc_like_data.npz"))
    rs, cs = hpf_hip.coo_marginals(X)
    assert np.array_equal(rs, np.asarray(X.sum(1)).ravel())
    assert np.array_equal(cs, np.asarray(X.sum(0)).ravel())
    for dt in (np.int64, np.float32, np.float64):
        rs2, cs2 = hpf_hip.coo_marginals(X.astype(dt))
        assert np.array_equal(rs, rs2) and np.array_equal(cs, cs2)
    m = scHPF(5)
    bp, dp = m._get_empirical_hypers(X)
    want_bp = m.ap * np.mean(X.sum(1)) / np.var(X.sum(1))
    want_dp = m.cp * np.mean(X.sum(0)) / np.var(X.sum(0))
    assert bp == want_bp and dp == want_dp
    bad = X.copy()
    bad.row = bad.row.copy()
    bad.row[3] = X.shape[0] + 5
    with pytest.raises(ValueError):
        hpf_hip.coo_marginals(bad)
