"""A CPU stand-in for schpf_amd.DeviceCAVI used ONLY by tests of the sharded driver
(schpf_amd/sharded.py) on boxes without a GPU.  Same duck-typed surface (step_local /
step_finish / exchange / loss_terms / set_gamma / get_gamma), arithmetic by the CPU oracle.
It states the sharded protocol in numpy: what goes into the exchange buffer and which sums
each update reads."""
import numpy as np
import torch
from scipy.special import gammaln

from oracle import hpf_oracle as orc


class OracleEngine(object):
    def __init__(self, X, nfactors, dtype=np.float64):
        self.X = X.tocoo()
        self.N, self.G = X.shape
        self.K = nfactors
        self.dtype = np.dtype(dtype)
        self._exchange = np.zeros(self.G * self.K + self.K, dtype=self.dtype)
        self.exchange = torch.from_numpy(self._exchange)      # shares memory
        self.g = {}
        self.hyp = None
        self.s_beta = None
        self.pending_xphi = None

    def set_hypers(self, a, c, bp, dp):
        self.hyp = (a, c, bp, dp)

    def set_gamma(self, name, shape, rate):
        self.g[name] = [np.array(shape, dtype=self.dtype), np.array(rate, dtype=self.dtype)]
        if name == "theta":
            self._exchange[self.G * self.K:] = (self.g["theta"][0] / self.g["theta"][1]).sum(0)
        if name == "beta":
            self.s_beta = (self.g["beta"][0] / self.g["beta"][1]).astype(np.float64).sum(0)

    def get_gamma(self, name):
        return self.g[name][0].copy(), self.g[name][1].copy()

    def init_phi_host(self, xphi):
        self.pending_xphi = np.asarray(xphi, dtype=np.float64)

    def step_local(self, freeze_genes=False, simultaneous=False, side=None):
        if side == "cell":
            return                         # this stand-in does both sides in the 'gene' (or only) call
        X, K = self.X, self.K
        ths, thr = self.g["theta"]
        bes, ber = self.g["beta"]
        if self.pending_xphi is not None:
            xphi = self.pending_xphi.astype(self.dtype)
            self.pending_xphi = None
        else:
            xphi = orc.compute_Xphi_data(X.data, X.row, X.col, ths, thr, bes, ber)
        self.a_theta = orc.compute_loading_shape_update(xphi, X.row, self.N, 0.0)
        if not freeze_genes:
            self._exchange[:self.G * K] = orc.compute_loading_shape_update(xphi, X.col, self.G, 0.0).ravel()

    def step_finish(self, freeze_genes=False, simultaneous=False):
        a, c, bp, dp = self.hyp
        K, G = self.K, self.G
        s_theta = self._exchange[G * K:].astype(np.float64).copy()       # all-reduced, OLD theta
        s_beta_for_theta = self.s_beta
        if not freeze_genes:
            eta_s, eta_r = self.g["eta"]
            bes = (c + self._exchange[:G * K].reshape(G, K)).astype(self.dtype)
            ber = ((eta_s / eta_r)[:, None] + s_theta[None, :]).astype(self.dtype)
            self.g["beta"] = [bes, ber]
            self.g["eta"][1] = (dp + (bes / ber).sum(1)).astype(self.dtype)
            new_s_beta = (bes / ber).astype(np.float64).sum(0)
            if not simultaneous:
                s_beta_for_theta = new_s_beta
            self.s_beta = new_s_beta
        xi_s, xi_r = self.g["xi"]
        ths = (a + self.a_theta).astype(self.dtype)
        thr = ((xi_s / xi_r)[:, None] + s_beta_for_theta[None, :]).astype(self.dtype)
        self.g["theta"] = [ths, thr]
        self.g["xi"][1] = (bp + (ths / thr).sum(1)).astype(self.dtype)
        self._exchange[G * K:] = (ths / thr).sum(0)

    def loss_terms(self):
        X = self.X
        ths, thr = self.g["theta"]
        bes, ber = self.g["beta"]
        llh = orc.compute_pois_llh(X.data, X.row, X.col, ths, thr, bes, ber).astype(np.float64)
        gl = gammaln(X.data + 1.0)
        return float((llh + gl).sum()), float(gl.sum()), int(X.nnz)


class OracleShardEngine(OracleEngine):
    """The stand-in with DeviceCAVI's constructor and life cycle (create, hint, upload, close), so
    that schpf_amd.sharded.ThreadedShards -- and through it scHPF.fit(X, devices=[...]) -- can be
    driven on a box without a GPU."""

    def __init__(self, ncells, ngenes, nfactors, dtype=np.float64, device=0):
        self._shape = (int(ncells), int(ngenes))
        self._k, self._dt = int(nfactors), np.dtype(dtype)
        self.nnz = 0

    def hint_sharded(self, on=True):
        pass

    def upload(self, X):
        assert tuple(X.shape) == self._shape
        OracleEngine.__init__(self, X, self._k, self._dt)
        self.nnz = int(X.nnz)

    def synchronize(self):
        pass

    def close(self):
        pass
