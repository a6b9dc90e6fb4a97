"""TEST INFRASTRUCTURE: an engine with HipEngine's interface backed by the CPU oracle, so that the
host-side driver logic (krotov.jl_amd/optimize.py) can be exercised without a GPU.  Never shipped,
never importable from the package."""
import numpy as np

import oracle_c as oc


class OracleEngine:
    def __init__(self, H0, Hc, tlist, K, hc_present=None, store_fw=True, prop_tol=0.0, **_):
        self.H0 = np.asarray(H0, np.complex128)
        self.Hc = np.asarray(Hc, np.complex128)
        self.tlist = np.asarray(tlist, float)
        self.K = K
        self.mask = None if hc_present is None else np.asarray(hc_present, np.uint8)
        self.X = self.Phi = self.psi0 = None
        self.calls = []

    def set_initial_states(self, psi0):
        self.psi0 = np.array(psi0, np.complex128)

    def forward(self, eps):
        self.calls.append("forward")
        psiT, self.Phi = oc.forward(self.H0, self.Hc, self.tlist, eps, self.psi0)
        return psiT

    def iteration(self, eps_guess, chiT, S, lam):
        self.calls.append("iteration")
        eps_new, ga, psiT, self.X, self.Phi = oc.iteration(self.H0, self.Hc, self.tlist, np.array(eps_guess), S,
                                                           lam, self.psi0, np.array(chiT), mask=self.mask)
        return eps_new, ga, psiT

    def read_storage(self, which, k, n):
        return (self.X if which == "bw" else self.Phi)[n, k].copy()

    def close(self):
        pass


def factory(H0, Hc, tlist, K, **kw):
    return OracleEngine(H0, Hc, tlist, K, **kw)
