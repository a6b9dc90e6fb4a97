"""TEST INFRASTRUCTURE: an engine with HipEngine's interface backed by the CPU oracle, so that the
host-side driver logic (krotov.jl_amd/optimize.py) can be exercised without a GPU.  Never shipped,
never importable from the package."""
import numpy as np

import oracle_c as oc


class OracleEngine:
    def __init__(self, H0, Hc, tlist, K, hc_present=None, traj_generator=None, store_fw=True, prop_tol=0.0, **_):
        self.H0 = np.asarray(H0, np.complex128)
        self.Hc = np.asarray(Hc, np.complex128)
        self.tlist = np.asarray(tlist, float)
        self.K = K
        self.mask = None if hc_present is None else np.asarray(hc_present, np.uint8)
        self.tg = None if traj_generator is None else np.asarray(traj_generator)
        self.X = self.Phi = self.psi0 = None
        self.calls = []

    def set_initial_states(self, psi0):
        self.psi0 = np.array(psi0, np.complex128)

    def forward(self, eps):
        self.calls.append("forward")
        if self.tg is not None:
            return self._forward_ensemble(eps)
        psiT, self.Phi = oc.forward(self.H0, self.Hc, self.tlist, eps, self.psi0)
        return psiT

    # per-trajectory generators (G > 1): plain NumPy restatement of optimize.jl:247-265, 279-371
    def _forward_ensemble(self, eps):
        import krotov_oracle as o
        out = [o.forward(self.H0[g], self.Hc[g], self.tlist, eps, self.psi0[k:k + 1]) for k, g in enumerate(self.tg)]
        self.Phi = np.concatenate([p[1] for p in out], axis=1)
        return np.concatenate([p[0] for p in out], axis=0)

    def _iteration_ensemble(self, eps, chiT, S, lam):
        import krotov_oracle as o
        K, (L, NT), tl = self.K, eps.shape, self.tlist
        X = [o.backward(self.H0[g], self.Hc[g], tl, eps, chiT[k:k + 1])[:, 0] for k, g in enumerate(self.tg)]
        self.X = np.stack(X, axis=1)
        psi = self.psi0.copy()
        eps_new = eps.copy()
        ga = np.zeros(L)
        for n in range(NT):
            du = np.zeros(L)
            for l in range(L):
                for k, g in enumerate(self.tg):
                    if self.mask is None or self.mask[g, l]:
                        du[l] += np.vdot(X[k][n], self.Hc[g, l] @ psi[k]).imag
            alpha = S[:, n] / lam
            eps_new[:, n] = eps[:, n] + alpha * du
            ga += alpha * np.abs(du) ** 2 * (tl[n + 1] - tl[n])
            for k, g in enumerate(self.tg):
                psi[k] = o.step_propagator(self.H0[g], self.Hc[g], eps_new[:, n], tl[n + 1] - tl[n]) @ psi[k]
        return eps_new, ga, psi

    def iteration(self, eps_guess, chiT, S, lam):
        self.calls.append("iteration")
        if self.tg is not None:
            return self._iteration_ensemble(np.array(eps_guess), np.array(chiT), np.asarray(S), np.asarray(lam))
        eps_new, ga, psiT, self.X, self.Phi = oc.iteration(self.H0, self.Hc, self.tlist, np.array(eps_guess), S,
                                                           lam, self.psi0, np.array(chiT), mask=self.mask)
        return eps_new, ga, psiT

    def read_storage(self, which, k, n):
        return (self.X if which == "bw" else self.Phi)[n, k].copy()

    def close(self):
        pass


def factory(H0, Hc, tlist, K, **kw):
    return OracleEngine(H0, Hc, tlist, K, **kw)
