"""CPU ORACLE (NumPy/SciPy twin) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module, and only as the checker.  The product path (krotov.jl_amd) never
imports anything from oracle/.

PARITY UNPINNED: the reference (JuliaQuantumControl/Krotov.jl) is Julia, `julia`
is not installed in the build container, and the arithmetic of the path lives in
the un-vendored, un-pinned packages QuantumPropagators.jl / QuantumControl.jl
(/root/reference/Project.toml:19 `QuantumControl = ">=0.11.1"`, no Manifest).
The reference ships no golden vectors.  This restatement is therefore pinned
only by (i) the threshold assertions of the reference's own TLS test
(/root/reference/test/test_tls_optimization.jl:66-67) and (ii) independent
mathematics (scipy.linalg.expm, closed-form 2x2 exponentials,
eigendecomposition) -- and, recalled from memory, the iteration table that the documentation of the Python
package `krotov` prints for the same problem family (tests/test_oracle.py).  See DESIGN.md "Oracle".

It restates, in the reference's own order of arithmetic:

* krotov_initial_fw_prop!   /root/reference/src/optimize.jl:247-265
* krotov_iteration          /root/reference/src/optimize.jl:279-371
* update_result! (tau/J_T)  /root/reference/src/optimize.jl:374-397
* finalize_result!          /root/reference/src/optimize.jl:400-406
* pulse / shape set-up      /root/reference/src/workspace.jl:101-106,118-125

with the upstream (QuantumPropagators ExpProp, piecewise-constant) semantics
    U_n = exp(-i (H0 + sum_l eps_l[n] H_l) (t_{n+1}-t_n))
and the backward propagation through the *adjoint* generator with
`backward=true` (/root/reference/src/workspace.jl:69,149-161):
    chi(t_n) = exp(-i H^dagger (-dt)) chi(t_{n+1}) = U_n^dagger chi(t_{n+1}).

Array conventions (NumPy, C-order): H0 (N,N), Hc (L,N,N), eps (L,NT),
states (K,N), storage (NT+1,K,N); all complex128 / float64.
"""
import numpy as np
from scipy.linalg import expm

__all__ = [
    "step_propagator", "forward", "backward", "update_forward", "iteration",
    "taus", "J_T_sm", "J_T_ss", "J_T_re", "chi_sm", "chi_ss", "chi_re",
    "blackman", "flattop", "discretize", "discretize_on_midpoints",
    "optimize_tls_kat",
]


# ---------------------------------------------------------------- propagators
def generator(H0, Hc, eps_n):
    """H(eps_n) = H0 + sum_l eps_n[l] * Hc[l]  (what `evaluate(generator, tlist, n)`
    yields for linear controls; consumed at optimize.jl:257,309,361)."""
    H = np.array(H0, dtype=np.complex128, copy=True)
    for l in range(len(Hc)):
        H = H + eps_n[l] * Hc[l]
    return H


def step_propagator(H0, Hc, eps_n, dt):
    """ExpProp step: dense exp(-i H dt) [UPSTREAM QuantumPropagators ExpProp]."""
    return expm(-1j * generator(H0, Hc, eps_n) * dt)


def forward(H0, Hc, tlist, eps, psi0, store=True):
    """krotov_initial_fw_prop! (optimize.jl:247-265) for all k.

    Returns (psiT (K,N), storage (NT+1,K,N) or None).  Storage slot n+1 holds the
    state after step n (optimize.jl:263); slot 0 holds psi0 (optimize.jl:254).
    """
    psi = np.array(psi0, dtype=np.complex128, copy=True)
    K, N = psi.shape
    NT = len(tlist) - 1
    Phi = np.empty((NT + 1, K, N), np.complex128) if store else None
    if store:
        Phi[0] = psi
    for n in range(NT):
        U = step_propagator(H0, Hc, eps[:, n], tlist[n + 1] - tlist[n])
        psi = psi @ U.T  # psi_k <- U psi_k for every k
        if store:
            Phi[n + 1] = psi
    return psi, Phi


def backward(H0, Hc, tlist, eps, chiT):
    """Backward sweep (optimize.jl:303-317): X[NT] = chi(T); X[n] = U_n^† X[n+1]."""
    chi = np.array(chiT, dtype=np.complex128, copy=True)
    K, N = chi.shape
    NT = len(tlist) - 1
    X = np.empty((NT + 1, K, N), np.complex128)
    X[NT] = chi
    for n in range(NT - 1, -1, -1):
        # propagator of the adjoint generator, backward in time: exp(-i H^† (-dt))
        Hn = generator(H0, Hc, eps[:, n])
        Ud = expm(-1j * Hn.conj().T * (-(tlist[n + 1] - tlist[n])))
        chi = chi @ Ud.T
        X[n] = chi
    return X


def update_forward(H0, Hc, tlist, eps_guess, S, lam, psi0, X, mask=None, store=True):
    """Sequential update sweep (optimize.jl:321-370).

    eps_new[l][n] = eps_guess[l][n] + S[l][n]/lam[l] * sum_k Im<chi_k(t_n)|H_l|psi_k(t_n)>
    with psi propagated under the *new* pulse.  Sum order: l outer, k inner, serial
    (optimize.jl:340-349).  Returns (eps_new, g_a_int, psiT, Phi).
    """
    L, NT = eps_guess.shape
    psi = np.array(psi0, dtype=np.complex128, copy=True)
    K, N = psi.shape
    eps_new = np.array(eps_guess, dtype=np.float64, copy=True)
    g_a_int = np.zeros(L)
    Phi = np.empty((NT + 1, K, N), np.complex128) if store else None
    if store:
        Phi[0] = psi
    for n in range(NT):
        dt = tlist[n + 1] - tlist[n]
        du = np.zeros(L)
        for l in range(L):
            if mask is not None and not mask[l]:
                continue  # mu === nothing, optimize.jl:344
            for k in range(K):
                du[l] += np.vdot(X[n, k], Hc[l] @ psi[k]).imag  # dot(chi, mu, psi), :346
        for l in range(L):
            alpha = S[l, n] / lam[l]
            eps_new[l, n] = eps_guess[l, n] + alpha * du[l]
            g_a_int[l] += alpha * abs(du[l]) ** 2 * dt
        U = step_propagator(H0, Hc, eps_new[:, n], dt)
        psi = psi @ U.T
        if store:
            Phi[n + 1] = psi
    return eps_new, g_a_int, psi, Phi


def iteration(H0, Hc, tlist, eps_guess, S, lam, psi0, chiT, mask=None):
    """One krotov_iteration (optimize.jl:279-371) given chi(T)."""
    X = backward(H0, Hc, tlist, eps_guess, chiT)
    eps_new, g_a_int, psiT, Phi = update_forward(H0, Hc, tlist, eps_guess, S, lam, psi0, X, mask)
    return eps_new, g_a_int, psiT, X, Phi


# ---------------------------------------------------------------- functionals
# [UPSTREAM-RECALLED QuantumControl.Functionals], all weights 1.
def taus(psiT, targets):
    """tau_k = <psi_k^tgt | psi_k(T)>  (taus!, optimize.jl:381)."""
    return np.einsum("kn,kn->k", np.conj(targets), psiT)


def J_T_sm(tau):
    return 1.0 - abs(np.mean(tau)) ** 2


def J_T_ss(tau):
    return 1.0 - np.mean(np.abs(tau) ** 2)


def J_T_re(tau):
    return 1.0 - np.mean(tau).real


def chi_sm(tau, targets):
    K = len(tau)
    return (np.sum(tau) / K ** 2) * targets


def chi_ss(tau, targets):
    K = len(tau)
    return (tau / K)[:, None] * targets


def chi_re(tau, targets):
    K = len(tau)
    return targets / (2.0 * K)


# ---------------------------------------------------------------- shapes / discretisation
# [UPSTREAM-RECALLED QuantumPropagators.Shapes / Controls]
def blackman(t, t0, T, a=0.16):
    dT = T - t0
    box = 1.0 if (t0 <= t <= T) else 0.0
    return 0.5 * box * (1.0 - a - np.cos(2 * np.pi * (t - t0) / dT) + a * np.cos(4 * np.pi * (t - t0) / dT))


def flattop(t, T, t_rise, t0=0.0, t_fall=None):
    if t_fall is None:
        t_fall = t_rise
    f = 0.0
    if t0 <= t <= T:
        f = 1.0
        if t <= t0 + t_rise:
            f = blackman(t, t0, t0 + 2 * t_rise)
        elif t >= T - t_fall:
            f = blackman(t, T - 2 * t_fall, T)
    return f


def discretize_on_midpoints(control, tlist):
    """workspace.jl:102,119,123 -> pulse values on the NT intervals."""
    tlist = np.asarray(tlist, float)
    NT = len(tlist) - 1
    if callable(control):
        tm = np.empty(NT)
        tm[0] = tlist[0]
        tm[-1] = tlist[-1]
        for i in range(1, NT - 1):
            tm[i] = tlist[i] + 0.5 * (tlist[i + 1] - tlist[i])
        return np.array([control(t) for t in tm], float)
    c = np.asarray(control, float)
    if len(c) == NT:
        return c.copy()
    if len(c) == NT + 1:
        # exact inverse of discretize(): c[i] = (v[i-1] + v[i]) / 2  [UPSTREAM-RECALLED; the hand-over tests
        # test_tls_optimization.jl:126,160 (J_T continuity to 1e-14) only hold if the round trip is exact]
        # [UPSTREAM-RECALLED QuantumPropagators.Controls.discretize_on_midpoints(::Vector, tlist)]: first AND last interval value
        # pinned to the first / last point value, the interior by the recursion -- on a round trip through discretize() identical
        # to the pure recursion (the last interval comes back exactly), on an arbitrary array it differs in the last value only
        v = np.empty(NT)
        v[0] = c[0]
        v[-1] = c[-1]
        for i in range(1, NT - 1):
            v[i] = 2.0 * c[i] - v[i - 1]
        return v
    raise ValueError("control array must have length len(tlist) or len(tlist)-1")


def discretize(pulse, tlist):
    """finalize_result! (optimize.jl:403-405): interval values -> values on tlist."""
    tlist = np.asarray(tlist, float)
    if callable(pulse):
        return np.array([pulse(t) for t in tlist], float)
    p = np.asarray(pulse, float)
    if len(p) == len(tlist):
        return p.copy()
    if len(p) == len(tlist) - 1:
        v = np.zeros(len(p) + 1)
        v[0] = p[0]
        v[-1] = p[-1]
        for i in range(1, len(v) - 1):
            v[i] = 0.5 * (p[i - 1] + p[i])
        return v
    raise ValueError("pulse array must have length len(tlist) or len(tlist)-1")


# ---------------------------------------------------------------- KAT driver
def tls_problem(npoints=501):
    """test_tls_optimization.jl:12-28,50-63."""
    H0 = -0.5 * np.array([[1, 0], [0, -1]], np.complex128)
    Hc = np.array([[[0, 1], [1, 0]]], np.complex128)
    tlist = np.linspace(0.0, 5.0, npoints)
    guess = lambda t: 0.2 * flattop(t, T=5.0, t_rise=0.3)
    psi0 = np.array([[1, 0]], np.complex128)
    tgt = np.array([[0, 1]], np.complex128)
    return H0, Hc, tlist, guess, psi0, tgt


def optimize_tls_kat(iters=5, npoints=501):
    """Restates optimize_krotov (optimize.jl:161-235) for KAT-TLS.
    Returns (J_T history, list of pulses per iteration, optimized control on tlist)."""
    H0, Hc, tlist, guess, psi0, tgt = tls_problem(npoints)
    NT = len(tlist) - 1
    eps = discretize_on_midpoints(guess, tlist)[None, :]
    S = np.ones((1, NT))
    lam = np.array([1.0])
    psiT, _ = forward(H0, Hc, tlist, eps, psi0)
    tau = taus(psiT, tgt)
    hist = [J_T_sm(tau)]
    pulses = [eps.copy()]
    for _ in range(iters):
        chiT = chi_sm(tau, tgt)
        eps, g, psiT, _, _ = iteration(H0, Hc, tlist, eps, S, lam, psi0, chiT)
        tau = taus(psiT, tgt)
        hist.append(J_T_sm(tau))
        pulses.append(eps.copy())
    return hist, pulses, discretize(eps[0], tlist)
