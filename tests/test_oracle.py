"""CPU: pins the oracle (test infrastructure) -- the reference's own TLS thresholds, the NumPy/SciPy
twin against the plain-C restatement, independent mathematics, and the committed golden fixtures."""
import os

import numpy as np
import pytest
from scipy.linalg import expm

import krotov_oracle as o

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_tls_kat_reference_thresholds():
    """/root/reference/test/test_tls_optimization.jl:66-67: J_T < 1e-3 after 5 iterations and
    1.0 < max|eps_opt| < 1.2."""
    hist, pulses, ctrl = o.optimize_tls_kat(iters=5, npoints=501)
    assert hist[-1] < 1e-3
    assert 1.0 < np.max(np.abs(ctrl)) < 1.2
    assert all(b < a for a, b in zip(hist, hist[1:]))  # Krotov is monotonic
    # regression values of the restatement (SURVEY 8c scratch check): not Julia's numbers
    ref = [0.9515, 0.7236, 0.1859, 1.055e-2, 4.29e-4, 1.74e-5]
    assert np.allclose(hist, ref, rtol=5e-3)
    assert abs(np.max(np.abs(ctrl)) - 1.0587) < 1e-3


# The iteration table that the documentation of the PYTHON package `krotov` prints for its first example ("Optimization of a
# State-to-State Transfer in a Two-Level-System"; also M. Goerz et al., SciPost Phys. 7, 080 (2019)): the same first-order update by
# the same author, on the problem the reference's own TLS test is modelled on -- H = -sigma_z / 2 + eps(t) sigma_x, |0> -> |1>,
# eps_0(t) = 0.2 flattop(t; T = 5, t_rise = 0.3, Blackman), 500 time points, lambda_a = 5, update shape = the same flattop.
# [UPSTREAM-RECALLED: written down from memory -- there is no network here --, so it does not lift "parity unpinned"; but fifteen
#  three-digit numbers do not agree with a computed sequence by accident.]  Columns: J_T, the printed "integral of g_a" -- which for
#  that package is sum |delta eps|^2 dt, without the factor lambda_a / S(t) --, and Delta J_T.
PY_KROTOV_DOCS_TABLE = [   # iter, J_T, int g_a dt, Delta J_T
    (0, 9.51e-01, 0.0, None),
    (1, 9.24e-01, 2.32e-03, -2.70e-02),
    (2, 8.83e-01, 3.53e-03, -4.11e-02),
    (3, 8.23e-01, 5.22e-03, -6.06e-02),
    (4, 7.38e-01, 7.39e-03, -8.52e-02),
    (5, 6.26e-01, 9.75e-03, -1.11e-01),
]


def python_krotov_docs_problem():
    H0 = -0.5 * np.array([[1, 0], [0, -1]], np.complex128)
    Hc = np.array([[[0, 1], [1, 0]]], np.complex128)
    tlist = np.linspace(0.0, 5.0, 500)
    eps = o.discretize_on_midpoints(lambda t: 0.2 * o.flattop(t, T=5.0, t_rise=0.3), tlist)[None, :]
    S = o.discretize_on_midpoints(lambda t: o.flattop(t, T=5.0, t_rise=0.3), tlist)[None, :]
    return H0, Hc, tlist, eps, S, np.array([5.0]), np.array([[1, 0]], np.complex128), np.array([[0, 1]], np.complex128)


def check_against_python_krotov_docs_table(J_T, d2):
    """J_T[i], d2[i] = sum |eps_i - eps_(i-1)|^2 dt of iterations 0..5 against the table: half a unit of the third digit (+ 20 %)."""
    for it, jt, ga, djt in PY_KROTOV_DOCS_TABLE:
        assert abs(J_T[it] - jt) < 0.6e-3, (it, J_T[it], jt)
        if it > 0:
            assert abs(d2[it] - ga) < 0.6e-5, (it, d2[it], ga)
            assert abs((J_T[it] - J_T[it - 1]) - djt) < (0.6e-4 if abs(djt) < 0.1 else 0.6e-3), (it, J_T[it] - J_T[it - 1], djt)


def test_tls_iteration_table_of_the_python_package_docs():
    H0, Hc, tlist, eps, S, lam, psi0, tgt = python_krotov_docs_problem()
    dt = np.diff(tlist)
    psiT, _ = o.forward(H0, Hc, tlist, eps, psi0)
    tau = o.taus(psiT, tgt)
    J_T, d2 = [o.J_T_sm(tau)], [0.0]
    for _ in range(5):
        old = eps.copy()
        eps, g, psiT, _, _ = o.iteration(H0, Hc, tlist, eps, S, lam, psi0, o.chi_sm(tau, tgt))
        tau = o.taus(psiT, tgt)
        J_T.append(o.J_T_sm(tau))
        d2.append(float(np.sum((eps - old) ** 2 * dt)))
        # the reference's own running integral (optimize.jl:356) carries lambda_a / S(t): (S / lambda) |Delta u|^2 dt = (lambda / S) |delta eps|^2 dt
        m = S[0] > 0
        assert abs(g[0] - np.sum((lam[0] / S[0][m]) * (eps - old)[0][m] ** 2 * dt[m])) < 1e-12
    check_against_python_krotov_docs_table(J_T, d2)


def test_tls_golden_fixture():
    g = np.load(os.path.join(GOLD, "tls_kat.npz"))
    hist, pulses, ctrl = o.optimize_tls_kat(iters=5, npoints=501)
    assert np.allclose(hist, g["J_T"], rtol=0, atol=1e-12)
    assert np.allclose(pulses[1][0], g["pulse_iter1"], rtol=0, atol=1e-12)
    assert np.allclose(pulses[5][0], g["pulse_iter5"], rtol=0, atol=1e-11)
    assert len(ctrl) == 501 and len(pulses[5][0]) == 500  # test_pulse_optimization.jl:22,28


def test_c_expm_against_scipy(oracle_c):
    rng = np.random.default_rng(5)
    for n, sc in [(2, 0.01), (3, 0.3), (8, 0.9), (16, 2.0), (16, 6.0), (9, 30.0), (64, 0.07)]:
        A = (rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))) * sc / np.sqrt(n)
        E = expm(A)
        assert np.max(np.abs(oracle_c.expm(A) - E)) / np.max(np.abs(E)) < 5e-14


def test_tls_closed_form_step():
    """exp(-i (a sz + b sx) dt) = cos(w dt) - i sin(w dt) (a sz + b sx)/w."""
    a, b, dt = -0.5, 0.37, 0.01
    H0 = a * np.diag([1.0, -1.0]).astype(complex)
    Hc = np.array([[[0, 1], [1, 0]]], complex)
    U = o.step_propagator(H0, Hc, [b], dt)
    w = np.hypot(a, b)
    H = H0 + b * Hc[0]
    Uc = np.cos(w * dt) * np.eye(2) - 1j * np.sin(w * dt) * H / w
    assert np.max(np.abs(U - Uc)) < 1e-15


def _case(seed, N, K, NT, L, nonherm=False):
    rng = np.random.default_rng(seed)

    def herm(n):
        A = rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))
        H = 0.5 * (A + A.conj().T)
        return H / np.max(np.abs(np.linalg.eigvalsh(H)))
    H0 = herm(N)
    if nonherm:
        H0 = H0 - 0.05j * np.diag(rng.uniform(0, 1, N))
    Hc = np.array([herm(N) for _ in range(L)])
    tlist = np.cumsum(np.r_[0.0, rng.uniform(0.05, 0.15, NT)])
    eps = rng.normal(size=(L, NT)) * 0.2
    S = rng.uniform(0.5, 1.0, (L, NT))
    lam = rng.uniform(2.0, 4.0, L)
    psi0 = rng.normal(size=(K, N)) + 1j * rng.normal(size=(K, N))
    psi0 /= np.linalg.norm(psi0, axis=1, keepdims=True)
    tgt = rng.normal(size=(K, N)) + 1j * rng.normal(size=(K, N))
    tgt /= np.linalg.norm(tgt, axis=1, keepdims=True)
    return H0, Hc, tlist, eps, S, lam, psi0, tgt


@pytest.mark.parametrize("seed,N,K,NT,L,nonherm", [(1, 2, 1, 30, 1, False), (2, 6, 3, 25, 2, False),
                                                  (3, 10, 4, 20, 3, True), (4, 16, 2, 15, 1, True)])
def test_c_oracle_matches_numpy_twin(oracle_c, seed, N, K, NT, L, nonherm):
    H0, Hc, tlist, eps, S, lam, psi0, tgt = _case(seed, N, K, NT, L, nonherm)
    psiT, Phi = o.forward(H0, Hc, tlist, eps, psi0)
    chiT = o.chi_ss(o.taus(psiT, tgt), tgt)
    ref = o.iteration(H0, Hc, tlist, eps, S, lam, psi0, chiT)
    for mode in (0, 1):
        psiT_c, Phi_c = oracle_c.forward(H0, Hc, tlist, eps, psi0, mode=mode, nthreads=2)
        assert np.max(np.abs(psiT_c - psiT)) < 1e-12
        assert np.max(np.abs(Phi_c - Phi)) < 1e-12
        got = oracle_c.iteration(H0, Hc, tlist, eps, S, lam, psi0, chiT, mode=mode, nthreads=2)
        for a, b in zip(ref, got):
            assert np.max(np.abs(a - b)) < 1e-11 * max(1.0, np.max(np.abs(a)))


def test_backward_is_adjoint_of_forward():
    """<chi(t_n)|psi(t_n)> is independent of n when both move under the same pulse (this is what makes
    X[n] the co-state at the LEFT edge of interval n, optimize.jl:315,330-336) -- also for a
    non-Hermitian generator, where U^dagger != U^-1."""
    H0, Hc, tlist, eps, S, lam, psi0, tgt = _case(9, 7, 2, 20, 2, True)
    _, Phi = o.forward(H0, Hc, tlist, eps, psi0)
    X = o.backward(H0, Hc, tlist, eps, tgt)
    ov = np.einsum("nki,nki->nk", np.conj(X), Phi)
    assert np.max(np.abs(ov - ov[0])) < 1e-13


def test_first_order_update_decreases_functional():
    """With S/lambda small the Krotov update must lower J_T (monotonic convergence)."""
    H0, Hc, tlist, eps, S, lam, psi0, tgt = _case(12, 5, 3, 40, 1, False)
    psiT, _ = o.forward(H0, Hc, tlist, eps, psi0)
    tau = o.taus(psiT, tgt)
    J0 = o.J_T_ss(tau)
    eps_new, ga, psiT1, _, _ = o.iteration(H0, Hc, tlist, eps, S, lam, psi0, o.chi_ss(tau, tgt))
    J1 = o.J_T_ss(o.taus(psiT1, tgt))
    assert J1 < J0 and np.all(ga >= 0)


@pytest.mark.parametrize("name,func", [("hermitian_n16", "ss"), ("nonhermitian_n8", "re")])
def test_golden_iteration_fixtures(oracle_c, name, func):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    psiT, _ = oracle_c.forward(g["H0"], g["Hc"], g["tlist"], g["eps"], g["psi0"], store=False)
    assert np.max(np.abs(psiT - g["psiT0"])) < 1e-12
    eps_new, ga, psiT1, X, _ = oracle_c.iteration(g["H0"], g["Hc"], g["tlist"], g["eps"], g["S"], g["lam"],
                                                  g["psi0"], g["chiT"])
    assert np.max(np.abs(eps_new - g["eps_new"])) < 1e-12
    assert np.max(np.abs(ga - g["g_a_int"])) < 1e-13
    assert np.max(np.abs(psiT1 - g["psiT1"])) < 1e-11
    assert np.max(np.abs(X[0] - g["X0"])) < 1e-12


def test_discretisation_round_trip():
    tlist = np.linspace(0, 5, 11)
    f = lambda t: 0.2 * o.flattop(t, T=5.0, t_rise=1.0)
    p = o.discretize_on_midpoints(f, tlist)
    assert len(p) == 10 and p[0] == f(0.0) and p[-1] == f(5.0)
    c = o.discretize(p, tlist)
    assert len(c) == 11 and c[0] == p[0] and c[-1] == p[-1] and c[3] == 0.5 * (p[2] + p[3])
    q = o.discretize_on_midpoints(p, tlist)
    assert q is not p and np.array_equal(q, p)  # a copy (test_pulse_optimization.jl:42)
    # length N_T + 1: first and last interval pinned to the end points, interior by the recursion [UPSTREAM-RECALLED]:
    # exact on a round trip through discretize() (what the hand-over tests test_tls_optimization.jl:126,160 need) ...
    rng = np.random.default_rng(3)
    pulse = rng.normal(size=10)
    back = o.discretize_on_midpoints(o.discretize(pulse, tlist), tlist)
    assert np.max(np.abs(back - pulse)) < 1e-14 and back[0] == pulse[0] and back[-1] == pulse[-1]
    # ... and on an arbitrary array of point values the last interval takes the last point's value, not the recursion's
    ctrl = rng.normal(size=11)
    v = o.discretize_on_midpoints(ctrl, tlist)
    assert v[0] == ctrl[0] and v[-1] == ctrl[-1]
    for i in range(1, 9):
        assert v[i] == 2.0 * ctrl[i] - v[i - 1]
    import krotov_jl_amd as kr      # the host mirror restates the same function (the package never imports the oracle)
    assert np.array_equal(kr.discretize_on_midpoints(ctrl, tlist), v)
