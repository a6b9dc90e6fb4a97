"""GPU parity at the edges of the plan selection: state dimensions either side of the padded sizes (8, 16, 32, 64, 128, 256),
trajectory counts either side of the CU count and of the matrix-core tiles, control counts up to and past the register-resident
limit (L = 4), one- and two-step grids, several generators with absent control operators -- each against the CPU oracle
(/root/reference/src/optimize.jl:247-371 restated) at the tolerances of test_gpu_parity.py.  Grids are short so that the oracle
finishes in seconds; which kernel family ran is recorded in the assertion message."""
import numpy as np
import pytest

import krotov_jl_amd as kr
from krotov_jl_amd import workloads as wl

pytestmark = pytest.mark.gpu

TOL_STATE = 1e-11
TOL_PULSE = 1e-10


def _check(oc, p, **kw):
    K, N = p["psi0"].shape
    eng = kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K, **kw)
    info = eng.info()
    tag = "N=%d K=%d L=%d NT=%d %s" % (N, K, len(p["lam"]), len(p["tlist"]) - 1, {k: info[k] for k in (
        "workgroups", "KT", "matrix_core_sweep", "rowsplit_R", "generic_update", "precomputed_propagators")})
    eng.set_initial_states(p["psi0"])
    psiT = eng.forward(p["eps"])
    psiT_o, _ = oc.forward(p["H0"], p["Hc"], p["tlist"], p["eps"], p["psi0"])
    assert np.max(np.abs(psiT - psiT_o)) < TOL_STATE, tag
    _, _, chiT = wl.chi_and_JT(p["functional"], psiT_o, p["tgt"])
    eps_new, ga, psiT1 = eng.iteration(p["eps"], chiT, p["S"], p["lam"])
    X_o = oc.backward(p["H0"], p["Hc"], p["tlist"], p["eps"], chiT)
    eps_o, ga_o, psiT1_o, _ = oc.update_forward(p["H0"], p["Hc"], p["tlist"], p["eps"], p["S"], p["lam"], p["psi0"], X_o)
    scale = max(np.max(np.abs(eps_o)), 1e-300)
    assert np.max(np.abs(eps_new - eps_o)) / scale < TOL_PULSE, tag
    assert np.max(np.abs(ga - ga_o)) < 1e-9 * np.max(np.abs(ga_o)) + 1e-24, tag
    assert np.max(np.abs(psiT1 - psiT1_o)) < 1e-9, tag
    # the last co-state written (t_0) and one in the middle
    NT = len(p["tlist"]) - 1
    for n in {0, NT // 2}:
        assert np.max(np.abs(eng.read_storage("bw", K - 1, n) - X_o[n, K - 1])) < TOL_STATE * max(1.0, np.max(np.abs(X_o))), tag
    eng.close()
    return info


@pytest.mark.parametrize("N", [1, 2, 7, 9, 15, 17, 31, 33, 63, 65, 127, 129, 200, 255, 256])
def test_state_dimension_edges(oracle_c, N):
    K = 5 if N > 64 else 37
    p = wl.dense_problem(N, 4, K, 2, 900 + N, 0.05, 2.0)
    _check(oracle_c, p)


@pytest.mark.parametrize("K", [1, 2, 255, 256, 257, 300, 511, 512, 513, 768, 1023, 1025, 1500, 2047, 2048, 2049, 4096, 4097])
def test_trajectory_count_edges(oracle_c, K):
    """Either side of one trajectory per CU, of the matrix-core tiles (256 x 2, 256 x 4), of several sets per workgroup
    (256 x 8, 256 x 16) and past them (generic kernels)."""
    p = wl.dense_problem(64, 3, K, 2, 1900 + K, 0.05, 2.0)
    _check(oracle_c, p)


@pytest.mark.parametrize("K", [255, 257, 600, 1100])
def test_trajectory_count_edges_small_n(oracle_c, K):
    p = wl.dense_problem(16, 5, K, 1, 2900 + K, 0.1, 1.0)
    _check(oracle_c, p)


@pytest.mark.parametrize("L", [1, 2, 3, 4, 5, 8])
@pytest.mark.parametrize("K", [6, 520])
def test_control_count_edges(oracle_c, L, K):
    p = wl.dense_problem(64, 3, K, L, 3900 + 10 * L + (K > 100), 0.05, 2.0)
    _check(oracle_c, p)


@pytest.mark.parametrize("NT", [1, 2, 3])
def test_shortest_grids(oracle_c, NT):
    for (N, K) in [(4, 3), (64, 256), (64, 512), (200, 4)]:
        p = wl.dense_problem(N, NT, K, 2, 4900 + NT + N, 0.05, 2.0)
        _check(oracle_c, p)


@pytest.mark.parametrize("N,K,G", [(8, 7, 3), (64, 300, 2), (64, 600, 3), (130, 6, 2)])
def test_generators_with_absent_controls(oracle_c, N, K, G):
    """Distinct generators per trajectory (workspace.jl:74) with control operators that are `nothing` for some of them
    (optimize.jl:344), on every kernel family.  The oracle runs per trajectory with that trajectory's generator; the update
    integrands are summed over all trajectories by hand (serial restatement of optimize.jl:328-370)."""
    import krotov_oracle as o
    rng = np.random.default_rng(5900 + N + K)
    L, NT = 2, 3
    p = wl.dense_problem(N, NT, K, L, 5900 + N, 0.05, 2.0)
    H0 = np.array([wl.random_hermitian(rng, N) for _ in range(G)])
    Hc = np.array([[wl.random_hermitian(rng, N) for _ in range(L)] for _ in range(G)])
    present = np.ones((G, L), np.uint8)
    present[G - 1, 1] = 0
    Hc[G - 1, 1] = 0.0
    tg = rng.integers(0, G, size=K).astype(np.int32)
    tg[:G] = np.arange(G)
    tl, eps, S, lam, psi0, tgt = p["tlist"], p["eps"], p["S"], p["lam"], p["psi0"], p["tgt"]
    with kr.HipEngine(H0, Hc, tl, K, traj_generator=tg, hc_present=present) as eng:
        eng.set_initial_states(psi0)
        psiT = eng.forward(eps)
        U = np.array([[o.step_propagator(H0[g], Hc[g], eps[:, n], tl[n + 1] - tl[n]) for n in range(NT)] for g in range(G)])
        psiT_o = psi0.copy()
        for n in range(NT):
            psiT_o = np.einsum("kij,kj->ki", U[tg, n], psiT_o)
        assert np.max(np.abs(psiT - psiT_o)) < TOL_STATE
        _, _, chiT = wl.chi_and_JT("ss", psiT_o, tgt)
        eps_new, ga, psiT1 = eng.iteration(eps, chiT, S, lam)
        info = eng.info()
    X = np.empty((NT + 1, K, N), complex)
    X[NT] = chiT
    for n in range(NT - 1, -1, -1):
        X[n] = np.einsum("kji,kj->ki", U[tg, n].conj(), X[n + 1])
    psi = psi0.copy()
    eps_o = eps.copy()
    for n in range(NT):
        du = np.zeros(L)
        for l in range(L):
            w = np.einsum("kij,kj->ki", Hc[tg, l], psi)
            du[l] = np.sum(np.einsum("ki,ki->k", X[n].conj(), w).imag * present[tg, l])
        eps_o[:, n] = eps[:, n] + S[:, n] / lam * du
        Un = np.array([o.step_propagator(H0[g], Hc[g], eps_o[:, n], tl[n + 1] - tl[n]) for g in range(G)])
        psi = np.einsum("kij,kj->ki", Un[tg], psi)
    assert np.max(np.abs(eps_new - eps_o)) / np.max(np.abs(eps_o)) < TOL_PULSE, info
    assert np.max(np.abs(psiT1 - psi)) < 1e-10, info
