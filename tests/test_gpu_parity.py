"""GPU parity: libkrotov_hip.so (through the C ABI) against the CPU oracle on the same seeded
inputs.  Floating point (fp64 complex): the engine applies exp(-iH dt) as a truncated Taylor
polynomial (truncation bound 1e-16 per step), the oracle uses Pade expm, and the global sum over
trajectories is taken in a different (fixed) order than the reference's serial k loop
(/root/reference/src/optimize.jl:340-349), so the stated tolerances are
    states / co-states : 1e-11 absolute on unit-norm vectors
    updated pulses     : 1e-10 relative to max|eps|
"""
import numpy as np
import pytest

import krotov_jl_amd as kr
from krotov_jl_amd import workloads as wl

pytestmark = pytest.mark.gpu

TOL_STATE = 1e-11
TOL_PULSE = 1e-10


def _rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def _run_case(oc, p, check_storage=True):
    K, N = p["psi0"].shape
    eng = kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K)
    eng.set_initial_states(p["psi0"])
    # a1: initial forward sweep
    psiT = eng.forward(p["eps"])
    psiT_o, Phi_o = oc.forward(p["H0"], p["Hc"], p["tlist"], p["eps"], p["psi0"])
    assert np.max(np.abs(psiT - psiT_o)) < TOL_STATE
    tau, JT, chiT = wl.chi_and_JT(p["functional"], psiT_o, p["tgt"])
    # a3: backward sweep
    eng.backward(p["eps"], chiT)
    X_o = oc.backward(p["H0"], p["Hc"], p["tlist"], p["eps"], chiT)
    if check_storage:
        X = eng.storage("bw")
        assert np.max(np.abs(X - X_o)) < TOL_STATE * max(1.0, np.max(np.abs(X_o)))
        Phi = eng.storage("fw")
        assert np.max(np.abs(Phi - Phi_o)) < TOL_STATE
    # a4-a7: update sweep
    eps_new, ga, psiT1 = eng.update_forward(p["eps"], p["S"], p["lam"])
    eps_o, ga_o, psiT1_o, Phi1_o = oc.update_forward(p["H0"], p["Hc"], p["tlist"], p["eps"], p["S"], p["lam"],
                                                     p["psi0"], X_o)
    assert _rel(eps_new, eps_o) < TOL_PULSE
    assert np.max(np.abs(ga - ga_o)) < 1e-9 * np.max(np.abs(ga_o)) + 1e-24  # (N = 1: the integrand is rounding noise)
    assert np.max(np.abs(psiT1 - psiT1_o)) < 1e-9
    # krotov_iteration in one call gives the same answer
    eps_new2, ga2, psiT2 = eng.iteration(p["eps"], chiT, p["S"], p["lam"])
    assert np.array_equal(eps_new2, eps_new)
    assert np.array_equal(psiT2, psiT1)
    info = eng.info()
    eng.close()
    return info


def test_tls_c1(oracle_c):
    _run_case(oracle_c, wl.config("C1"))


def test_small_dense_cases(oracle_c):
    for (N, NT, K, L, seed) in [(3, 20, 1, 1, 1), (5, 30, 3, 2, 2), (8, 25, 7, 1, 3), (12, 20, 9, 3, 4),
                                (16, 40, 5, 2, 5), (20, 15, 6, 4, 6), (33, 12, 3, 2, 7), (64, 10, 5, 2, 8)]:
        p = wl.dense_problem(N, NT, K, L, seed, 0.1, 2.0)
        _run_case(oracle_c, p)


def test_nonhermitian_and_nonuniform_grid(oracle_c):
    rng = np.random.default_rng(11)
    p = wl.dense_problem(10, 30, 4, 2, 12, 0.1, 2.0)
    p["H0"] = p["H0"] - 0.05j * np.diag(rng.uniform(0, 1, 10)) + 0.02 * rng.normal(size=(10, 10))
    p["tlist"] = np.cumsum(np.r_[0.0, rng.uniform(0.05, 0.2, 30)])
    _run_case(oracle_c, p)


def test_large_norm_substeps(oracle_c):
    """||H dt|| > 1 forces Taylor sub-stepping."""
    p = wl.dense_problem(8, 12, 3, 1, 21, 0.1, 50.0)
    p["H0"] = p["H0"] * 40.0
    _run_case(oracle_c, p)


def test_c2_full(oracle_c):
    info = _run_case(oracle_c, wl.config("C2"), check_storage=False)
    assert info["NP"] == 16


def test_c3_reduced_time(oracle_c):
    """C3's operators and ensemble (N=64, K=256, L=2) on a shortened grid the oracle finishes in seconds."""
    p = wl.dense_problem(64, 40, 256, 2, 64001, 0.05, 5.0)
    info = _run_case(oracle_c, p, check_storage=False)
    assert info["NP"] == 64 and info["workgroups"] == 256


def test_golden_fixtures_on_gpu():
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for name in ("hermitian_n16", "nonhermitian_n8"):
        g = np.load(os.path.join(gold, name + ".npz"))
        K = g["psi0"].shape[0]
        with kr.HipEngine(g["H0"], g["Hc"], g["tlist"], K) as eng:
            eng.set_initial_states(g["psi0"])
            assert np.max(np.abs(eng.forward(g["eps"]) - g["psiT0"])) < TOL_STATE
            eps_new, ga, psiT1 = eng.iteration(g["eps"], g["chiT"], g["S"], g["lam"])
            assert _rel(eps_new, g["eps_new"]) < TOL_PULSE
            assert _rel(ga, g["g_a_int"]) < 1e-9
            assert np.max(np.abs(psiT1 - g["psiT1"])) < 1e-10
            NT = len(g["tlist"]) - 1
            for k in range(K):
                assert np.max(np.abs(eng.read_storage("bw", k, 0) - g["X0"][k])) < TOL_STATE
                assert np.max(np.abs(eng.read_storage("bw", k, NT // 2) - g["X_mid"][k])) < TOL_STATE
                assert np.max(np.abs(eng.read_storage("fw", k, NT) - psiT1[k])) == 0.0


def test_tls_optimization_through_driver_on_gpu():
    """test_tls_optimization.jl:47-70 through the host mirror of optimize(problem; method=Krotov) with the
    HIP engine: the reference's own thresholds, and the TLS golden pulses."""
    import io
    import os
    eps = lambda t: 0.2 * kr.flattop(t, T=5, t_rise=0.3)
    H = kr.hamiltonian(-0.5 * np.array([[1, 0], [0, -1]], complex), (np.array([[0, 1], [1, 0]], complex), eps))
    tlist = np.linspace(0, 5, 501)
    traj = kr.Trajectory(np.array([1, 0], complex), H, target_state=np.array([0, 1], complex))
    snapshots = {}

    def grab(wrk, it, eps_ip1, eps_i):
        snapshots[it] = eps_ip1[0].copy()
    problem = kr.ControlProblem([traj], tlist, iter_stop=5, J_T=kr.J_T_sm)
    res = kr.optimize(problem, method="Krotov", callback=grab, print_iters_out=io.StringIO())
    assert res.J_T < 1e-3
    assert 1.0 < np.max(np.abs(res.optimized_controls[0])) < 1.2
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tls_kat.npz"))
    assert np.max(np.abs(snapshots[1] - g["pulse_iter1"])) < 1e-10
    assert np.max(np.abs(snapshots[5] - g["pulse_iter5"])) < 1e-8
    assert abs(res.J_T - g["J_T"][5]) < 1e-10


def test_device_functional_chain_matches_host_chain():
    """kr_iteration_functional_dev (tau, chi(T) on the device, pulses in HBM) against the host-chi path,
    for J_T_sm / J_T_ss / J_T_re (optimize.jl:297-302,381-386)."""
    import torch
    for fname, fid in (("sm", 0), ("ss", 1), ("re", 2)):
        p = wl.dense_problem(12, 30, 10, 2, 31 + fid, 0.1, 0.5)
        K = 10
        with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K) as a, kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K) as b:
            for e in (a, b):
                e.set_initial_states(p["psi0"])
                e.set_target_states(p["tgt"])
            # host chain
            eps = p["eps"].copy()
            psiT = a.forward(eps)
            for _ in range(3):
                tau, JT, chiT = wl.chi_and_JT(fname, psiT, p["tgt"])
                eps, ga, psiT = a.iteration(eps, chiT, p["S"], p["lam"])
            # device chain
            b.set_update_params(p["S"], p["lam"])
            e0 = torch.from_numpy(p["eps"].copy()).cuda()
            e1 = torch.empty_like(e0)
            torch.cuda.synchronize()
            b.forward_dev(e0.data_ptr())
            for _ in range(3):
                b.iteration_functional_dev(fid, e0.data_ptr(), e1.data_ptr())
                e0, e1 = e1, e0
            b.synchronize()
            assert _rel(e0.cpu().numpy(), eps) < TOL_PULSE
            assert np.max(np.abs(b.final_states() - psiT)) < 1e-10
            assert _rel(b.g_a_int(), ga) < 1e-9
            tau_d = b.taus()
            assert np.max(np.abs(tau_d - wl.chi_and_JT(fname, psiT, p["tgt"])[0])) < 1e-11


def test_absent_control_derivative_is_skipped(oracle_c):
    """mu_kl === nothing (optimize.jl:344): that control gets no update."""
    p = wl.dense_problem(8, 20, 3, 2, 55, 0.1, 1.0)
    K = 3
    mask = np.array([1, 0], np.uint8)
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K, hc_present=mask) as eng:
        eng.set_initial_states(p["psi0"])
        psiT = eng.forward(p["eps"])
        _, _, chiT = wl.chi_and_JT("ss", psiT, p["tgt"])
        eps_new, ga, _ = eng.iteration(p["eps"], chiT, p["S"], p["lam"])
    assert np.array_equal(eps_new[1], p["eps"][1]) and ga[1] == 0.0
    X = oracle_c.backward(p["H0"], p["Hc"], p["tlist"], p["eps"], chiT)
    eps_o = oracle_c.update_forward(p["H0"], p["Hc"], p["tlist"], p["eps"], p["S"], p["lam"], p["psi0"], X, mask=mask)[0]
    assert _rel(eps_new, eps_o) < TOL_PULSE


def test_call_order_errors():
    p = wl.dense_problem(4, 5, 2, 1, 1, 0.1, 1.0)
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], 2) as eng:
        with pytest.raises(kr.KrotovHipError):
            eng.forward(p["eps"])                       # initial states not set
        eng.set_initial_states(p["psi0"])
        with pytest.raises(kr.KrotovHipError):
            eng.update_forward(p["eps"], p["S"], p["lam"])  # no backward sweep yet
        with pytest.raises(kr.KrotovHipError):
            eng.read_storage("bw", 0, 99)


def test_full_size_properties_c3():
    """BASELINE size (N=64, N_T=1000, K=256, L=2): size-independent properties instead of the oracle --
    norm conservation of every trajectory (Hermitian generator), <chi(t_n)|psi(t_n)> constant along the
    grid, J_T decreases monotonically over three iterations, run-to-run bitwise reproducibility."""
    p = wl.config("C3")
    K = 256
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K) as eng:
        eng.set_initial_states(p["psi0"])
        psiT = eng.forward(p["eps"])
        assert np.max(np.abs(np.linalg.norm(psiT, axis=1) - 1.0)) < 1e-12
        eng.backward(p["eps"], p["tgt"])
        for n in (0, 333, 1000):
            ov = [np.vdot(eng.read_storage("bw", k, n), eng.read_storage("fw", k, n)) for k in (0, 17, 255)]
            ovT = [np.vdot(p["tgt"][k], psiT[k]) for k in (0, 17, 255)]
            assert np.max(np.abs(np.array(ov) - np.array(ovT))) < 1e-12
        lam = np.full(2, 0.02)
        eps = p["eps"].copy()
        J = [wl.chi_and_JT("ss", psiT, p["tgt"])[1]]
        first = None
        for it in range(3):
            _, _, chiT = wl.chi_and_JT("ss", psiT, p["tgt"])
            eps_new, ga, psiT = eng.iteration(eps, chiT, p["S"], lam)
            if it == 0:
                again = eng.iteration(eps, chiT, p["S"], lam)
                assert np.array_equal(again[0], eps_new) and np.array_equal(again[2], psiT)
            eps = eps_new
            J.append(wl.chi_and_JT("ss", psiT, p["tgt"])[1])
            assert np.max(np.abs(np.linalg.norm(psiT, axis=1) - 1.0)) < 1e-12
        assert J[3] < J[2] < J[1] < J[0]


@pytest.mark.parametrize("rowsplit", ["1", "0"])
def test_large_state_dimension(oracle_c, monkeypatch, rowsplit):
    """64 < N <= 256, Hermitian and Liouvillian: row-split kernels (R workgroups share a trajectory and all-gather
    the vector after every matvec round, kr_device_rowsplit.h) and the one-workgroup-per-trajectory streamed
    kernels (kr_device_stream.h)."""
    monkeypatch.setenv("KROTOV_HIP_ROWSPLIT", rowsplit)
    for (N, NT, K, L, seed) in [(65, 6, 2, 1, 41), (100, 5, 3, 2, 42), (128, 4, 2, 1, 43), (200, 3, 2, 3, 44),
                                (130, 3, 40, 1, 45), (256, 3, 60, 2, 46), (256, 2, 70, 1, 48), (200, 2, 70, 2, 49)]:
        p = wl.dense_problem(N, NT, K, L, seed, 0.05, 2.0)
        info = _run_case(oracle_c, p, check_storage=(K < 10))
        assert info["NP"] in (128, 256)
        # K = 70 at NP = 256 needs 280 workgroups: with two controls the streamed kernels take over, with ONE control the
        # pair-mode groups carry two pairs each in turn (round 3)
        fits = K * (info["NP"] // 64) <= 256 or (info["NP"] == 256 and L <= 2 and K <= 256)   # (one control: pair mode with sets; two: turns)
        assert (info["rowsplit_R"] > 0) == (rowsplit == "1" and fits)
    p = wl.liouvillian_problem(n_levels=16, NT=4, K=3, seed=256001)   # C5's operators, short grid
    info = _run_case(oracle_c, p)
    assert info["NP"] == 256


@pytest.mark.parametrize("N,NT,K,L", [(100, 4, 130, 2), (128, 3, 256, 1), (72, 3, 200, 2)])
def test_one_workgroup_per_trajectory_with_resident_generator(oracle_c, N, NT, K, L):
    """64 < N <= 128 with more trajectories than HALF the CUs (round 3): the row-split kernels with R = 1 -- one workgroup per
    trajectory, H(eps) resident, generator blocks streamed per STEP -- instead of the streamed-generator kernels (per ROUND).
    Same sweeps (/root/reference/src/optimize.jl:247-265, 303-317, 321-370) against the oracle; L >= 3 stays with the streamed
    kernels (the four-control instantiation spills)."""
    p = wl.dense_problem(N, NT, K, L, 4400 + N + K, 0.05, 2.0)
    info = _run_case(oracle_c, p, check_storage=False)
    assert info["NP"] == 128 and info["rowsplit_R"] == 1 and info["workgroups"] == K, info
    p3 = wl.dense_problem(100, 3, 140, 3, 4501, 0.05, 2.0)
    with kr.HipEngine(p3["H0"], p3["Hc"], p3["tlist"], 140) as eng:
        assert eng.info()["rowsplit_R"] == 0     # three controls: streamed-generator kernels


@pytest.mark.parametrize("levels,NT,K,sets", [(12, 3, 70, 2), (12, 2, 140, 3), (16, 2, 250, 4)])
def test_pair_mode_with_several_pairs_per_group(oracle_c, levels, NT, K, sets):
    """128 < N <= 256, one control, shared generator and more pairs than groups of 8 workgroups (round 3): every group carries
    2..4 pairs in turn (kr_rowsplit_pair_sweep<.., MULTI>) -- known-pulse sweeps set after set, update sweep with ONE exchange per
    step over all sets (/root/reference/src/optimize.jl:339-349 sums over ALL trajectories) -- instead of the streamed-generator
    kernels.  Open-system generators (C5's construction, /root/reference/src/workspace.jl:69,149-161: true adjoint backwards)."""
    p = wl.liouvillian_problem(n_levels=levels, NT=NT, K=K, seed=9100 + K)
    info = _run_case(oracle_c, p, check_storage=False)
    assert info["NP"] == 256 and info["rowsplit_R"] == 8 and info["KT"] == 2 and info["workgroups"] == 256, info
    assert info["K_padded"] == 64 * sets, info


def test_rowsplit_with_substeps_and_generators(oracle_c):
    """Row-split kernels with ||H dt|| > 1 (sub-stepping: every sub-step gathers) and G > 1."""
    p = wl.dense_problem(96, 4, 3, 1, 47, 0.1, 50.0)
    p["H0"] = p["H0"] * 25.0
    info = _run_case(oracle_c, p)
    assert info["rowsplit_R"] > 0


def test_unsupported_sizes_fail_loudly(monkeypatch):
    """What no kernel of this build carries fails with KR_ERR_UNSUPPORTED and a message: more than 32 controls; with the
    generic fallback forbidden (KROTOV_HIP_GENERIC=0) also N > 1024 (round 3: the row-split kernels carry N <= 1024)."""
    with pytest.raises(kr.KrotovHipError) as ei:
        kr.HipEngine(np.eye(4), np.tile(np.eye(4), (33, 1, 1)), np.linspace(0, 1, 3), 1)
    assert ei.value.code == kr._capi.KR_ERR_UNSUPPORTED
    monkeypatch.setenv("KROTOV_HIP_GENERIC", "0")
    with pytest.raises(kr.KrotovHipError) as ei:
        kr.HipEngine(np.eye(1100), np.eye(1100)[None], np.linspace(0, 1, 3), 1)
    assert ei.value.code == kr._capi.KR_ERR_UNSUPPORTED and "N = 1100" in str(ei.value)


@pytest.mark.parametrize("herm", ["0", "2"])
def test_per_trajectory_generators(oracle_c, monkeypatch, herm):
    """G > 1: an ensemble of different H_k (control_derivs[k][l], workspace.jl:74; optimize.jl:343).  The oracle
    is run per generator group and the update integrands are summed over ALL trajectories by hand.  Both update-kernel
    variants (general / Hermitian control operators)."""
    import krotov_oracle as o
    monkeypatch.setenv("KROTOV_HIP_HERM", herm)
    rng = np.random.default_rng(77)
    N, NT, L = 12, 25, 2
    G, K = 3, 8
    tg = np.array([0, 2, 1, 0, 2, 2, 1, 0], np.int32)
    base = wl.dense_problem(N, NT, K, L, 99, 0.1, 1.5)
    H0 = np.array([base["H0"] * (1.0 + 0.1 * g) + 0.05 * wl.random_hermitian(rng, N) for g in range(G)])
    Hc = np.array([[base["Hc"][l] * (1.0 - 0.07 * g) for l in range(L)] for g in range(G)])
    present = np.ones((G, L), np.uint8)
    present[1, 1] = 0   # generator 1 does not depend on control 1
    tl, eps, S, lam, psi0, tgt = base["tlist"], base["eps"], base["S"], base["lam"], base["psi0"], base["tgt"]
    with kr.HipEngine(H0, Hc, tl, K, traj_generator=tg, hc_present=present) as eng:
        eng.set_initial_states(psi0)
        psiT = eng.forward(eps)
        # oracle forward per trajectory with its own generator
        psiT_o = np.array([o.forward(H0[tg[k]], Hc[tg[k]], tl, eps, psi0[k:k + 1])[0][0] for k in range(K)])
        assert np.max(np.abs(psiT - psiT_o)) < TOL_STATE
        _, _, chiT = wl.chi_and_JT("ss", psiT_o, tgt)
        eps_new, ga, psiT1 = eng.iteration(eps, chiT, S, lam)
        X0 = eng.read_storage("bw", 5, 0)
    # oracle update sweep with per-trajectory generators (serial restatement of optimize.jl:328-370)
    X = [o.backward(H0[tg[k]], Hc[tg[k]], tl, eps, chiT[k:k + 1])[:, 0] for k in range(K)]
    assert np.max(np.abs(X0 - X[5][0])) < TOL_STATE
    psi = psi0.copy()
    eps_o = eps.copy()
    for n in range(NT):
        du = np.zeros(L)
        for l in range(L):
            for k in range(K):
                if present[tg[k], l]:
                    du[l] += np.vdot(X[k][n], Hc[tg[k], l] @ psi[k]).imag
        eps_o[:, n] = eps[:, n] + S[:, n] / lam * du
        for k in range(K):
            psi[k] = o.step_propagator(H0[tg[k]], Hc[tg[k]], eps_o[:, n], tl[n + 1] - tl[n]) @ psi[k]
    assert _rel(eps_new, eps_o) < TOL_PULSE
    assert np.max(np.abs(psiT1 - psi)) < 1e-10


def test_edge_shapes(oracle_c):
    """Smallest and ragged shapes: N = 1, 2; N_T = 1; L = 4; K that does not fill the last tile."""
    for (N, NT, K, L, seed) in [(1, 5, 2, 1, 61), (2, 1, 1, 1, 62), (7, 1, 5, 2, 63), (9, 6, 13, 4, 64),
                                (31, 4, 3, 3, 65), (64, 3, 1, 4, 66), (17, 5, 70, 1, 67)]:
        p = wl.dense_problem(N, NT, K, L, seed, 0.1, 2.0)
        _run_case(oracle_c, p, check_storage=(K * NT < 200))


def test_many_trajectories_per_workgroup(oracle_c):
    """K = 600 at N = 64 forces KT > 1 (several trajectories per team) and a ragged last tile."""
    p = wl.dense_problem(64, 6, 600, 2, 71, 0.05, 5.0)
    info = _run_case(oracle_c, p, check_storage=False)
    assert info["KT"] >= 2 and info["K_padded"] >= 600


def test_taylor_sweeps_equal_precomputed_propagators(oracle_c, monkeypatch):
    """The known-pulse sweeps have two implementations (Taylor rounds per step / batched propagators + one round
    per step); both must agree with the oracle and with each other."""
    p = wl.dense_problem(20, 30, 6, 2, 81, 0.1, 2.0)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("KROTOV_HIP_PRECOMP", mode)
        with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], 6) as eng:
            assert eng.info()["precomputed_propagators"] == int(mode)
            eng.set_initial_states(p["psi0"])
            psiT = eng.forward(p["eps"])
            eng.backward(p["eps"], p["tgt"])
            res[mode] = (psiT, eng.storage("bw"))
    psiT_o, _ = oracle_c.forward(p["H0"], p["Hc"], p["tlist"], p["eps"], p["psi0"])
    X_o = oracle_c.backward(p["H0"], p["Hc"], p["tlist"], p["eps"], p["tgt"])
    for mode in ("1", "0"):
        assert np.max(np.abs(res[mode][0] - psiT_o)) < TOL_STATE
        assert np.max(np.abs(res[mode][1] - X_o)) < TOL_STATE
    assert np.max(np.abs(res["1"][1] - res["0"][1])) < 1e-13


def test_no_forward_storage_and_tolerance_option(oracle_c):
    p = wl.dense_problem(10, 20, 4, 1, 91, 0.1, 2.0)
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], 4, store_fw=False, prop_tol=1e-10) as eng:
        eng.set_initial_states(p["psi0"])
        psiT = eng.forward(p["eps"])
        psiT_o, _ = oracle_c.forward(p["H0"], p["Hc"], p["tlist"], p["eps"], p["psi0"], store=False)
        err = np.max(np.abs(psiT - psiT_o))
        assert err < 1e-8          # looser propagator tolerance, still small
        with pytest.raises(kr.KrotovHipError):
            eng.read_storage("fw", 0, 1)


def test_ensemble_through_driver_on_gpu():
    """G > 1 through optimize(problem; method=Krotov): three detuned TLS + one more trajectory, HIP engine."""
    import io
    eps = lambda t: 0.2 * kr.flattop(t, T=5, t_rise=0.3)
    H1 = np.array([[0, 1], [1, 0]], complex)
    tlist = np.linspace(0, 5, 101)
    trajs = []
    for om in (0.9, 1.0, 1.1):
        H = kr.hamiltonian(-0.5 * om * np.array([[1, 0], [0, -1]], complex), (H1, eps))
        trajs.append(kr.Trajectory(np.array([1, 0], complex), H, target_state=np.array([0, 1], complex)))
    trajs.append(kr.Trajectory(np.array([0, 1], complex), trajs[1].generator, target_state=np.array([1, 0], complex)))
    problem = kr.ControlProblem(trajs, tlist, iter_stop=4, J_T=kr.J_T_ss)
    res = kr.optimize(problem, store_iter_info=["J_T"], print_iters_out=io.StringIO())
    J = [r[0] for r in res.records]
    assert all(b < a for a, b in zip(J, J[1:])) and res.converged


def test_n64_large_norm_scaling_and_squaring(oracle_c, monkeypatch):
    """N = 64 with ||H dt|| ~ 6: the matrix-core propagator build (kr_expm_mfma64) has to scale and square; the
    update sweep has to sub-step.  Also run with the VALU build kernel."""
    p = wl.dense_problem(64, 6, 5, 2, 101, 0.1, 50.0)
    p["H0"] = p["H0"] * 60.0
    for mfma in ("1", "0"):
        monkeypatch.setenv("KROTOV_HIP_EXPM_MFMA", mfma)
        _run_case(oracle_c, p, check_storage=True)


def test_rowsplit_mailbox_tag_wrap(oracle_c):
    """More than 256 launches on one handle: the row-split vector mailbox's launch tag wraps and the mailbox is
    cleared; results must stay identical from launch to launch."""
    p = wl.dense_problem(70, 3, 2, 1, 49, 0.05, 2.0)
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], 2) as eng:
        assert eng.info()["rowsplit_R"] > 0
        eng.set_initial_states(p["psi0"])
        first = eng.forward(p["eps"])
        for _ in range(300):
            again = eng.forward(p["eps"])
        assert np.array_equal(again, first)
    psiT_o, _ = oracle_c.forward(p["H0"], p["Hc"], p["tlist"], p["eps"], p["psi0"], store=False)
    assert np.max(np.abs(first - psiT_o)) < TOL_STATE


@pytest.mark.parametrize("kt", ["1", "2", "4"])
def test_trajectories_per_team_variants(oracle_c, monkeypatch, kt):
    """Every KT instantiation (trajectories per team) at every padded size, forced through KROTOV_HIP_KT."""
    monkeypatch.setenv("KROTOV_HIP_KT", kt)
    for (N, NT, K, L, seed) in [(6, 8, 21, 1, 121), (13, 6, 11, 2, 122), (30, 5, 9, 1, 123), (50, 4, 7, 3, 124)]:
        p = wl.dense_problem(N, NT, K, L, seed, 0.1, 2.0)
        info = _run_case(oracle_c, p, check_storage=True)
        assert info["KT"] == int(kt)


def test_exchange_schedules_agree_bitwise(oracle_c, monkeypatch):
    """The two schedules of the per-step exchange (wave 0 publishes and polls / publisher wave + delayed polling wave,
    delay calibrated at kr_create) change timing only: results must agree bit for bit, and with the oracle."""
    p = wl.dense_problem(40, 30, 80, 2, 131, 0.05, 1.0)  # NP = 64, one trajectory per workgroup: 80 workgroups
    out = {}
    for delay in ["-1", "0", "16", "auto"]:
        if delay == "auto":
            monkeypatch.delenv("KROTOV_HIP_POLL_DELAY", raising=False)
        else:
            monkeypatch.setenv("KROTOV_HIP_POLL_DELAY", delay)
        eng = kr.HipEngine(p["H0"], p["Hc"], p["tlist"], 80)
        info = eng.info()
        assert info["workgroups"] >= 64
        if delay != "auto":
            assert info["exchange_poll_delay"] == int(delay)
        eng.set_initial_states(p["psi0"])
        psiT = eng.forward(p["eps"])
        _, _, chiT = wl.chi_and_JT(p["functional"], psiT, p["tgt"])
        out[delay] = eng.iteration(p["eps"], chiT, p["S"], p["lam"])
        eng.close()
    for delay in ["0", "16", "auto"]:
        for x, y in zip(out[delay], out["-1"]):
            assert np.array_equal(x, y)
    monkeypatch.setenv("KROTOV_HIP_POLL_DELAY", "16")
    _run_case(oracle_c, p)


def test_rowsplit_xcd_grouping_changes_timing_only(oracle_c, monkeypatch):
    """K % 8 == 0: the R workgroups of a trajectory are laid out on one XCD (checked in the kernel through
    HW_REG_XCC_ID) and all-gather through that XCD's L2; any other placement uses the write-through path.  Same bits."""
    out = {}
    for (N, NT, K, L, seed) in [(130, 4, 8, 1, 141), (100, 3, 16, 2, 142)]:   # R = 32 and R = 8..16
        p = wl.dense_problem(N, NT, K, L, seed, 0.05, 2.0)
        for grp in ["0", "1"]:
            monkeypatch.setenv("KROTOV_HIP_XCD_GROUP", grp)
            with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K) as eng:
                assert eng.info()["rowsplit_R"] > 1
                eng.set_initial_states(p["psi0"])
                psiT = eng.forward(p["eps"])
                _, _, chiT = wl.chi_and_JT(p["functional"], psiT, p["tgt"])
                out[grp] = (psiT,) + tuple(eng.iteration(p["eps"], chiT, p["S"], p["lam"]))
        for x, y in zip(out["0"], out["1"]):
            assert np.array_equal(x, y)
        monkeypatch.setenv("KROTOV_HIP_XCD_GROUP", "1")
        _run_case(oracle_c, p, check_storage=True)


@pytest.mark.parametrize("kt", ["1", "2"])
def test_hermitian_controls_path(oracle_c, monkeypatch, kt):
    """Update-kernel variant for Hermitian control operators (Im<chi|H_l|psi> = Im<H_l chi|psi>, phase A moved into the
    exchange window) against the general variant and the oracle, including sub-stepping, absent derivatives, several
    trajectories per team, both exchange schedules; non-Hermitian control operators must select the general variant."""
    monkeypatch.setenv("KROTOV_HIP_KT", kt)
    cases = [wl.dense_problem(40, 12, 70, 2, 151, 0.05, 1.0),    # NP = 64, >= 64 workgroups at KT = 1
             wl.dense_problem(13, 9, 300, 3, 152, 0.1, 2.0),     # NP = 16, LMAX = 4
             wl.dense_problem(5, 7, 3, 1, 153, 0.1, 2.0),        # one workgroup
             wl.dense_problem(24, 6, 40, 2, 154, 0.1, 50.0)]     # large updates
    cases[3]["H0"] = cases[3]["H0"] * 30.0                        # ||H dt|| > 1: sub-steps
    for p in cases:
        K = p["psi0"].shape[0]
        res = {}
        for mode in ["0", "2"]:
            monkeypatch.setenv("KROTOV_HIP_HERM", mode)
            with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K) as eng:
                assert eng.info()["hermitian_controls_path"] == (1 if mode == "2" else 0)
                eng.set_initial_states(p["psi0"])
                psiT = eng.forward(p["eps"])
                _, _, chiT = wl.chi_and_JT(p["functional"], psiT, p["tgt"])
                res[mode] = eng.iteration(p["eps"], chiT, p["S"], p["lam"])
        assert _rel(res["2"][0], res["0"][0]) < TOL_PULSE
        assert np.max(np.abs(res["2"][2] - res["0"][2])) < 1e-10
        monkeypatch.setenv("KROTOV_HIP_HERM", "2")
        _run_case(oracle_c, p)
        monkeypatch.setenv("KROTOV_HIP_POLL_DELAY", "3")
        _run_case(oracle_c, p, check_storage=False)
        monkeypatch.delenv("KROTOV_HIP_POLL_DELAY")
    # without forward storage the Hermitian-controls variant gives the same pulses
    p = cases[0]
    outs = []
    for store in [True, False]:
        with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], 70, store_fw=store) as eng:
            eng.set_initial_states(p["psi0"])
            psiT = eng.forward(p["eps"])
            _, _, chiT = wl.chi_and_JT(p["functional"], psiT, p["tgt"])
            outs.append(eng.iteration(p["eps"], chiT, p["S"], p["lam"]))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][2], outs[1][2])
    # a non-Hermitian control operator: the general variant must be used even when the other one is forced
    p = wl.dense_problem(12, 6, 5, 2, 155, 0.1, 2.0)
    p["Hc"] = p["Hc"].copy()
    p["Hc"][1, 0, 3] += 0.25
    monkeypatch.setenv("KROTOV_HIP_HERM", "2")
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], 5) as eng:
        assert eng.info()["hermitian_controls_path"] == 0
    _run_case(oracle_c, p)


def test_c4_and_c5_shapes_short_grid(oracle_c):
    """BASELINE shapes C4 (N=64, K=1024: four trajectories per team) and C5 (N=256 Liouvillian, K=64: row-split kernels,
    four workgroups per trajectory on one XCD) on a short time grid against the oracle (shared-propagator mode)."""
    p = wl.dense_problem(64, 6, 1024, 2, 64001, 0.05, 5.0)
    info = _run_case(oracle_c, p, check_storage=False)
    assert info["KT"] == 4 and info["workgroups"] == 256
    p = wl.liouvillian_problem(n_levels=16, NT=5, K=64, seed=256001)
    info = _run_case(oracle_c, p, check_storage=False)
    assert info["rowsplit_R"] == 8 and info["KT"] == 2 and info["workgroups"] == 256     # round 3: pair mode (two trajectories per group of 8)


@pytest.mark.parametrize("case", ["liouvillian", pytest.param("hermitian_substeps", marks=pytest.mark.soak), "n130", pytest.param("odd_k", marks=pytest.mark.soak), "np128_local", "np128"])
def test_rowsplit_pair_mode_agrees_with_single_mode_and_oracle(oracle_c, monkeypatch, case):
    """Row-split PAIR mode (kr_device_rowsplit2.h, round 3): groups of 2R workgroups interleave TWO trajectories (one's all-gather
    behind the other's row-dot, H(eps) and the control block resident) where the single mode would leave 64 rows per workgroup.
    Same arithmetic per trajectory, so the two modes must agree far inside the parity tolerances -- known-pulse sweeps (forward,
    true-adjoint backward) and the update sweep, across rebuilds of the H(eps) block (N_T > 32), with sub-stepping, with a state
    dimension that does not fill the padded size -- and pair mode must meet the oracle at them."""
    if case == "liouvillian":
        p = wl.liouvillian_problem(n_levels=16, NT=40, K=64, seed=256001, lam=0.05)
    elif case == "hermitian_substeps":
        p = wl.dense_problem(200, 36, 48, 1, 4200, 0.05, 0.5)
        p["H0"] = p["H0"] * 50.0          # ||H dt|| ~ 2.5: every step is sub-stepped
    elif case == "odd_k":          # 33 trajectories: the 17th pair carries one zero trajectory
        p = wl.dense_problem(256, 34, 33, 1, 194, 0.01, 0.2)
    elif case == "np128_local":    # 64 < N <= 128: single mode R = 2, pair mode groups of 4; 48 pairs -> one XCD per group
        p = wl.dense_problem(100, 36, 96, 1, 195, 0.02, 0.2)
    elif case == "np128":          # 50 pairs: placement-independent path
        p = wl.dense_problem(128, 34, 100, 1, 196, 0.02, 0.2)
    else:
        p = wl.dense_problem(130, 35, 40, 1, 193, 0.02, 0.2)
    r_single = 2 if case.startswith("np128") else 4
    res = {}
    ref = _oracle_iteration(None, p)   # (SciPy twin: one expm per step; each mode meets it BEFORE the modes are compared)
    for mode in ["0", "1"]:
        monkeypatch.setenv("KROTOV_HIP_RS_PAIR", mode)
        res[mode], info = _iterate_against_oracle(ref, p, "%s, KROTOV_HIP_RS_PAIR=%s" % (case, mode))
        assert info["rowsplit_R"] == (2 * r_single if mode == "1" else r_single) and info["KT"] == (2 if mode == "1" else 1), info
    assert _rel(res["1"][0], res["0"][0]) < 1e-12
    assert np.max(np.abs(res["1"][2] - res["0"][2])) < 1e-12 * max(1.0, np.max(np.abs(res["0"][2])))
    monkeypatch.setenv("KROTOV_HIP_RS_PAIR", "1")
    if case in ("n130", "np128_local"):     # (and the C oracle -- the reference's execution order -- on two of the shapes)
        _run_case(oracle_c, p, check_storage=False)


def test_full_size_properties_c5():
    """BASELINE size C5 (N=256 Liouvillian, N_T=2000, K=64): <<chi(t_n)|rho(t_n)>> constant along the grid (the backward
    sweep uses the true adjoint of the non-Hermitian generator), trace of every density matrix conserved, J_T decreases,
    run-to-run bitwise reproducibility of the row-split kernels."""
    p = wl.config("C5")
    K, N = p["psi0"].shape
    n_lev = int(round(np.sqrt(N)))
    tr = np.eye(n_lev).reshape(-1)          # <<1|rho>> = trace
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K) as eng:
        eng.set_initial_states(p["psi0"])
        psiT = eng.forward(p["eps"])
        tr0 = p["psi0"] @ tr
        assert np.max(np.abs(psiT @ tr - tr0)) < 1e-11
        _, J0, chiT = wl.chi_and_JT(p["functional"], psiT, p["tgt"])
        eng.backward(p["eps"], chiT)
        NT = p["eps"].shape[1]
        for k in (0, 31, 63):
            ov = [np.vdot(eng.read_storage("bw", k, n), eng.read_storage("fw", k, n)) for n in (0, NT // 3, NT)]
            assert np.max(np.abs(np.array(ov) - ov[-1])) < 1e-11 * max(1.0, abs(ov[-1]))
        eps_new, ga, psiT1 = eng.iteration(p["eps"], chiT, p["S"], p["lam"])
        again = eng.iteration(p["eps"], chiT, p["S"], p["lam"])
        assert np.array_equal(again[0], eps_new) and np.array_equal(again[2], psiT1)
        assert wl.chi_and_JT(p["functional"], psiT1, p["tgt"])[1] < J0
        assert np.max(np.abs(psiT1 @ tr - tr0)) < 1e-11


@pytest.mark.parametrize("herm", ["0", "2"])
def test_matrix_core_sweep_variants(oracle_c, monkeypatch, herm):
    """The update sweep on the matrix cores (kr_device_mfma.h: v_mfma_f64_4x4x4_4b_f64, two trajectories per 4-column tile)
    forced for every NP = 64 shape: ragged K (half-empty tiles), N < 64 (zero padding), L = 1..4 (LMAX = 4 keeps five
    operand sets in registers), one and two tiles per workgroup, sub-stepping, non-Hermitian drift, absent derivatives,
    per-trajectory generators, both exchange schedules, no forward storage -- general and Hermitian-controls variants."""
    monkeypatch.setenv("KROTOV_HIP_MFMA_SWEEP", "1")
    monkeypatch.setenv("KROTOV_HIP_HERM", herm)
    rng = np.random.default_rng(2024)
    for (N, NT, K, L, seed, kt) in [(64, 9, 1, 1, 201, "0"), (64, 7, 5, 2, 202, "0"), (40, 8, 7, 3, 203, "0"), (33, 6, 3, 4, 204, "0"),
                                    (64, 6, 9, 2, 205, "4"), (50, 5, 6, 4, 206, "4"), (64, 5, 300, 2, 207, "0")]:
        monkeypatch.setenv("KROTOV_HIP_KT", kt)
        p = wl.dense_problem(N, NT, K, L, seed, 0.05, 0.5)
        info = _run_case(oracle_c, p, check_storage=(K < 20))
        assert info["NP"] == 64 and info["threads"] == 256 and info["KT"] in (2, 4)
        assert info["hermitian_controls_path"] == (1 if herm == "2" and not (L > 2 and info["KT"] == 4) else 0)
    monkeypatch.setenv("KROTOV_HIP_KT", "0")
    # ||H dt|| ~ 5: every step is sub-stepped
    p = wl.dense_problem(64, 5, 6, 2, 208, 0.1, 20.0)
    p["H0"] = p["H0"] * 50.0
    _run_case(oracle_c, p)
    # non-Hermitian drift and control operator (the Hermitian-controls variant must not be selected), non-uniform grid
    p = wl.dense_problem(48, 12, 5, 2, 209, 0.05, 0.5)
    p["H0"] = p["H0"] - 0.05j * np.diag(rng.uniform(0, 1, 48)) + 0.02 * rng.normal(size=(48, 48))
    p["Hc"] = p["Hc"].copy()
    p["Hc"][0, 1, 7] += 0.3
    p["tlist"] = np.cumsum(np.r_[0.0, rng.uniform(0.02, 0.08, 12)])
    info = _run_case(oracle_c, p)
    assert info["hermitian_controls_path"] == 0
    # delayed-poll exchange schedule
    monkeypatch.setenv("KROTOV_HIP_POLL_DELAY", "5")
    _run_case(oracle_c, wl.dense_problem(64, 6, 8, 2, 210, 0.05, 0.5), check_storage=False)
    monkeypatch.delenv("KROTOV_HIP_POLL_DELAY")
    # absent derivative + no forward storage
    p = wl.dense_problem(64, 8, 6, 2, 211, 0.05, 0.5)
    mask = np.array([0, 1], np.uint8)
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], 6, hc_present=mask, store_fw=False) as eng:
        eng.set_initial_states(p["psi0"])
        psiT = eng.forward(p["eps"])
        _, _, chiT = wl.chi_and_JT("ss", psiT, p["tgt"])
        eps_new, ga, psiT1 = eng.iteration(p["eps"], chiT, p["S"], p["lam"])
    X = oracle_c.backward(p["H0"], p["Hc"], p["tlist"], p["eps"], chiT)
    eps_o, ga_o, psiT_o, _ = oracle_c.update_forward(p["H0"], p["Hc"], p["tlist"], p["eps"], p["S"], p["lam"], p["psi0"], X, mask=mask)
    assert np.array_equal(eps_new[0], p["eps"][0]) and _rel(eps_new, eps_o) < TOL_PULSE and np.max(np.abs(psiT1 - psiT_o)) < 1e-10
    # per-trajectory generators: three generators over 11 trajectories (tiles never mix generators)
    G, K, N, NT, L = 3, 11, 64, 6, 2
    base = wl.dense_problem(N, NT, K, L, 212, 0.05, 0.5)
    tg = np.array([0, 1, 2, 0, 1, 2, 2, 2, 0, 0, 0], np.int32)
    H0 = np.array([base["H0"] * (1.0 + 0.1 * g) for g in range(G)])
    Hc = np.array([[base["Hc"][l] * (1.0 - 0.05 * g) for l in range(L)] for g in range(G)])
    with kr.HipEngine(H0, Hc, base["tlist"], K, traj_generator=tg) as eng:
        assert eng.info()["threads"] == 256
        eng.set_initial_states(base["psi0"])
        psiT = eng.forward(base["eps"])
        _, _, chiT = wl.chi_and_JT("ss", psiT, base["tgt"])
        eps_new, ga, psiT1 = eng.iteration(base["eps"], chiT, base["S"], base["lam"])
    import krotov_oracle as o
    X = [o.backward(H0[tg[k]], Hc[tg[k]], base["tlist"], base["eps"], chiT[k:k + 1])[:, 0] for k in range(K)]
    psi, eps_o = base["psi0"].copy(), base["eps"].copy()
    for n in range(NT):
        du = np.array([sum(np.vdot(X[k][n], Hc[tg[k], l] @ psi[k]).imag for k in range(K)) for l in range(L)])
        eps_o[:, n] = base["eps"][:, n] + base["S"][:, n] / base["lam"] * du
        for k in range(K):
            psi[k] = o.step_propagator(H0[tg[k]], Hc[tg[k]], eps_o[:, n], base["tlist"][n + 1] - base["tlist"][n]) @ psi[k]
    assert _rel(eps_new, eps_o) < TOL_PULSE and np.max(np.abs(psiT1 - psi)) < 1e-10


def _iterate(p, **kw):
    K = p["psi0"].shape[0]
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K, **kw) as eng:
        eng.set_initial_states(p["psi0"])
        psiT = eng.forward(p["eps"])
        _, _, chiT = wl.chi_and_JT(p["functional"], psiT, p["tgt"])
        return eng.iteration(p["eps"], chiT, p["S"], p["lam"]), eng.info()


def test_economized_step_polynomial(oracle_c, monkeypatch):
    """Hermitian generators: the update sweep evaluates the Chebyshev-economized polynomial of the step (same truncation bound,
    fewer matvec rounds) through the Taylor kernels' Horner loop.  It must agree with the Taylor plan (KROTOV_HIP_ECON=0) far inside
    the parity tolerances, with the oracle at them, fall back to Taylor where the updated pulse leaves the 1 % margin the row of
    the step was built for (large updates), and stay off for generators with a non-Hermitian drift."""
    cases = [wl.dense_problem(64, 12, 256, 2, 171, 0.05, 5.0),     # C3's shape: VALU kernel, Hermitian-controls variant
             wl.dense_problem(64, 8, 512, 2, 172, 0.05, 5.0),      # matrix-core sweep, one tile
             wl.dense_problem(64, 6, 1024, 2, 173, 0.05, 5.0),     # matrix-core sweep, two tiles
             wl.dense_problem(16, 20, 32, 1, 174, 0.1, 1.0),       # C2's shape
             wl.dense_problem(33, 9, 5, 3, 175, 0.4, 2.0),         # larger ||H dt||: several degrees saved
             wl.dense_problem(20, 10, 7, 2, 176, 0.1, 2e-3)]       # huge updates: most steps leave the margin -> Taylor plan
    for ci, p in enumerate(cases):
        res = {}
        for mode in ["0", "1"]:
            monkeypatch.setenv("KROTOV_HIP_ECON", mode)
            res[mode], info = _iterate(p)
            assert info["economized_polynomial"] == int(mode)
        assert _rel(res["1"][0], res["0"][0]) < 1e-12, ci
        assert np.max(np.abs(res["1"][2] - res["0"][2])) < 2e-13 * max(1.0, np.max(np.abs(res["0"][2]))), ci
        monkeypatch.setenv("KROTOV_HIP_ECON", "1")
        _run_case(oracle_c, p, check_storage=False)
    # a looser tolerance: lower degrees on both plans, still consistent with each other at that tolerance
    p = cases[4]
    monkeypatch.setenv("KROTOV_HIP_ECON", "0")
    tight, _ = _iterate(p)
    out = {}
    for mode in ["0", "1"]:
        monkeypatch.setenv("KROTOV_HIP_ECON", mode)
        out[mode], _ = _iterate(p, prop_tol=1e-9)
    assert _rel(out["1"][0], out["0"][0]) < 1e-6 and _rel(out["1"][0], tight[0]) < 1e-6
    # non-Hermitian drift: the polynomial is not valid off the real axis -> Taylor
    monkeypatch.setenv("KROTOV_HIP_ECON", "1")
    p = wl.dense_problem(12, 6, 5, 2, 177, 0.1, 2.0)
    p["H0"] = p["H0"] - 0.05j * np.eye(12)
    (_, _, _), info = _iterate(p)
    assert info["economized_polynomial"] == 0
    _run_case(oracle_c, p)


def _oracle_iteration(oc, p):
    """Forward sweep + one iteration of the oracle.  oc = the C oracle (the reference's execution order: one expm per trajectory and
    step; all host threads, the sums keep the serial order) or None = the NumPy/SciPy twin (one expm per step, shared by the
    trajectories: the large shared-generator cases)."""
    if oc is None:
        import krotov_oracle as o
        psiT_o, _ = o.forward(p["H0"], p["Hc"], p["tlist"], p["eps"], p["psi0"], store=False)
        _, _, chiT = wl.chi_and_JT(p["functional"], psiT_o, p["tgt"])
        eps_o, ga_o, psiT1_o, X_o, _ = o.iteration(p["H0"], p["Hc"], p["tlist"], p["eps"], p["S"], p["lam"], p["psi0"], chiT)
    else:
        nt = oc.max_threads()
        psiT_o, _ = oc.forward(p["H0"], p["Hc"], p["tlist"], p["eps"], p["psi0"], store=False, nthreads=nt)
        _, _, chiT = wl.chi_and_JT(p["functional"], psiT_o, p["tgt"])
        eps_o, ga_o, psiT1_o, X_o, _ = oc.iteration(p["H0"], p["Hc"], p["tlist"], p["eps"], p["S"], p["lam"], p["psi0"], chiT, nthreads=nt)
    return dict(psiT=psiT_o, chiT=chiT, eps_new=eps_o, g_a_int=ga_o, psiT1=psiT1_o, X0=X_o[0].copy())


def _iterate_against_oracle(ref, p, label):
    """One forward sweep + one iteration on a fresh handle, EVERY stage against the oracle (`ref` = _oracle_iteration(oc, p)), so that a
    disagreement between two kernel variants names the wrong one and the sweep it comes from; on a mismatch the same handle repeats
    the backward sweep, which tells a transient fault from a persistent one.  Returns what _iterate returns."""
    K = p["psi0"].shape[0]
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K) as eng:
        eng.set_initial_states(p["psi0"])
        psiT = eng.forward(p["eps"])
        e_fw = np.max(np.abs(psiT - ref["psiT"]))
        out = eng.iteration(p["eps"], ref["chiT"], p["S"], p["lam"])
        X0 = np.array([eng.read_storage("bw", k, 0) for k in range(K)])
        scale = max(1.0, np.max(np.abs(ref["X0"])))
        e_bw = np.max(np.abs(X0 - ref["X0"])) / scale
        e_eps, e_psi = _rel(out[0], ref["eps_new"]), np.max(np.abs(out[2] - ref["psiT1"]))
        info, plan = eng.info(), eng.plan()
        if not (e_fw < TOL_STATE and e_bw < TOL_STATE and e_eps < TOL_PULSE and e_psi < 1e-9):
            eng.backward(p["eps"], ref["chiT"])   # the same handle once more: transient or persistent?
            X0b = np.array([eng.read_storage("bw", k, 0) for k in range(K)])
            import os
            env = {k: v for k, v in os.environ.items() if k.startswith("KROTOV_HIP")}
            pytest.fail("%s against the oracle: forward psi(T) %.2e, backward X[0] %.2e (same handle again: %.2e), pulses %.2e, psi(T)' %.2e; %s; "
                        "norm bounds %s; environment %s" % (label, e_fw, e_bw, np.max(np.abs(X0b - ref["X0"])) / scale, e_eps, e_psi, plan,
                                                          np.array2string(eng.norm_bounds(), precision=6), env))
    return out, info


def test_rowsplit_one_control_instantiation(oracle_c, monkeypatch):
    """Row-split kernels (64 < N <= 256) with L = 1 run an instantiation compiled for one control (fewer accumulators, four columns in
    flight in the streaming passes).  BOTH instantiations (KROTOV_HIP_RS_LMAX1 = 0 / 1) must meet the oracle at the parity tolerances
    at every stage -- checked first, so that a failure names the instantiation and the sweep -- and then agree with each other far
    inside them: Hermitian and non-Hermitian generators, N that does not fill the padded size, K that gives R = 4 and R = 8
    workgroups per trajectory."""
    cases = [wl.dense_problem(100, 40, 6, 1, 191, 0.02, 2.0), wl.dense_problem(256, 36, 24, 1, 192, 0.01, 2.0),
             wl.dense_problem(130, 35, 40, 1, 193, 0.02, 5.0)]
    cases[2]["H0"] = cases[2]["H0"] - 0.03j * np.diag(np.random.default_rng(5).uniform(0, 1, 130))
    for ci, p in enumerate(cases):
        res = {}
        ref = _oracle_iteration(oracle_c, p)
        for mode in ["0", "1"]:
            monkeypatch.setenv("KROTOV_HIP_RS_LMAX1", mode)
            res[mode], info = _iterate_against_oracle(ref, p, "case %d, KROTOV_HIP_RS_LMAX1=%s" % (ci, mode))
            assert info["rowsplit_R"] > 0 and info["LMAX"] == (1 if mode == "1" else 2), info
        assert _rel(res["1"][0], res["0"][0]) < 1e-12, ci
        assert np.max(np.abs(res["1"][2] - res["0"][2])) < 1e-12 * max(1.0, np.max(np.abs(res["0"][2]))), ci
