"""GPU: the generic (multi-pass) kernels of kr_device_generic.h -- the path behind the capacity limits of the persistent
kernels (N > 256, L > 4, more trajectories than co-resident workgroups) and the carrier of the per-step exchanges that run
outside a kernel (RCCL all-reduce, host callback) -- against the CPU oracle, same tolerances as test_gpu_parity.py."""
import os
import sys

import numpy as np
import pytest

import krotov_jl_amd as kr
from krotov_jl_amd import workloads as wl
from krotov_jl_amd import _capi as capi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL_STATE, TOL_PULSE = 1e-11, 1e-10


def _rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def _check(oc, p, check_storage=True, **kw):
    K, N = p["psi0"].shape
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K, **kw) as eng:
        info = eng.info()
        eng.set_initial_states(p["psi0"])
        psiT = eng.forward(p["eps"])
        psiT_o, Phi_o = oc.forward(p["H0"], p["Hc"], p["tlist"], p["eps"], p["psi0"])
        assert np.max(np.abs(psiT - psiT_o)) < TOL_STATE
        _, _, chiT = wl.chi_and_JT(p["functional"], psiT_o, p["tgt"])
        eps_new, ga, psiT1 = eng.iteration(p["eps"], chiT, p["S"], p["lam"])
        eps_o, ga_o, psiT1_o, X_o, Phi1_o = oc.iteration(p["H0"], p["Hc"], p["tlist"], p["eps"], p["S"], p["lam"], p["psi0"], chiT)
        assert _rel(eps_new, eps_o) < TOL_PULSE
        assert np.max(np.abs(ga - ga_o)) < 1e-9 * np.max(np.abs(ga_o)) + 1e-24   # (N = 1: the integrand is rounding noise)
        assert np.max(np.abs(psiT1 - psiT1_o)) < 1e-9
        if check_storage:
            assert np.max(np.abs(eng.storage("bw") - X_o)) < TOL_STATE * max(1.0, np.max(np.abs(X_o)))
            assert np.max(np.abs(eng.storage("fw") - Phi1_o)) < 1e-10
        again = eng.iteration(p["eps"], chiT, p["S"], p["lam"])
        assert np.array_equal(again[0], eps_new) and np.array_equal(again[2], psiT1)   # run-to-run bitwise
    return info


def test_generic_kernels_forced_on_small_shapes(oracle_c, monkeypatch):
    """KROTOV_HIP_GENERIC=1: every sweep on the generic kernels, shapes the persistent kernels also cover."""
    monkeypatch.setenv("KROTOV_HIP_GENERIC", "1")
    for (N, NT, K, L, seed) in [(1, 4, 2, 1, 301), (2, 6, 1, 1, 302), (7, 9, 5, 2, 303), (16, 12, 9, 3, 304), (33, 8, 6, 2, 305),
                                (64, 6, 10, 2, 306), (100, 4, 3, 1, 307)]:
        info = _check(oracle_c, wl.dense_problem(N, NT, K, L, seed, 0.1, 1.0))
        assert info["generic_known"] == 1 and info["generic_update"] == 1
    p = wl.dense_problem(12, 8, 4, 1, 308, 0.1, 20.0)    # ||H dt|| > 1: sub-steps
    p["H0"] = p["H0"] * 40.0
    _check(oracle_c, p)
    rng = np.random.default_rng(5)
    p = wl.dense_problem(10, 15, 4, 2, 309, 0.1, 1.0)    # non-Hermitian, non-uniform grid
    p["H0"] = p["H0"] - 0.05j * np.diag(rng.uniform(0, 1, 10)) + 0.02 * rng.normal(size=(10, 10))
    p["tlist"] = np.cumsum(np.r_[0.0, rng.uniform(0.05, 0.2, 15)])
    _check(oracle_c, p)


def test_beyond_the_capacity_of_the_persistent_kernels(oracle_c):
    """N > 256, L > 4 and more trajectories than 256 co-resident workgroups can carry: no KR_ERR_UNSUPPORTED any more
    (the reference has no such limits, /root/reference/src/workspace.jl:128-161) -- the generic kernels take over."""
    info = _check(oracle_c, wl.dense_problem(300, 3, 2, 5, 311, 0.02, 1.0), check_storage=False)   # (round 3: N <= 512 with up to four
    assert info["generic_known"] == 1 and info["generic_update"] == 1                              #  controls runs on the row-split kernels)
    info = _check(oracle_c, wl.dense_problem(9, 10, 5, 6, 312, 0.1, 1.0))           # six controls
    assert info["generic_update"] == 1
    info = _check(oracle_c, wl.dense_problem(64, 3, 1100, 2, 313, 0.05, 5.0), check_storage=False)   # K = 1100 at N = 64
    assert info["generic_update"] == 0 and info["traj_per_wg"] == 8      # round 3: two tile sets per workgroup, still persistent
    info = _check(oracle_c, wl.dense_problem(64, 3, 4200, 2, 314, 0.05, 5.0), check_storage=False)   # past 16 trajectories per CU
    assert info["generic_update"] == 1 and info["K_padded"] >= 4200
    info = _check(oracle_c, wl.dense_problem(8, 3, 8200, 1, 315, 0.05, 1.0), check_storage=False)   # > 2048 generic tiles (round 3: kr_create
    assert info["generic_update"] == 1 and info["K_padded"] >= 8200                               #  sized them against the persistent mailbox)


@pytest.mark.parametrize("N,K,L", [(130, 33, 1), (400, 5, 2), (129, 4, 3)])
def test_generic_kernels_with_more_than_48_kb_of_lds(oracle_c, monkeypatch, N, K, L):
    """Generic kernels whose dynamic LDS exceeds the 48 KB a kernel gets by default (N = 130 with four trajectories per tile: 49.7 KB;
    N = 400 with two: 59 KB): the limit is raised per kernel and device -- and must leave room for the kernel's STATIC LDS (the
    fused forward step of round 3 carries 2 KB of it: raising the dynamic limit to the full 160 KB failed with `invalid argument`,
    found by tools/plan_fuzz.py at the end of round 3)."""
    monkeypatch.setenv("KROTOV_HIP_GENERIC", "1")
    info = _check(oracle_c, wl.dense_problem(N, 3, K, L, 330 + N, 0.03, 1.0), check_storage=False)
    assert info["generic_known"] == 1 and info["generic_update"] == 1


def test_generic_with_generators_and_masks(oracle_c, monkeypatch):
    monkeypatch.setenv("KROTOV_HIP_GENERIC", "1")
    import krotov_oracle as o
    rng = np.random.default_rng(78)
    N, NT, L, G, K = 11, 12, 2, 3, 9
    tg = np.array([0, 2, 1, 0, 2, 2, 1, 0, 1], np.int32)
    base = wl.dense_problem(N, NT, K, L, 320, 0.1, 1.5)
    H0 = np.array([base["H0"] * (1.0 + 0.1 * g) + 0.05 * wl.random_hermitian(rng, N) for g in range(G)])
    Hc = np.array([[base["Hc"][l] * (1.0 - 0.07 * g) for l in range(L)] for g in range(G)])
    present = np.ones((G, L), np.uint8)
    present[2, 0] = 0
    tl, eps, S, lam, psi0, tgt = base["tlist"], base["eps"], base["S"], base["lam"], base["psi0"], base["tgt"]
    with kr.HipEngine(H0, Hc, tl, K, traj_generator=tg, hc_present=present) as eng:
        assert eng.info()["generic_update"] == 1
        eng.set_initial_states(psi0)
        psiT = eng.forward(eps)
        _, _, chiT = wl.chi_and_JT("ss", psiT, tgt)
        eps_new, ga, psiT1 = eng.iteration(eps, chiT, S, lam)
    X = [o.backward(H0[tg[k]], Hc[tg[k]], tl, eps, chiT[k:k + 1])[:, 0] for k in range(K)]
    psi, eps_o = psi0.copy(), eps.copy()
    for n in range(NT):
        du = np.zeros(L)
        for l in range(L):
            for k in range(K):
                if present[tg[k], l]:
                    du[l] += np.vdot(X[k][n], Hc[tg[k], l] @ psi[k]).imag
        eps_o[:, n] = eps[:, n] + S[:, n] / lam * du
        for k in range(K):
            psi[k] = o.step_propagator(H0[tg[k]], Hc[tg[k]], eps_o[:, n], tl[n + 1] - tl[n]) @ psi[k]
    assert _rel(eps_new, eps_o) < TOL_PULSE and np.max(np.abs(psiT1 - psi)) < 1e-10


@pytest.mark.parametrize("exchange", [capi.KR_EXCHANGE_RCCL, capi.KR_EXCHANGE_CALLBACK])
def test_stepped_update_sweep_one_rank(oracle_c, exchange):
    """exchange = RCCL / CALLBACK on one rank: the known-pulse sweeps stay on the persistent kernels, the update sweep runs
    step by step (partial -> reduce -> exchange -> pulse update -> forward step).  RCCL: a world-1 communicator really is
    created (librccl loaded with dlopen) and ncclAllReduce really runs once per step."""
    p = wl.dense_problem(20, 25, 12, 2, 330, 0.1, 1.0)
    calls = []
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], 12, exchange=exchange) as eng:
        info = eng.info()
        assert info["generic_known"] == 0 and info["generic_update"] == 1
        assert info["exchange"] == ("rccl" if exchange == capi.KR_EXCHANGE_RCCL else "callback")
        if exchange == capi.KR_EXCHANGE_RCCL:
            eng.rccl_init(kr.HipEngine.rccl_unique_id())
        else:
            eng.set_exchange_callback(lambda buf: calls.append(buf.copy()))
        eng.set_initial_states(p["psi0"])
        psiT = eng.forward(p["eps"])
        _, _, chiT = wl.chi_and_JT("ss", psiT, p["tgt"])
        eps_new, ga, psiT1 = eng.iteration(p["eps"], chiT, p["S"], p["lam"])
    eps_o, ga_o, psiT1_o, _, _ = oracle_c.iteration(p["H0"], p["Hc"], p["tlist"], p["eps"], p["S"], p["lam"], p["psi0"], chiT)
    assert _rel(eps_new, eps_o) < TOL_PULSE and _rel(ga, ga_o) < 1e-9 and np.max(np.abs(psiT1 - psiT1_o)) < 1e-10
    if exchange == capi.KR_EXCHANGE_CALLBACK:
        assert len(calls) == 25 and calls[0].shape == (2,)


def _cb_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import krotov_jl_amd as kr2
    from krotov_jl_amd import workloads as wl2
    from krotov_jl_amd.distributed import shard_range
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        p = wl2.dense_problem(24, 30, 21, 2, 340, 0.1, 1.0)
        K = 21
        a, b = shard_range(K, rank, world)
        eng = kr2.HipEngine(p["H0"], p["Hc"], p["tlist"], b - a, device=0, rank=rank, world=world, K_global=K,
                            exchange=capi.KR_EXCHANGE_CALLBACK)

        def allreduce(buf):
            t = torch.from_numpy(buf)       # shares memory: summed in place
            dist.all_reduce(t)
        eng.set_exchange_callback(allreduce)
        eng.set_initial_states(p["psi0"][a:b])
        psiT = eng.forward(p["eps"])
        _, _, chiT = wl2.chi_and_JT("ss", psiT, p["tgt"][a:b], K_global=K)
        eps, ga, psiT = eng.iteration(p["eps"], chiT, p["S"], p["lam"])
        q.put((rank, eps, ga, psiT))
        dist.barrier()
        eng.close()
        dist.destroy_process_group()
    except Exception as exc:  # noqa: BLE001
        q.put((rank, "ERROR: %r" % (exc,), None, None))
        raise


def test_callback_exchange_two_ranks_vs_unsharded_oracle(oracle_c):
    """Two ranks sharing the GPU, the per-step sum crossing them through a HOST callback (a gloo all-reduce here; an MPI
    all-reduce from Julia in production): equal to the unsharded oracle, identical pulses on both ranks."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_cb_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = {}
    for _ in range(2):
        rank, eps, ga, psiT = q.get(timeout=240)
        assert not isinstance(eps, str), eps
        got[rank] = (eps, ga, psiT)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1])
    p = wl.dense_problem(24, 30, 21, 2, 340, 0.1, 1.0)
    psiT_o, _ = oracle_c.forward(p["H0"], p["Hc"], p["tlist"], p["eps"], p["psi0"], store=False)
    _, _, chiT = wl.chi_and_JT("ss", psiT_o, p["tgt"])
    eps_o, ga_o, psiT1_o, _, _ = oracle_c.iteration(p["H0"], p["Hc"], p["tlist"], p["eps"], p["S"], p["lam"], p["psi0"], chiT)
    assert _rel(got[0][0], eps_o) < TOL_PULSE and _rel(got[0][1], ga_o) < 1e-9
    assert np.max(np.abs(np.concatenate([got[0][2], got[1][2]]) - psiT1_o)) < 1e-9


@pytest.mark.parametrize("method", [capi.KR_PROP_CHEBY, capi.KR_PROP_NEWTON])
def test_propagator_methods_vs_oracle(oracle_c, method):
    """kr_config.prop_method (the reference's prop_method, /root/reference/src/optimize.jl:108,137-153): Chebyshev
    (Hermitian generators; spectral range from the control range, widened 5x in the update sweep as transform_control_ranges
    does, optimize.jl:238-244) and Newton interpolation at Leja points (any generator), against the oracle's exact expm at
    the tolerances of the Taylor path."""
    cases = [wl.dense_problem(12, 20, 5, 2, 601, 0.1, 5.0), wl.dense_problem(40, 10, 7, 1, 602, 0.05, 5.0),
             wl.dense_problem(6, 15, 3, 3, 603, 0.1, 5.0)]
    big = wl.dense_problem(10, 8, 4, 1, 604, 0.1, 50.0)
    big["H0"] = big["H0"] * 30.0          # ||H dt|| ~ 3: sub-steps for Newton, a long Chebyshev series
    cases.append(big)
    rng = np.random.default_rng(9)
    nonuni = wl.dense_problem(9, 12, 4, 2, 605, 0.1, 5.0)
    nonuni["tlist"] = np.cumsum(np.r_[0.0, rng.uniform(0.05, 0.2, 12)])   # one table row per distinct dt
    cases.append(nonuni)
    for p in cases:
        info = _check(oracle_c, p, prop_method=method, options={"CHEBY_RECURRENCE": 1})
        assert info["generic_known"] == 1 and info["generic_update"] == 1
    if method == capi.KR_PROP_CHEBY:
        # round 5: without the option a Chebyshev request with a persistent plan stays on the persistent kernels (the update sweep's
        # economized polynomial IS the Chebyshev truncation of the step, in powers of the generator) -- same tolerances
        for p in cases:
            K = p["psi0"].shape[0]
            info = _check(oracle_c, p, prop_method=method)
            assert info["generic_known"] == 0 and info["generic_update"] == 0
            with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K, prop_method=method) as eng:
                assert "prop_method = CHEBY is served by the persistent kernels" in eng.plan(), eng.plan()
        # ... and without a persistent plan (L > 4 here) the recurrence on the generic kernels, as before
        p = wl.dense_problem(9, 8, 3, 5, 608, 0.1, 5.0)
        info = _check(oracle_c, p, prop_method=method)
        assert info["generic_known"] == 1 and info["generic_update"] == 1
    if method == capi.KR_PROP_NEWTON:     # non-Hermitian generator (a Liouvillian): Newton only
        p = wl.liouvillian_problem(n_levels=4, NT=12, K=5, seed=606, dt=0.1, lam=5.0)
        _check(oracle_c, p, prop_method=method)
    else:
        p = wl.dense_problem(8, 5, 2, 1, 607, 0.1, 1.0)
        p["H0"] = p["H0"] + 0.01j * np.eye(8)
        with pytest.raises(kr.KrotovHipError) as ei:
            kr.HipEngine(p["H0"], p["Hc"], p["tlist"], 2, prop_method=method)
        assert "Hermitian" in str(ei.value)


def test_polynomial_propagator_reports_a_control_that_leaves_its_range(oracle_c):
    """Chebyshev / Newton are initialised for 5 x the guess amplitude; an update beyond that must be reported, not
    propagated with a polynomial that no longer covers the spectrum."""
    p = wl.dense_problem(8, 10, 3, 1, 610, 0.1, 1e-4)      # tiny lambda_a: a huge update
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], 3, prop_method=capi.KR_PROP_CHEBY, options={"CHEBY_RECURRENCE": 1}) as eng:
        eng.set_initial_states(p["psi0"])
        psiT = eng.forward(p["eps"])
        _, _, chiT = wl.chi_and_JT("ss", psiT, p["tgt"])
        with pytest.raises(kr.KrotovHipError) as ei:
            eng.iteration(p["eps"], chiT, p["S"], p["lam"])
        assert ei.value.code == capi.KR_ERR_UNSUPPORTED and "left the range" in str(ei.value)


@pytest.mark.parametrize("shape,deg,persistent", [((10, 25, 6, 2), 3, 1), ((64, 12, 256, 2), 3, 1), ((64, 8, 600, 2), 2, 1), ((33, 9, 40, 3), 3, 1),
                                                   ((10, 25, 6, 2), 5, 0), ((100, 6, 5, 1), 3, 1), ((100, 5, 200, 2), 2, 1), ((130, 5, 40, 1), 3, 1),
                                                   ((300, 4, 12, 2), 3, 1), ((130, 4, 6, 1), 4, 0), ((130, 4, 100, 1), 3, 1), ((100, 6, 5, 2), 3, 1), ((100, 5, 100, 1), 3, 1)],
                         ids=["np16", "c3_shape_vector", "matrix_cores", "lmax4", "degree5_generic_update", "rowsplit_np128_few", "rowsplit_np128_r1",
                              "rowsplit_pair_np256", "rowsplit_turns_np512", "rowsplit_degree4_generic_update", "rowsplit_pair_sets_np256", "rowsplit_single_np128", "rowsplit_pair_np128_k100"])
def test_nonlinear_polynomial_controls(shape, deg, persistent, request):
    """Non-linear controls, static part: H(eps) = H0 + sum_l f_l(eps_l) H_l with polynomial f_l; the
    control derivative mu_l = f_l'(eps_l) H_l is evaluated at the GUESS value of the step, as _eval_mu does
    (/root/reference/src/optimize.jl:268-272,337,345).  Against a serial restatement of optimize.jl:328-370 built on the
    oracle's exact step propagator.  Round 5: degree <= 3 runs on the PERSISTENT update kernels (kr_sweep_update /
    kr_mf_update<.., POLY> at N <= 64, the row-split single / turns kernels above with a run-time branch, the pair mode with
    instantiations of its own; generic_update == 0), the known-pulse sweeps of every persistent family take the amplitudes
    in the place of the pulses; higher degrees keep the stepped generic update."""
    import krotov_oracle as o
    N, NT, K, L = shape
    p = wl.dense_problem(N, NT, K, L, 620 + N, 0.1 if N <= 16 else 0.05, 0.5)
    H0, Hc, tl, eps, S, lam, psi0, tgt = (p[k] for k in ("H0", "Hc", "tlist", "eps", "S", "lam", "psi0", "tgt"))
    rng = np.random.default_rng(621)
    poly = np.zeros((L, deg + 1))
    poly[0, :3] = [0.0, 1.0, 0.5]                       # f_0 = x + x^2 / 2
    for l in range(1, L):
        poly[l, :2] = [0.1, 0.8]
        poly[l, deg] = 2.0 / l                          # f_l = 0.1 + 0.8 x + (2 / l) x^deg
    f = lambda l, x: sum(poly[l, q] * x ** q for q in range(deg + 1))
    fp = lambda l, x: sum(q * poly[l, q] * x ** (q - 1) for q in range(1, deg + 1))
    amp = lambda e: np.array([f(l, e[l]) for l in range(L)])
    with kr.HipEngine(H0, Hc, tl, K, control_poly=poly) as eng:
        info = eng.info()
        assert info["generic_update"] == (0 if persistent else 1), eng.plan()
        assert info["generic_known"] == 0, eng.plan()
        if N > 64 and persistent:
            assert info["rowsplit_R"] >= 1, eng.plan()
            assert ("kr_rowsplit_pair_sweep" in eng.plan()) == ("pair" in request.node.callspec.id), eng.plan()
        eng.set_initial_states(psi0)
        psiT = eng.forward(eps)
        Us = [o.step_propagator(H0, Hc, amp(eps[:, n]), tl[n + 1] - tl[n]) for n in range(NT)]
        psi = psi0.copy()
        for n in range(NT):
            psi = psi @ Us[n].T
        assert np.max(np.abs(psiT - psi)) < 1e-11
        _, _, chiT = wl.chi_and_JT("ss", psi, tgt)
        eps_new, ga, psiT1 = eng.iteration(eps, chiT, S, lam)
        again = eng.iteration(eps, chiT, S, lam)
        assert np.array_equal(again[0], eps_new)
    X = np.empty((NT + 1, K, N), complex)
    X[NT] = chiT
    for n in range(NT - 1, -1, -1):
        X[n] = X[n + 1] @ Us[n].conj()      # chi <- U^dagger chi
    psi, eps_o, ga_o = psi0.copy(), eps.copy(), np.zeros(L)
    for n in range(NT):
        for l in range(L):
            du = fp(l, eps[l, n]) * np.sum(np.einsum("ki,ki->k", X[n].conj(), psi @ Hc[l].T).imag)
            eps_o[l, n] = eps[l, n] + S[l, n] / lam[l] * du
            ga_o[l] += S[l, n] / lam[l] * du * du * (tl[n + 1] - tl[n])
        psi = psi @ o.step_propagator(H0, Hc, amp(eps_o[:, n]), tl[n + 1] - tl[n]).T
    assert _rel(eps_new, eps_o) < TOL_PULSE and _rel(ga, ga_o) < 1e-9 and np.max(np.abs(psiT1 - psi)) < 1e-10
    assert np.max(np.abs(eps_new - eps)) > 1e-4


@pytest.mark.parametrize("shape", [(20, 14, 30, 2), (64, 10, 300, 2), (100, 5, 9, 1)], ids=["np32", "np64_two_generators_300", "rowsplit_np128"])
def test_nonlinear_controls_with_per_trajectory_generators(shape):
    """Non-linear controls together with an ensemble of distinct generators (both at once, /root/reference/src/
    optimize.jl:268-272,337,343-347 -- mu_kl = f_l'(eps_l) H_l^(g(k)) differs per trajectory): the persistent update kernels, against the
    serial restatement built on the oracle's exact step propagator, generator by generator."""
    import krotov_oracle as o
    N, NT, K, L = shape
    p = wl.dense_problem(N, NT, K, L, 640 + N, 0.05, 0.5)
    tl, eps, S, lam, psi0, tgt = (p[k] for k in ("tlist", "eps", "S", "lam", "psi0", "tgt"))
    rng = np.random.default_rng(641)
    H0 = np.stack([p["H0"], 0.7 * p["H0"] + 0.1 * np.diag(rng.normal(size=N))])
    Hc = np.stack([p["Hc"], 1.3 * p["Hc"]])
    tg = rng.integers(0, 2, K).astype(np.int32)
    tg[:2] = [0, 1]
    poly = np.array([[0.0, 1.0, 0.5, 0.3], [0.05, 0.9, 0.0, 1.0]][:L])
    f = lambda l, x: sum(poly[l, q] * x ** q for q in range(4))
    fp = lambda l, x: sum(q * poly[l, q] * x ** (q - 1) for q in range(1, 4))
    amp = lambda e: np.array([f(l, e[l]) for l in range(L)])
    with kr.HipEngine(H0, Hc, tl, K, control_poly=poly, traj_generator=tg) as eng:
        info = eng.info()
        assert info["generic_update"] == 0 and info["generic_known"] == 0, eng.plan()
        eng.set_initial_states(psi0)
        psiT = eng.forward(eps)
        Us = [[o.step_propagator(H0[g], Hc[g], amp(eps[:, n]), tl[n + 1] - tl[n]) for n in range(NT)] for g in range(2)]
        psi = psi0.copy()
        for n in range(NT):
            for g in range(2):
                psi[tg == g] = psi[tg == g] @ Us[g][n].T
        assert np.max(np.abs(psiT - psi)) < 1e-11
        _, _, chiT = wl.chi_and_JT("ss", psi, tgt)
        eps_new, ga, psiT1 = eng.iteration(eps, chiT, S, lam)
    X = np.empty((NT + 1, K, N), complex)
    X[NT] = chiT
    for n in range(NT - 1, -1, -1):
        for g in range(2):
            X[n][tg == g] = X[n + 1][tg == g] @ Us[g][n].conj()
    psi, eps_o, ga_o = psi0.copy(), eps.copy(), np.zeros(L)
    for n in range(NT):
        for l in range(L):
            du = 0.0
            for g in range(2):
                m = tg == g
                du += np.sum(np.einsum("ki,ki->k", X[n][m].conj(), psi[m] @ Hc[g][l].T).imag)
            du *= fp(l, eps[l, n])
            eps_o[l, n] = eps[l, n] + S[l, n] / lam[l] * du
            ga_o[l] += S[l, n] / lam[l] * du * du * (tl[n + 1] - tl[n])
        for g in range(2):
            m = tg == g
            psi[m] = psi[m] @ o.step_propagator(H0[g], Hc[g], amp(eps_o[:, n]), tl[n + 1] - tl[n]).T
    assert _rel(eps_new, eps_o) < TOL_PULSE and _rel(ga, ga_o) < 1e-9 and np.max(np.abs(psiT1 - psi)) < 1e-10
    assert np.max(np.abs(eps_new - eps)) > 1e-4
