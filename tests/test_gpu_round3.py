"""GPU tests added in round 3 (review of round 2):

  * the device-resident functional chain (kr_iteration_functional_dev: tau, chi(T) on the device) pinned to the ORACLE --
    three chained iterations of oracle/krotov_oracle.py's taus / chi_* / iteration, not the package's own host chain
    (/root/reference/src/optimize.jl:297-302,381-386) -- on one handle and on the two-shard device list;
  * BASELINE config 4 at FULL size on one GPU (N=64, N_T=1000, K=1024): size-independent properties;
  * J_T_sm across two ranks when the carrier is the host callback (the tau sum travels through it as well);
  * the boundary hardening: configurable spin bound, a device list of one entry, kr_plan_description.
"""
import os
import sys

import numpy as np
import pytest

import krotov_jl_amd as kr
from krotov_jl_amd import _capi as capi
from krotov_jl_amd import workloads as wl

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TOL_PULSE, TOL_STATE, TOL_GA = 1e-10, 1e-10, 1e-9


def _rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def _oracle_chain(p, fname, iters):
    """optimize_krotov's loop (optimize.jl:209-216) restated with the oracle only: tau -> chi(T) -> krotov_iteration."""
    import krotov_oracle as o
    chi_of = {"sm": o.chi_sm, "ss": o.chi_ss, "re": o.chi_re}[fname]
    J_of = {"sm": o.J_T_sm, "ss": o.J_T_ss, "re": o.J_T_re}[fname]
    eps = p["eps"].copy()
    psiT, _ = o.forward(p["H0"], p["Hc"], p["tlist"], eps, p["psi0"], store=False)
    J = []
    ga = None
    for _ in range(iters):
        tau = o.taus(psiT, p["tgt"])
        J.append(J_of(tau))
        eps, ga, psiT, _, _ = o.iteration(p["H0"], p["Hc"], p["tlist"], eps, p["S"], p["lam"], p["psi0"], chi_of(tau, p["tgt"]))
    tau = o.taus(psiT, p["tgt"])
    J.append(J_of(tau))
    return eps, ga, psiT, tau, J


@pytest.mark.parametrize("fname,fid", [("sm", 0), ("ss", 1), ("re", 2)])
@pytest.mark.parametrize("devices", [None, [0, 0]])
def test_device_functional_chain_vs_oracle_chain(fname, fid, devices):
    """Three chained iterations with tau / chi(T) formed ON THE DEVICE against three chained ORACLE iterations."""
    import torch
    p = wl.dense_problem(12, 30, 10, 2, 731 + fid, 0.1, 0.5)
    K = 10
    eps_o, ga_o, psiT_o, tau_o, J_o = _oracle_chain(p, fname, 3)
    assert J_o[-1] < J_o[0] and np.max(np.abs(eps_o - p["eps"])) > 1e-4      # the chain really moves the pulse
    kw = {} if devices is None else {"devices": devices}
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K, **kw) as eng:
        eng.set_initial_states(p["psi0"])
        eng.set_target_states(p["tgt"])
        eng.set_update_params(p["S"], p["lam"])
        e0 = torch.from_numpy(p["eps"].copy()).cuda()
        e1 = torch.empty_like(e0)
        torch.cuda.synchronize()
        eng.forward_dev(e0.data_ptr())
        for _ in range(3):
            eng.iteration_functional_dev(fid, e0.data_ptr(), e1.data_ptr())
            e0, e1 = e1, e0
        eng.synchronize()
        assert _rel(e0.cpu().numpy(), eps_o) < TOL_PULSE
        assert np.max(np.abs(eng.final_states() - psiT_o)) < TOL_STATE
        assert _rel(eng.g_a_int(), ga_o) < TOL_GA
        assert np.max(np.abs(eng.taus() - tau_o)) < 1e-11


def test_c4_full_size_one_gpu_properties():
    """BASELINE config 4 as stated, on ONE GPU: N=64, N_T=1000, K=1024, L=2 (two matrix-core tiles per workgroup).  The oracle
    would need hours, so the size-independent properties: every trajectory keeps its norm; <chi_k(t_n)|psi_k(t_n)> under the
    GUESS pulse is the same at every n (chi is propagated by the adjoint of what propagates psi:
    /root/reference/src/workspace.jl:69,149-161) -- sampled from the two storages; a second engine reproduces pulses and
    states bit for bit; J_T falls monotonically over three iterations."""
    p = wl.config("C4")
    K, N = p["psi0"].shape
    NT = p["eps"].shape[1]
    assert (K, N, NT) == (1024, 64, 1000)
    lam = np.full(2, 0.2)
    outs = []
    for run in range(2):
        with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K) as eng:
            info = eng.info()
            assert info["matrix_core_sweep"] == 1 and info["workgroups"] == 256 and info["generic_update"] == 0, eng.plan()
            eng.set_initial_states(p["psi0"])
            psiT = eng.forward(p["eps"])
            assert np.max(np.abs(np.linalg.norm(psiT, axis=1) - 1.0)) < 1e-11
            tau, J0, chiT = wl.chi_and_JT("ss", psiT, p["tgt"])
            if run == 0:
                # forward storage (guess pulse) against backward storage (guess pulse): invariant overlaps
                eng.backward(p["eps"], chiT)
                for k in (0, 511, 1023):
                    ref = np.vdot(chiT[k], psiT[k])
                    for n in (0, 1, 250, 777, 1000):
                        ov = np.vdot(eng.read_storage("bw", k, n), eng.read_storage("fw", k, n))
                        assert abs(ov - ref) < 1e-10 * np.linalg.norm(chiT[k]), (k, n, ov, ref)   # |psi| = 1
            eps = p["eps"].copy()
            J = [J0]
            for _ in range(3):
                _, _, chiT = wl.chi_and_JT("ss", psiT, p["tgt"])
                eps, ga, psiT = eng.iteration(eps, chiT, p["S"], lam)
                J.append(wl.chi_and_JT("ss", psiT, p["tgt"])[1])
            assert np.max(np.abs(np.linalg.norm(psiT, axis=1) - 1.0)) < 1e-11
            assert all(b < a for a, b in zip(J, J[1:])), J
            assert np.all(ga > 0.0) and np.max(np.abs(eps - p["eps"])) > 1e-4
            # the forward storage of the update sweep ends in the final states, bit for bit
            for k in (0, 1023):
                assert np.array_equal(eng.read_storage("fw", k, NT), psiT[k])
            outs.append((eps.copy(), psiT.copy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


# ---- J_T_sm across ranks through the host callback -------------------------------------------------------------------
def _sm_cb_worker(rank, world, port, q):
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
        p = wl2.dense_problem(12, 30, 10, 2, 731, 0.1, 0.5)
        K = 10
        a, b = shard_range(K, rank, world)
        eng = kr2.HipEngine(p["H0"], p["Hc"], p["tlist"], b - a, device=0, rank=rank, world=world, K_global=K,
                            exchange=capi.KR_EXCHANGE_CALLBACK)
        counts = []

        def allreduce(buf):
            counts.append(len(buf))
            t = torch.from_numpy(buf)       # shares memory: summed in place
            dist.all_reduce(t)
        eng.set_exchange_callback(allreduce)
        eng.set_initial_states(p["psi0"][a:b])
        eng.set_target_states(p["tgt"][a:b])
        eng.set_update_params(p["S"], p["lam"])
        e0 = torch.from_numpy(p["eps"].copy()).cuda()
        e1 = torch.empty_like(e0)
        torch.cuda.synchronize()
        eng.forward_dev(e0.data_ptr())
        for _ in range(3):
            eng.iteration_functional_dev(0, e0.data_ptr(), e1.data_ptr())    # J_T_sm
            e0, e1 = e1, e0
        eng.synchronize()
        q.put((rank, e0.cpu().numpy(), eng.g_a_int(), eng.final_states(), counts))
        dist.barrier()
        eng.close()
        dist.destroy_process_group()
    except Exception as exc:  # noqa: BLE001
        q.put((rank, "ERROR: %r" % (exc,), None, None, None))
        raise


def test_device_chain_j_t_sm_two_ranks_callback_carrier():
    """Round-2 advisor finding: with exchange = CALLBACK (or RCCL) and world > 1 the J_T_sm chain could not run -- it wanted
    peer mailboxes those handles never have.  The sum of the taus now travels through the handle's own carrier (one callback
    of 2 doubles per iteration, next to the N_T callbacks of L doubles).  Against the unsharded ORACLE chain."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29950 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_sm_cb_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = {}
    for _ in range(2):
        rank, eps, ga, psiT, counts = q.get(timeout=240)
        assert not isinstance(eps, str), eps
        got[rank] = (eps, ga, psiT, counts)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert np.array_equal(got[0][0], got[1][0])
    assert got[0][3].count(2) >= 3 and len(got[0][3]) == 3 * (30 + 1)      # per iteration: one tau sum (2 doubles; L = 2 as well) + N_T steps
    p = wl.dense_problem(12, 30, 10, 2, 731, 0.1, 0.5)
    eps_o, ga_o, psiT_o, _, _ = _oracle_chain(p, "sm", 3)
    assert _rel(got[0][0], eps_o) < TOL_PULSE and _rel(got[0][1], ga_o) < TOL_GA
    assert np.max(np.abs(np.concatenate([got[0][2], got[1][2]]) - psiT_o)) < TOL_STATE


# ---- boundary hardening --------------------------------------------------------------------------------------------
def test_device_list_of_one_names_the_device():
    p = wl.dense_problem(8, 5, 6, 1, 520, 0.1, 1.0)
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], 6, devices=[0]) as eng:
        assert "on device 0" in eng.plan()
        eng.set_initial_states(p["psi0"])
        assert np.max(np.abs(np.linalg.norm(eng.forward(p["eps"]), axis=1) - 1.0)) < 1e-12
    with pytest.raises(kr.KrotovHipError) as ei:     # (round 2: silently ran on the current device)
        kr.HipEngine(p["H0"], p["Hc"], p["tlist"], 6, devices=[99])
    assert ei.value.code == capi.KR_ERR_INVALID and "out of range" in str(ei.value)


def test_plan_description_names_kernels_and_reasons():
    p = wl.dense_problem(16, 6, 40, 1, 530, 0.1, 1.0)
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], 40) as eng:
        assert "kr_sweep_update" in eng.plan() and "exchange = mailbox" in eng.plan()
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], 40, max_workgroups=2) as eng:      # 40 trajectories on 2 workgroups
        assert "generic" in eng.plan() and "co-resident" in eng.plan()
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], 40, spin_timeout_ms=250) as eng:
        assert "spin bound 250 ms" in eng.plan()
    with pytest.raises(kr.KrotovHipError):
        kr.HipEngine(p["H0"], p["Hc"], p["tlist"], 40, spin_timeout_ms=-1)


def _spin_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import time
    import torch.distributed as dist
    import krotov_jl_amd as kr2
    from krotov_jl_amd import workloads as wl2
    from krotov_jl_amd.distributed import connect_peers
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        p = wl2.dense_problem(8, 6, 8, 1, 540, 0.1, 1.0)
        eng = kr2.HipEngine(p["H0"], p["Hc"], p["tlist"], 4, device=0, rank=rank, world=world, K_global=8, max_workgroups=64,
                            spin_timeout_ms=300)
        connect_peers(eng, dist)
        eng.set_initial_states(p["psi0"][4 * rank:4 * rank + 4])
        psiT = eng.forward(p["eps"])
        _, _, chiT = wl2.chi_and_JT("ss", psiT, p["tgt"][4 * rank:4 * rank + 4], K_global=8)
        res = None
        if rank == 0:      # rank 1 never enters the update sweep: rank 0's kernels must give up after ~0.3 s, not hang, not 4 s
            t0 = time.perf_counter()
            try:
                eng.iteration(p["eps"], chiT, p["S"], p["lam"])
                res = ("no error", time.perf_counter() - t0)
            except kr2.KrotovHipError as exc:
                res = (exc.code, time.perf_counter() - t0, str(exc))
        dist.barrier()
        q.put((rank, res))
        eng.close()
        dist.destroy_process_group()
    except Exception as exc:  # noqa: BLE001
        q.put((rank, "ERROR: %r" % (exc,)))
        raise


def test_spin_bound_is_configurable():
    """kr_config.spin_timeout_ms (round 2: a fixed 4 s): a rank whose peer never arrives gives up after the configured bound
    with KR_ERR_TIMEOUT and a message that names the exchange."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29850 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_spin_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = dict(q.get(timeout=240) for _ in range(2))
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert not isinstance(got[0], str) and not isinstance(got[1], str), got
    code, secs = got[0][0], got[0][1]
    assert code == capi.KR_ERR_TIMEOUT, got[0]
    assert 0.25 < secs < 2.5, got[0]
    assert "exchange" in got[0][2]


# ---- several tile sets per workgroup with per-trajectory generators and an absent control derivative ------------------
def test_tile_sets_with_per_trajectory_generators_and_masks():
    """K = 1500 at N = 64 with G = 3 different generators (control_derivs[k][l], /root/reference/src/workspace.jl:74) of which one
    lacks control 1 (mu === nothing, optimize.jl:344): the workgroups carry two tile sets each and never mix generators.  Against a
    serial restatement of optimize.jl:328-370 built on the oracle's exact step propagator."""
    import krotov_oracle as o
    rng = np.random.default_rng(91)
    N, NT, L, G = 64, 6, 2, 3
    counts = [700, 500, 300]
    K = sum(counts)
    tg = rng.permutation(np.repeat(np.arange(G, dtype=np.int32), counts)).astype(np.int32)
    base = wl.dense_problem(N, NT, K, L, 920, 0.05, 0.05)
    H0 = np.array([base["H0"] * (1.0 + 0.1 * g) + 0.05 * wl.random_hermitian(rng, N) for g in range(G)])
    Hc = np.array([[base["Hc"][l] * (1.0 - 0.07 * g) for l in range(L)] for g in range(G)])
    present = np.ones((G, L), np.uint8)
    present[1, 1] = 0
    tl, eps, S, lam, psi0, tgt = base["tlist"], base["eps"], base["S"], base["lam"], base["psi0"], base["tgt"]
    with kr.HipEngine(H0, Hc, tl, K, traj_generator=tg, hc_present=present) as eng:
        info = eng.info()
        assert info["matrix_core_sweep"] == 1 and info["traj_per_wg"] == 8 and info["generic_update"] == 0, eng.plan()
        eng.set_initial_states(psi0)
        psiT = eng.forward(eps)
        U = [[o.step_propagator(H0[g], Hc[g], eps[:, n], tl[n + 1] - tl[n]) for n in range(NT)] for g in range(G)]
        psi = psi0.copy()
        for n in range(NT):
            for g in range(G):
                ks = tg == g
                psi[ks] = psi[ks] @ U[g][n].T
        assert np.max(np.abs(psiT - psi)) < 1e-11
        _, _, chiT = wl.chi_and_JT("ss", psi, tgt)
        eps_new, ga, psiT1 = eng.iteration(eps, chiT, S, lam)
        X_probe = eng.read_storage("bw", 777, 2)
    X = np.empty((NT + 1, K, N), complex)
    X[NT] = chiT
    for n in range(NT - 1, -1, -1):
        for g in range(G):
            ks = tg == g
            X[n, ks] = X[n + 1, ks] @ U[g][n].conj()       # chi <- U^dagger chi
    assert np.max(np.abs(X_probe - X[2, 777])) < 1e-11
    psi, eps_o, ga_o = psi0.copy(), eps.copy(), np.zeros(L)
    for n in range(NT):
        dt = tl[n + 1] - tl[n]
        for l in range(L):
            du = 0.0
            for g in range(G):
                if present[g, l]:
                    ks = tg == g
                    du += np.sum(np.conj(X[n, ks]) * (psi[ks] @ Hc[g, l].T)).imag
            eps_o[l, n] = eps[l, n] + S[l, n] / lam[l] * du
            ga_o[l] += S[l, n] / lam[l] * du * du * dt
        for g in range(G):
            ks = tg == g
            psi[ks] = psi[ks] @ o.step_propagator(H0[g], Hc[g], eps_o[:, n], dt).T
    assert np.max(np.abs(eps_o - eps)) > 1e-5
    assert _rel(eps_new, eps_o) < TOL_PULSE and _rel(ga, ga_o) < TOL_GA and np.max(np.abs(psiT1 - psi)) < TOL_STATE


def test_propagator_build_paterson_stockmeyer_agrees_with_horner_and_oracle(monkeypatch):
    """The matrix-core propagator build (kr_expm_mfma64) evaluates the step polynomial of degree 5..7 in the Paterson-Stockmeyer
    form (four products) instead of Horner (m - 1): same coefficients, so the known-pulse sweeps
    (/root/reference/src/optimize.jl:247-265, 303-317) must agree with the Horner build to rounding and with the oracle's exact step
    propagator to the usual tolerance -- Hermitian generator (economized rows, m = 7) and with the rows switched off (Taylor)."""
    import krotov_oracle as o
    for econ in ("1", "0"):
        monkeypatch.setenv("KROTOV_HIP_ECON", econ)
        p = wl.dense_problem(64, 24, 6, 2, 4711, 0.05, 0.04)
        out = {}
        for ps in ("1", "0"):
            monkeypatch.setenv("KROTOV_HIP_EXPM_PS", ps)
            with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], 6) as eng:
                assert eng.info()["precomputed_propagators"] == 1
                eng.set_initial_states(p["psi0"])
                psiT = eng.forward(p["eps"])
                _, _, chiT = wl.chi_and_JT("ss", psiT, p["tgt"])
                eps_new, ga, psiT2 = eng.iteration(p["eps"], chiT, p["S"], p["lam"])
                out[ps] = (psiT.copy(), eps_new.copy(), psiT2.copy())
        for a, b in zip(out["1"], out["0"]):
            assert np.max(np.abs(a - b)) < 5e-14
        psi_ref, _ = o.forward(p["H0"], p["Hc"], p["tlist"], p["eps"], p["psi0"], store=False)
        assert np.max(np.abs(out["1"][0] - psi_ref)) < 1e-11


def _serial_iteration(o, H0, Hc, present, tg, tl, eps, S, lam, psi0, tgt):
    """optimize.jl:303-370 restated serially on the oracle's exact step propagator, with per-trajectory generators (tg) and
    control_derivs === nothing where present[g, l] == 0 (optimize.jl:344)."""
    G, L, NT = len(H0), eps.shape[0], eps.shape[1]
    K, N = psi0.shape
    U = [[o.step_propagator(H0[g], Hc[g], eps[:, n], tl[n + 1] - tl[n]) for n in range(NT)] for g in range(G)]
    psi = psi0.copy()
    for n in range(NT):
        for g in range(G):
            ks = tg == g
            psi[ks] = psi[ks] @ U[g][n].T
    psiT = psi.copy()
    _, _, chiT = wl.chi_and_JT("ss", psiT, tgt)
    X = np.empty((NT + 1, K, N), complex)
    X[NT] = chiT
    for n in range(NT - 1, -1, -1):
        for g in range(G):
            ks = tg == g
            X[n, ks] = X[n + 1, ks] @ U[g][n].conj()
    psi, eps_o, ga_o = psi0.copy(), eps.copy(), np.zeros(L)
    for n in range(NT):
        dt = tl[n + 1] - tl[n]
        for l in range(L):
            du = 0.0
            for g in range(G):
                if present[g, l]:
                    ks = tg == g
                    du += np.sum(np.conj(X[n, ks]) * (psi[ks] @ Hc[g, l].T)).imag
            eps_o[l, n] = eps[l, n] + S[l, n] / lam[l] * du
            ga_o[l] += S[l, n] / lam[l] * du * du * dt
        for g in range(G):
            ks = tg == g
            psi[ks] = psi[ks] @ o.step_propagator(H0[g], Hc[g], eps_o[:, n], dt).T
    return psiT, chiT, eps_o, ga_o, psi


@pytest.mark.parametrize("N,K", [(64, 256), (32, 160), (16, 24)])
def test_window_variant_with_generators_and_missing_controls(N, K):
    """The window variant of the update kernel (all control operators Hermitian; round 3: the partial sums of step n + 1 are
    deposited with the last barrier of step n; N = 16, K = 24: the one-XCD placement runs it with wave 1 publishing) with G = 3
    generators of which one lacks control 1 (mu === nothing, /root/reference/src/optimize.jl:344) -- the carried c_l and the
    deposited sums must respect the mask of each workgroup's generator.  Against the serial restatement on the oracle."""
    import krotov_oracle as o
    rng = np.random.default_rng(1200 + N)
    NT, L, G = 7, 2, 3
    tg = rng.permutation(np.arange(K) % G).astype(np.int32)
    base = wl.dense_problem(N, NT, K, L, 930 + N, 0.05, 0.05)
    H0 = np.array([base["H0"] * (1.0 + 0.1 * g) + 0.05 * wl.random_hermitian(rng, N) for g in range(G)])
    Hc = np.array([[base["Hc"][l] * (1.0 - 0.07 * g) for l in range(L)] for g in range(G)])
    present = np.ones((G, L), np.uint8)
    present[1, 1] = 0
    tl, eps, S, lam, psi0, tgt = base["tlist"], base["eps"], base["S"], base["lam"], base["psi0"], base["tgt"]
    psiT_o, chiT, eps_o, ga_o, psi1_o = _serial_iteration(o, H0, Hc, present, tg, tl, eps, S, lam, psi0, tgt)
    with kr.HipEngine(H0, Hc, tl, K, traj_generator=tg, hc_present=present) as eng:
        info = eng.info()
        assert info["hermitian_controls_path"] == 1 and info["generic_update"] == 0, eng.plan()
        eng.set_initial_states(psi0)
        psiT = eng.forward(eps)
        assert np.max(np.abs(psiT - psiT_o)) < 1e-11
        eps_new, ga, psiT1 = eng.iteration(eps, chiT, S, lam)
    assert np.max(np.abs(eps_o - eps)) > 1e-6
    assert _rel(eps_new, eps_o) < TOL_PULSE and _rel(ga, ga_o) < TOL_GA and np.max(np.abs(psiT1 - psi1_o)) < TOL_STATE


@pytest.mark.parametrize("case", ["hermitian128", "hermitian200", "weak_dissipation", "strong_dissipation",
                                  "single_r1", "single_two_controls", "single_generators", "streamed"])
def test_economized_polynomial_in_rowsplit_kernels(oracle_c, monkeypatch, case):
    """The row-split and streamed-generator kernels (64 < N <= 256) evaluate the economized step polynomial too
    (round 3), in all three sweeps: for Hermitian generators by the rule of the register-resident kernels, and for generators that
    are only NEARLY Hermitian -- weak dissipation -- by the ellipse bound of kr_econ_rows_kernel (Crouzeix-Palencia on an ellipse
    around the numerical range).  With and without it (KROTOV_HIP_ECON_PAIR) the sweeps agree far inside the parity tolerances and
    meet the oracle at them; far from Hermitian no row saves a degree, every step takes Taylor's plan and the results are
    bitwise those without it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_parity as tp
    pair = True
    if case == "hermitian128":
        p = wl.dense_problem(128, 40, 96, 1, 881, 0.05, 2.0)
    elif case == "hermitian200":
        p = wl.dense_problem(200, 36, 48, 1, 882, 0.05, 2.0)
    elif case == "single_r1":              # one workgroup per trajectory, resident generator
        p, pair = wl.dense_problem(128, 36, 256, 1, 885, 0.05, 2.0), False
    elif case == "streamed":               # more trajectories than the row-split plans take: streamed-generator kernels
        p, pair = wl.dense_problem(200, 30, 70, 3, 888, 0.05, 2.0), False     # (three controls: with two, turns in the row-split kernels)
    elif case == "single_two_controls":    # R = 8 workgroups per trajectory, two controls
        p, pair = wl.dense_problem(200, 36, 24, 2, 886, 0.05, 2.0), False
    elif case == "single_generators":      # per-trajectory generators, one of them not Hermitian: its steps take Taylor's plan
        p, pair = wl.dense_problem(100, 34, 12, 2, 887, 0.05, 2.0), False
        H0 = np.stack([p["H0"], p["H0"] * 0.9 - 0.2j * np.diag(np.random.default_rng(8).uniform(0, 1, 100))])
        Hc = np.stack([p["Hc"], p["Hc"] * 1.1])
        p["H0"], p["Hc"], p["traj_generator"] = H0, Hc, np.arange(12) % 2
    else:
        p = wl.dense_problem(160, 36, 40, 1, 883, 0.05, 2.0)
        g = 0.002 if case == "weak_dissipation" else 0.3
        p["H0"] = p["H0"] - 1j * g * np.diag(np.random.default_rng(6).uniform(0, 1, 160))
    if case in ("hermitian200", "single_two_controls", "streamed"):   # a non-uniform time grid: the rows follow dt_n
        dts = np.diff(p["tlist"]) * np.random.default_rng(9).uniform(0.6, 1.4, len(p["tlist"]) - 1)
        p["tlist"] = np.concatenate([[0.0], np.cumsum(dts)])
    kw = {"traj_generator": p["traj_generator"]} if "traj_generator" in p else {}
    res = {}
    for mode in ["0", "1"]:
        monkeypatch.setenv("KROTOV_HIP_ECON_PAIR", mode)
        res[mode], info = tp._iterate(p, **kw)
        if case == "streamed":
            assert info["rowsplit_R"] == 0 and info["NP"] == 256 and info["generic_update"] == 0, info
        else:
            assert info["rowsplit_R"] >= 1 and info["KT"] == (2 if pair else 1), info
        assert info["economized_polynomial"] == int(mode), info
    if case == "strong_dissipation":
        assert np.array_equal(res["1"][0], res["0"][0]) and np.array_equal(res["1"][2], res["0"][2])
    else:
        assert _rel(res["1"][0], res["0"][0]) < 1e-12
        assert not np.array_equal(res["1"][2], res["0"][2])               # (a different polynomial did run)
        assert np.max(np.abs(res["1"][2] - res["0"][2])) < 2e-13 * max(1.0, np.max(np.abs(res["0"][2])))
    monkeypatch.setenv("KROTOV_HIP_ECON_PAIR", "1")
    if case != "single_generators":
        tp._run_case(oracle_c, p, check_storage=(case == "hermitian128"))
    else:   # per-trajectory generators: the serial restatement on the oracle's exact step propagator
        import krotov_oracle as o
        tg = p["traj_generator"]
        _, _, eps_o, ga_o, psi1_o = _serial_iteration(o, p["H0"], p["Hc"], np.ones((2, 2), np.uint8), tg, p["tlist"], p["eps"], p["S"], p["lam"],
                                                      p["psi0"], p["tgt"])
        assert _rel(res["1"][0], eps_o) < TOL_PULSE and np.max(np.abs(res["1"][2] - psi1_o)) < 1e-9


def test_options_travel_in_kr_config(monkeypatch):
    """Schedule options are part of kr_config (`options`, "NAME=integer;..."): what a caller behind the C ABI uses instead of the
    KROTOV_HIP_* environment variables, which stay the fallback.  An option in kr_config wins over the environment; a name the
    library does not know, or a value that is not an integer, fails kr_create with a message instead of being ignored."""
    p = wl.dense_problem(128, 6, 96, 1, 884, 0.05, 2.0)
    K = p["psi0"].shape[0]

    def plan(**kw):
        with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K, **kw) as eng:
            return eng.info()
    monkeypatch.delenv("KROTOV_HIP_RS_PAIR", raising=False)
    assert plan()["KT"] == 2                                             # pair mode by default at this shape
    assert plan(options={"RS_PAIR": 0})["KT"] == 1
    assert plan(options="KROTOV_HIP_RS_PAIR=0; ECON=0")["KT"] == 1       # prefix optional, several separators
    monkeypatch.setenv("KROTOV_HIP_RS_PAIR", "0")
    assert plan()["KT"] == 1                                             # the environment is the fallback ...
    assert plan(options="RS_PAIR=1")["KT"] == 2                          # ... and kr_config wins over it
    for bad in ["RS_PEAR=0", "RS_PAIR=yes", "RS_PAIR"]:
        with pytest.raises(Exception) as ei:
            plan(options=bad)
        assert "options" in str(ei.value) and bad in str(ei.value)


@pytest.mark.parametrize("N,NT,K,L,R", [(300, 6, 3, 1, 64), (400, 4, 8, 2, 32), pytest.param(512, 3, 2, 1, 64, marks=pytest.mark.soak), (260, 34, 4, 2, 64), (280, 4, 3, 4, 64)])
def test_state_dimension_up_to_512_on_the_rowsplit_kernels(oracle_c, N, NT, K, L, R):
    """256 < N <= 512 (round 3): the single-mode row-split kernels with NP = 512 -- 64 or 32 workgroups per trajectory, 8 or 16 rows
    each, the all-gather polled by two waves -- instead of the generic multi-pass kernels (N = 512, K = 4: 14.6 / 19 us per step
    against 6.3 / 1.6 ms), for up to 8 trajectories.  All three sweeps against the oracle; the last case non-Hermitian, across a
    rebuild of the H(eps) block."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_parity as tp
    p = wl.dense_problem(N, NT, K, L, 7700 + N, 0.02, 2.0)
    if N == 260:
        p["H0"] = p["H0"] - 0.01j * np.diag(np.random.default_rng(3).uniform(0, 1, N))
    info = tp._run_case(oracle_c, p, check_storage=(N == 300))
    assert info["NP"] == 512 and info["rowsplit_R"] == R and info["workgroups"] == K * R, info
    assert info["generic_known"] == 0 and info["generic_update"] == 0, info


@pytest.mark.parametrize("full", [False, pytest.param(True, marks=pytest.mark.soak)], ids=["one_shape_per_family", "all_shapes"])
def test_state_dimension_above_256_with_more_trajectories(oracle_c, full):
    """(round 5: the default selection runs one or two shapes per kernel family, the whole list is a soak leg.)  More trajectories than one NP = 512 / 1024 row-split launch carries (round 3).  The KNOWN-pulse sweeps run the row-split kernel
    in batches of trajectories (independent launches); the update sweep runs kr_rowsplit_turns_update -- every group of workgroups
    carries up to 16 trajectories in turn, H(eps) resident once, one exchange per step (N = 512, K = 32: 79 / 103 us per step against
    ~6 / 1.5 ms on the generic kernels) -- or, beyond 16 turns or with per-trajectory generators, the stepped
    generic update.  More than four controls (two beyond N = 512): the generic kernels everywhere, as before.  Same results."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_parity as tp
    # turns (264: across a rebuild of H(eps); 520 with 8: a multiple of 8 trajectories on TWO groups -- the one-XCD layout of the
    # workgroup ids goes by the groups of a launch, not by the trajectories)
    # ((600, 2, 3, 2): NP = 1024 with two controls in two turns -- kr_rowsplit_turns_update<1024, 8, 2>, round 4)
    for (N, NT, K, L, Kp) in [(300, 4, 20, 1, 24), (264, 34, 9, 2, 16), (520, 2, 3, 1, 4), (400, 3, 70, 2, 72), (516, 2, 8, 1, 8), (600, 2, 3, 2, 4)][(0 if full else 1):(6 if full else 3)]:
        p = wl.dense_problem(N, NT, K, L, 7800 + N, 0.02, 2.0)
        info = tp._run_case(oracle_c, p, check_storage=(N == 264))
        assert info["generic_known"] == 0 and info["generic_update"] == 0 and info["K_padded"] == Kp and info["workgroups"] == 256, info
    # 128 < N <= 256 with two controls and more than 64 trajectories: turns in groups of 8 workgroups instead of the streamed kernels
    for (N, NT, K, L, Kp) in [(256, 36, 128, 2, 128), (200, 4, 70, 2, 96), (160, 3, 250, 2, 256)][(0 if full else 1):(3 if full else 2)]:
        p = wl.dense_problem(N, NT, K, L, 7800 + N, 0.02, 2.0)
        info = tp._run_case(oracle_c, p, check_storage=(N == 200))
        assert info["NP"] == 256 and info["rowsplit_R"] == 8 and info["generic_update"] == 0 and info["K_padded"] == Kp, info
    for (K, Kp) in [(9, 12), (16, 16)][:(2 if full else 1)]:                       # three controls: 8 rows per workgroup, 4 trajectories at once, 3 / 4 turns
        p = wl.dense_problem(264, 4, K, 3, 7800 + 264 + K, 0.02, 2.0)
        info = tp._run_case(oracle_c, p, check_storage=False)
        assert info["generic_update"] == 0 and info["rowsplit_R"] == 64 and info["K_padded"] == Kp, info
    for (N, NT, K, L) in [(300, 2, 130, 1)]:                                       # seventeen turns: batched known-pulse sweeps + generic update
        p = wl.dense_problem(N, NT, K, L, 7800 + N + K, 0.02, 2.0)
        info = tp._run_case(oracle_c, p, check_storage=False)
        assert info["generic_known"] == 0 and info["generic_update"] == 1 and info["rowsplit_R"] in (32, 64), info
    for (N, NT, K, L) in [(258, 3, 3, 5), (520, 2, 1, 3)]:
        p = wl.dense_problem(N, NT, K, L, 7800 + N, 0.02, 2.0)
        info = tp._run_case(oracle_c, p, check_storage=False)
        assert info["generic_known"] == 1 and info["generic_update"] == 1, info


@pytest.mark.parametrize("N,NT,K,L", [(640, 4, 2, 1), (1024, 3, 1, 2), pytest.param(520, 36, 2, 2, marks=pytest.mark.soak)])
def test_state_dimension_up_to_1024_on_the_rowsplit_kernels(N, NT, K, L):
    """512 < N <= 1024 (round 3): NP = 1024, 128 workgroups per trajectory with 8 rows each, four polling waves in the all-gather, at
    most two trajectories.  Against the serial restatement on the NumPy oracle's exact step propagator (the C oracle's plain-loop
    Pade would take minutes at this size); the last case across a rebuild of the H(eps) block."""
    import krotov_oracle as o
    p = wl.dense_problem(N, NT, K, L, 7900 + N, 0.02, 2.0)
    H0, Hc = p["H0"][None], p["Hc"][None]
    tg = np.zeros(K, np.int32)
    psiT_o, chiT, eps_o, ga_o, psi1_o = _serial_iteration(o, H0, Hc, np.ones((1, L), np.uint8), tg, p["tlist"], p["eps"], p["S"], p["lam"],
                                                          p["psi0"], p["tgt"])
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K) as eng:
        info = eng.info()
        assert info["NP"] == 1024 and info["rowsplit_R"] == 128 and info["workgroups"] == 128 * K and info["generic_update"] == 0, eng.plan()
        eng.set_initial_states(p["psi0"])
        psiT = eng.forward(p["eps"])
        assert np.max(np.abs(psiT - psiT_o)) < TOL_STATE
        eps_new, ga, psiT1 = eng.iteration(p["eps"], chiT, p["S"], p["lam"])
    assert _rel(eps_new, eps_o) < TOL_PULSE and _rel(ga, ga_o) < TOL_GA and np.max(np.abs(psiT1 - psi1_o)) < TOL_STATE


@pytest.mark.parametrize("N,NT,K,L,fname,fid", [(300, 6, 20, 1, "ss", 1), (264, 5, 9, 2, "sm", 0), (256, 5, 128, 2, "re", 2), (520, 4, 3, 1, "sm", 0)])
def test_device_functional_chain_on_turns_plans(N, NT, K, L, fname, fid):
    """kr_iteration_functional_dev (tau, chi(T) on the device) on handles whose update sweep runs in turns and whose known-pulse
    sweeps run in batches (zero trajectories pad the last turn: they must not enter tau, J_T or the sums): two chained iterations,
    bitwise equal to the chain with chi(T) formed on the host."""
    import torch
    p = wl.dense_problem(N, NT, K, L, 6100 + N, 0.03, 2.0)
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K) as eng:
        assert "kr_rowsplit_turns_update" in eng.plan(), eng.plan()
        eng.set_initial_states(p["psi0"])
        psiT = eng.forward(p["eps"])
        eps = p["eps"].copy()
        for _ in range(2):
            _, _, chiT = wl.chi_and_JT(fname, psiT, p["tgt"])
            eps, ga, psiT = eng.iteration(eps, chiT, p["S"], p["lam"])
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K) as eng:
        eng.set_initial_states(p["psi0"])
        eng.set_target_states(p["tgt"])
        eng.set_update_params(p["S"], p["lam"])
        e0 = torch.from_numpy(p["eps"].copy()).cuda()
        e1 = torch.empty_like(e0)
        torch.cuda.synchronize()
        eng.forward_dev(e0.data_ptr())
        for _ in range(2):
            eng.iteration_functional_dev(fid, e0.data_ptr(), e1.data_ptr())
            e0, e1 = e1, e0
        eng.synchronize()
        eps_dev, psiT_dev = e0.cpu().numpy(), eng.final_states()
    assert np.max(np.abs(eps - p["eps"])) > 1e-7
    assert _rel(eps_dev, eps) < 1e-12 and np.max(np.abs(psiT_dev - psiT)) < 1e-12
