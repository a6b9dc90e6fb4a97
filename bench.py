#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X Krotov engine.

A "step" = one full Krotov iteration of the hot path on the headline configuration
(BASELINE.json: N=64 dense Hamiltonian, N_T=1000 intervals, K=256 trajectories, L=2 controls,
fp64 complex), chained in HBM: tau/chi(T) on the device -> backward sweep -> sequential
update+forward sweep (kr_iteration_functional_dev).  Pulses, states, targets and storages are
resident in HBM when the timed region starts.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3|C2|C4|C5] [--scaling strong|weak]

  N > 1: launched by torch.distributed.run, one rank per GPU.  DEFAULT = STRONG scaling: the ensemble BASELINE.json
  names (C3: K=256 in total, C4: K=1024, C5: K=64) is block-partitioned over the N GPUs -- the reference threads over a
  FIXED set of trajectories (/root/reference/src/optimize.jl:182,303,321,360) --, so the metric string is the same at
  every N and J_T before / after must not depend on N.  `--scaling weak` keeps the per-GPU count fixed instead
  (K_global = N x K, the metric string then names that larger ensemble).  The per-step sum of the L update
  integrands (optimize.jl:339-349) crosses GPUs through in-kernel peer mailboxes over xGMI, or through
  one ncclAllReduce per time step (`--exchange rccl`, and the fallback when the mailboxes cannot be wired).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel =
the update sweep, durations from HIP events recorded inside libkrotov_hip on its stream) and
`cpu_baseline` (a bounded sample of the same workload executed the way the reference executes it, on one host thread).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector = matrix peak (spec; MI355X_MICROARCH.md has no fp64 row; measured on the box:
                          # profiles/r02_ubench_mfma4x4.txt, 512 flop / 16 cycles / SIMD)
HBM_PEAK_GBS = 8000.0
METRIC_FMT = "Krotov iters/sec + state·timesteps/sec, N=%d nt=%d K=%d traj"   # BASELINE.json's metric, sizes filled in


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3", help="C2 | C3 (headline, default) | C4 | C5 | D:N,NT,K,L (workloads.py)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong (default) = the config's K trajectories in total, sharded over the GPUs; "
                         "weak = the config's K on EVERY GPU (K_global = N x K)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exchange", default="auto", choices=["auto", "mailbox", "rccl"],
                    help="N > 1: how the per-step sum of the L update integrands crosses GPUs: in-kernel peer mailboxes over "
                         "xGMI (default; falls back to rccl if they cannot be wired) or one ncclAllReduce per time step")
    ap.add_argument("--traj", type=int, default=0,
                    help="(testing) use only the first TRAJ trajectories of the config (strong: in total; weak: per GPU)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline
# ---------------------------------------------------------------------------------------------------------------------
_PADE = {3: (120., 60., 12., 1.),
         5: (30240., 15120., 3360., 420., 30., 1.),
         7: (17297280., 8648640., 1995840., 277200., 25200., 1512., 56., 1.),
         9: (17643225600., 8821612800., 2075673600., 302702400., 30270240., 2162160., 110880., 3960., 90., 1.),
         13: (64764752532480000., 32382376266240000., 7771770303897600., 1187353796428800., 129060195264000.,
              10559470521600., 670442572800., 33522128640., 1323241920., 40840800., 960960., 16380., 182., 1.)}
_THETA = (1.495585217958292e-2, 2.539398330063230e-1, 9.504178996162932e-1, 2.097847961257068e0, 5.371920351148152e0)


def expm_lapack(A):
    """exp(A) the way Julia's LinearAlgebra.exp!(::Matrix) computes it [UPSTREAM-RECALLED]: Higham (2005) scaling and squaring,
    Pade order 3/5/7/9/13 chosen from ||A||_1, products by BLAS ZGEMM (numpy @), the final solve by LAPACK ZGESV
    (numpy.linalg.solve).  The same algorithm as oracle/krotov_oracle.c:orc_expm, with library kernels instead of loops."""
    n = A.shape[0]
    nA = np.abs(A).sum(axis=0).max()
    I = np.eye(n, dtype=A.dtype)
    if nA <= _THETA[3]:
        m = 3 if nA <= _THETA[0] else 5 if nA <= _THETA[1] else 7 if nA <= _THETA[2] else 9
        b = _PADE[m]
        A2 = A @ A
        P = A2
        T = b[1] * I
        V = b[0] * I
        for k in range(1, m // 2 + 1):
            T = T + b[2 * k + 1] * P
            V = V + b[2 * k] * P
            if 2 * (k + 1) <= m:
                P = P @ A2
        U = A @ T
        s = 0
    else:
        r = nA / _THETA[4]
        s = max(0, int(np.ceil(np.log2(r)))) if r > 1.0 else 0
        A = A * 2.0 ** (-s)
        b = _PADE[13]
        A2 = A @ A
        A4 = A2 @ A2
        A6 = A4 @ A2
        U = A @ (A6 @ (b[13] * A6 + b[11] * A4 + b[9] * A2) + b[7] * A6 + b[5] * A4 + b[3] * A2 + b[1] * I)
        V = A6 @ (b[12] * A6 + b[10] * A4 + b[8] * A2) + b[6] * A6 + b[4] * A4 + b[2] * A2 + b[0] * I
    E = np.linalg.solve(V - U, V + U)
    for _ in range(s):
        E = E @ E
    return E


def cpu_reference_order_lapack(p, Ks, NTs, log):
    """One Krotov iteration in the REFERENCE'S execution order (use_threads = false, /root/reference/src/workspace.jl:66) with
    library linear algebra on ONE thread: per trajectory and per step build H(eps_n), a dense LAPACK/BLAS expm, one matvec
    (prop_step! with ExpProp, optimize.jl:257,309,361); the update integrand summed l-outer / k-inner, serially
    (optimize.jl:340-349).  This is the closest thing to what Julia executes that this box can time."""
    from threadpoolctl import threadpool_limits
    H0, Hc = p["H0"], p["Hc"]
    L = Hc.shape[0]
    tl = p["tlist"][:NTs + 1]
    eps, S, lam = p["eps"][:, :NTs], p["S"][:, :NTs], p["lam"]
    psi0, tgt = p["psi0"][:Ks], p["tgt"][:Ks]
    N = H0.shape[0]

    def step(e, sdt, adj, v):
        H = H0.copy()
        for l in range(L):
            H = H + e[l] * Hc[l]
        if adj:
            H = H.conj().T
        return expm_lapack(-1j * sdt * H) @ v

    with threadpool_limits(limits=1):
        psi = psi0.copy()
        for n in range(NTs):                        # initial forward sweep (not timed: the iteration is the unit)
            U = expm_lapack(-1j * (tl[n + 1] - tl[n]) * (H0 + np.tensordot(eps[:, n], Hc, axes=(0, 0))))
            psi = psi @ U.T
        from krotov_jl_amd import workloads as wl
        chi = wl.chi_and_JT(p["functional"], psi, tgt)[2]
        t0 = time.perf_counter()
        X = np.empty((NTs + 1, Ks, N), complex)
        X[NTs] = chi
        for k in range(Ks):                         # backward sweep, trajectory by trajectory (optimize.jl:303-317)
            c = chi[k]
            for n in range(NTs - 1, -1, -1):
                c = step(eps[:, n], -(tl[n + 1] - tl[n]), True, c)
                X[n, k] = c
        psi = psi0.copy()
        e_new = eps.copy()
        for n in range(NTs):                        # update sweep (optimize.jl:328-370)
            dt = tl[n + 1] - tl[n]
            for l in range(L):
                du = 0.0
                for k in range(Ks):
                    du += np.vdot(X[n, k], Hc[l] @ psi[k]).imag
                e_new[l, n] = eps[l, n] + S[l, n] / lam[l] * du
            for k in range(Ks):
                psi[k] = step(e_new[:, n], dt, False, psi[k])
        t_it = time.perf_counter() - t0
    log("cpu reference_order_lapack_1thread: %.3f s/iter on sample K=%d NT=%d" % (t_it, Ks, NTs))
    return {"value": 2.0 * Ks * NTs / t_it, "iter_s": t_it, "cores": 1, "K": Ks, "NT": NTs}


def cpu_baseline(p, log):
    """CPU baseline ("port": Julia cannot run here) on BOUNDED samples of the same workload -- same operators and pulses,
    fewer trajectories and time steps.  Unit: state*timesteps/s = 2*K*NT / t_iteration, like the GPU metric.
    `value` = the reference's execution order with LAPACK/BLAS linear algebra on one thread (what Julia's ExpProp does per
    prop_step!); the plain-C oracle variants (naive loops) and an optimised shared-propagator ZGEMM variant are reported
    beside it under `variants`, each with the threads it really used."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_c as oc
    from krotov_jl_amd import workloads as wl
    N = p["H0"].shape[0]
    K_all, NT_all = p["psi0"].shape[0], len(p["tlist"]) - 1
    nth = oc.max_threads()
    out = {}
    big = N >= 64
    huge = N > 64
    # primary: reference order, LAPACK expm, 1 thread; >= K=64 x 200 intervals at N=64 (review of round 2)
    Ks, NTs = (min(8, K_all), min(24, NT_all)) if huge else (min(64, K_all), min(200 if big else 500, NT_all))
    out["reference_order_lapack_1thread"] = cpu_reference_order_lapack(p, Ks, NTs, log)
    # (name, mode, threads, trajectories, intervals): the threaded variants get enough trajectories to occupy the cores
    variants = [("reference_order_naive_c_1thread", 0, 1, min(32, K_all), min(64 if big else 500, NT_all)),
                ("reference_order_naive_c_allcores", 0, nth, min(max(32, nth), K_all), min(48 if big else 500, NT_all)),
                ("shared_propagator_naive_c_allcores", 1, nth, min(max(32, nth), K_all), min(200 if big else 500, NT_all))]
    for name, mode, nt_, Kv, NTv in variants:
        if huge:
            Kv, NTv = min(Kv, 16), min(NTv, 8)
        tl = p["tlist"][:NTv + 1]
        eps, S = p["eps"][:, :NTv], p["S"][:, :NTv]
        psi0, tgt = p["psi0"][:Kv], p["tgt"][:Kv]
        psiT, _ = oc.forward(p["H0"], p["Hc"], tl, eps, psi0, store=False, mode=1, nthreads=nth)
        _, _, chiT = wl.chi_and_JT(p["functional"], psiT, tgt)
        t0 = time.perf_counter()
        oc.iteration(p["H0"], p["Hc"], tl, eps, S, p["lam"], psi0, chiT, mode=mode, nthreads=nt_)
        t_it = time.perf_counter() - t0
        used = min(nt_, Kv)   # OpenMP static schedule over the trajectories: no more threads than trajectories do work
        out[name] = {"value": 2.0 * Kv * NTv / t_it, "iter_s": t_it, "cores": used, "K": Kv, "NT": NTv}
        log("cpu %s: %.3f s/iter on sample K=%d NT=%d (%d threads)" % (name, t_it, Kv, NTv, used))
    # "optimised CPU": one LAPACK/Pade expm (scipy.linalg.expm) per step + OpenBLAS ZGEMM over all trajectories, NumPy's
    # thread pool; the update integrand as one batched ZGEMM + einsum (not the reference's serial k loop)
    try:
        out["shared_propagator_blas_zgemm_allcores"] = cpu_blas_variant(p, min(256, K_all), min(100 if big else 500, NT_all) if not huge else 16, log)
    except Exception as exc:  # noqa: BLE001 -- the BLAS variant is a courtesy figure
        log("cpu blas variant failed: %r" % (exc,))
    base = out["reference_order_lapack_1thread"]
    return {"value": base["value"], "unit": "state*timesteps/s", "cores": 1, "kind": "port",
            "sample": "same H0/Hc/pulses, K=%d of %d trajectories, first %d of %d intervals; one Krotov iteration in the "
                      "reference's execution order (use_threads=false): per trajectory and per step H(eps_n), dense Pade "
                      "expm with BLAS ZGEMM + LAPACK ZGESV (Higham 2005, as Julia's exp(::Matrix)), matvec; serial l-outer/"
                      "k-inner update sum; NumPy driver, BLAS pinned to 1 thread"
                      % (base["K"], K_all, base["NT"], NT_all),
            "variants": {k: {"value": v["value"], "cores": v["cores"], "sample_K": v["K"], "sample_NT": v["NT"]}
                         for k, v in out.items()}}


def cpu_blas_variant(p, Ks, NTs, log):
    """One Krotov iteration with BLAS/LAPACK doing the arithmetic (what a tuned CPU code would run): shared step
    propagator from scipy.linalg.expm, states advanced by one ZGEMM per step, update integrand by ZGEMM + einsum."""
    from scipy.linalg import expm
    H0, Hc = p["H0"], p["Hc"]
    L = Hc.shape[0]
    tl = p["tlist"][:NTs + 1]
    eps, S, lam = p["eps"][:, :NTs], p["S"][:, :NTs], p["lam"]
    psi0, tgt = p["psi0"][:Ks], p["tgt"][:Ks]

    def gen(e):
        return H0 + np.tensordot(e, Hc, axes=(0, 0))
    psi = psi0.copy()
    for n in range(NTs):
        psi = psi @ expm(-1j * gen(eps[:, n]) * (tl[n + 1] - tl[n])).T
    tau = np.einsum("kn,kn->k", np.conj(tgt), psi)
    chi = (tau / Ks)[:, None] * tgt
    t0 = time.perf_counter()
    X = np.empty((NTs + 1, Ks, H0.shape[0]), complex)
    X[NTs] = chi
    for n in range(NTs - 1, -1, -1):
        chi = chi @ expm(-1j * gen(eps[:, n]).conj().T * (-(tl[n + 1] - tl[n]))).T
        X[n] = chi
    psi = psi0.copy()
    e_new = eps.copy()
    for n in range(NTs):
        for l in range(L):
            du = np.einsum("kn,kn->", np.conj(X[n]), psi @ Hc[l].T).imag
            e_new[l, n] = eps[l, n] + S[l, n] / lam[l] * du
        psi = psi @ expm(-1j * gen(e_new[:, n]) * (tl[n + 1] - tl[n])).T
    t_it = time.perf_counter() - t0
    log("cpu shared_propagator_blas_zgemm_allcores: %.3f s/iter on sample K=%d NT=%d" % (t_it, Ks, NTs))
    return {"value": 2.0 * Ks * NTs / t_it, "iter_s": t_it, "cores": os.cpu_count() or 1, "K": Ks, "NT": NTs}


# ---------------------------------------------------------------------------------------------------------------------
def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    def log(msg):
        if rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    import torch
    if rank == 0 and not os.path.exists(os.path.join(ROOT, "krotov.jl_amd", "lib", "libkrotov_hip.so")):
        import __graft_entry__ as ge   # clean checkout: build the HIP library first (hipcc, gfx950)
        ge.build()
    import krotov_jl_amd as kr
    from krotov_jl_amd import workloads as wl
    from krotov_jl_amd import distributed as krd

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # KROTOV_BENCH_BACKEND=gloo lets several ranks share one GPU (rehearsal of the N > 1 path on a 1-GPU box)
    backend = os.environ.get("KROTOV_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    red_dev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    p = wl.config(args.config)
    if args.traj > 0:
        p["psi0"], p["tgt"] = p["psi0"][:args.traj], p["tgt"][:args.traj]
    K_cfg, N = p["psi0"].shape
    L, NT = p["eps"].shape
    if world > 1 and args.scaling == "weak":
        # weak scaling: every rank carries its own K_cfg trajectories (same operators and pulses, rank-specific states),
        # the functional is normalised by the global K.
        K_global, K_local = K_cfg * world, K_cfg
        q = wl.dense_problem(N, NT, K_local, L, 64001 + 1000 * rank, p["tlist"][1] - p["tlist"][0], p["lam"][0])
        p["psi0"], p["tgt"] = q["psi0"], q["tgt"]
    else:
        # strong scaling: the named ensemble, block-partitioned (the loops the reference marks @threadsif)
        K_global = K_cfg
        if K_global < world:
            raise SystemExit("bench.py: %d trajectories cannot be sharded over %d GPUs" % (K_global, world))
        k0, k1 = krd.shard_range(K_global, rank, world)
        K_local = k1 - k0
        p["psi0"], p["tgt"] = np.ascontiguousarray(p["psi0"][k0:k1]), np.ascontiguousarray(p["tgt"][k0:k1])
    scaling = "weak" if (world > 1 and args.scaling == "weak") else "strong"

    parallelism = "trajectories block-partitioned over %d GPU(s), %s scaling (K_global=%d, %d on this rank); per-step sum of %d doubles" % (
        world, scaling, K_global, K_local, L)
    exchange_used = "none (one GPU)"
    eng = None
    max_wg = int(os.environ.get("KROTOV_BENCH_MAX_WG", "0"))
    if world > 1:
        from krotov_jl_amd import _capi as capi

        def all_ok(ok):
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=red_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag.item()) == 1

        def attempt(what, fn):
            """LOCAL work only (no collective inside): returns (ok, value); every rank then agrees through all_ok before any
            collective that depends on it is entered -- a rank that failed never leaves the others waiting in a collective it
            skipped (round-2 advisor finding)."""
            try:
                return True, fn()
            except Exception as exc:  # noqa: BLE001
                print("[bench] rank %d: %s failed: %r" % (rank, what, exc), file=sys.stderr, flush=True)
                return False, None

        def close_engine():
            nonlocal eng
            if eng is not None:
                eng.close()
            eng = None

        def setup_mailbox():
            """peer mailboxes over xGMI (IPC-mapped): create -> export -> [all_gather] -> connect, agreeing after every local phase"""
            nonlocal eng
            ok, e = attempt("kr_create (mailbox exchange)", lambda: kr.HipEngine(
                p["H0"], p["Hc"], p["tlist"], K_local, device=local_rank, rank=rank, world=world, K_global=K_global,
                exchange=capi.KR_EXCHANGE_MAILBOX, max_workgroups=max_wg))
            eng = e
            if not all_ok(ok):
                return False
            ok, mine = attempt("kr_ipc_export", eng.ipc_export)
            if not all_ok(ok):
                return False
            handles = [None] * world
            dist.all_gather_object(handles, mine)          # collective: every rank is here
            ok, _ = attempt("kr_ipc_connect", lambda: eng.ipc_connect(handles))
            if not all_ok(ok):
                return False
            dist.barrier()
            return True

        def setup_rccl():
            nonlocal eng
            ok, e = attempt("kr_create (RCCL exchange)", lambda: kr.HipEngine(
                p["H0"], p["Hc"], p["tlist"], K_local, device=local_rank, rank=rank, world=world, K_global=K_global,
                exchange=capi.KR_EXCHANGE_RCCL))
            eng = e
            if not all_ok(ok):
                return False
            ok, uid = (attempt("kr_rccl_get_unique_id", kr.HipEngine.rccl_unique_id) if rank == 0 else (True, None))
            ids = [uid if ok else None]
            dist.broadcast_object_list(ids, src=0)         # collective: every rank is here, whatever rank 0 got
            if ids[0] is None:
                return False
            ok, _ = attempt("kr_rccl_init", lambda: eng.rccl_init(ids[0]))   # (ncclCommInitRank: entered by all ranks)
            return all_ok(ok)

        want = args.exchange
        if want in ("auto", "mailbox"):
            if setup_mailbox():
                exchange_used = "in-kernel peer mailboxes over xGMI (hipIpc-mapped), one launch per sweep"
            else:
                close_engine()
                if want == "mailbox":
                    raise SystemExit("bench.py: --exchange mailbox requested but the peer mailboxes could not be wired on every rank")
                log("falling back to the per-step RCCL all-reduce on all ranks")
                want = "rccl"
        if want == "rccl":
            if not setup_rccl():
                # no silent degradation to independent replicas: a scaling harness must not read that as a scaling point
                raise SystemExit("bench.py: neither peer mailboxes nor the RCCL exchange could be set up on every rank")
            exchange_used = "ncclAllReduce (RCCL) of %d doubles per time step between per-step kernels" % L
    if eng is None:
        eng = kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K_local, device=local_rank, K_global=K_global)
    func = {"sm": 0, "ss": 1, "re": 2}[p["functional"]]
    dev = torch.device("cuda", local_rank)

    def global_JT():
        """J_T of the WHOLE ensemble from this rank's final states (host side, outside the timed region)"""
        tau = wl.chi_and_JT(p["functional"], eng.final_states(), p["tgt"], K_global)[0]
        if dist is not None:
            parts = [None] * world
            dist.all_gather_object(parts, tau)
            tau = np.concatenate(parts)
        if p["functional"] == "sm":
            return 1.0 - abs(np.sum(tau) / K_global) ** 2
        if p["functional"] == "ss":
            return 1.0 - np.sum(np.abs(tau) ** 2) / K_global
        return 1.0 - np.sum(tau).real / K_global

    def prime(e):
        """states, targets, update parameters, initial forward sweep; returns J_T of the guess"""
        e.set_initial_states(p["psi0"])
        e.set_target_states(p["tgt"])
        e.set_update_params(p["S"], p["lam"])
        a_ = torch.from_numpy(np.ascontiguousarray(p["eps"])).to(dev)
        b_ = torch.empty_like(a_)
        torch.cuda.synchronize()
        e.forward_dev(a_.data_ptr())
        e.synchronize()
        return a_, b_, global_JT()

    eps_a, eps_b, J0 = prime(eng)
    if world > 1 and "mailboxes" in exchange_used:
        # Trial sweep: the in-kernel mailboxes have never run across real xGMI links in the build environment.  If ANY rank's
        # bounded spin gives up (KR_ERR_TIMEOUT) all ranks move to the RCCL carrier together -- a slower but plain path.
        dist.barrier()

        def trial():
            eng.iteration_functional_dev(func, eps_a.data_ptr(), eps_b.data_ptr())
            eng.synchronize()
        ok, _ = attempt("trial sweep through the peer mailboxes", trial)

        def pulses_agree():
            """every rank sums the ranks' partial sums in rank order, so the updated pulses must be BITWISE equal on all ranks;
            a stale or torn mailbox read would show here (NaN compares unequal: also a mismatch)"""
            lo, hi = eps_b.to(red_dev).clone(), eps_b.to(red_dev).clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            same = bool(torch.equal(lo, hi)) and bool(torch.isfinite(lo).all())
            if not same:
                log("the ranks' updated pulses differ after the trial sweep through the peer mailboxes")
            return same
        if all_ok(ok) and pulses_agree():
            eps_a, eps_b, J0 = prime(eng)      # restart from the guess: the trial iteration is not part of the run
        elif args.exchange == "mailbox":
            raise SystemExit("bench.py: --exchange mailbox requested but the trial sweep through the peer mailboxes failed")
        else:
            log("falling back to the per-step RCCL all-reduce on all ranks")
            close_engine()
            if not setup_rccl():
                raise SystemExit("bench.py: the mailbox trial failed and the RCCL exchange could not be set up on every rank")
            exchange_used = "ncclAllReduce (RCCL) of %d doubles per time step between per-step kernels (mailbox trial failed)" % L
            eps_a, eps_b, J0 = prime(eng)

    def sync_all():
        eng.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def one_iteration():
        nonlocal eps_a, eps_b
        eng.iteration_functional_dev(func, eps_a.data_ptr(), eps_b.data_ptr())
        eps_a, eps_b = eps_b, eps_a   # pulse ping-pong, optimize.jl:216

    sync_all()   # all ranks enter the first cross-GPU sweep together (bounded in-kernel spins)
    for _ in range(args.warmup):
        one_iteration()
    sync_all()
    eng.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_iteration()
    sync_all()
    elapsed = time.perf_counter() - t0
    tm = eng.timings_ms()
    eng.enable_timing(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    J1 = global_JT()
    info = eng.info()
    plan = eng.plan()

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        units = 2.0 * K_global * NT     # one backward + one forward state-step per trajectory per interval, ALL ranks
        value = units * args.steps / elapsed
        # dominant kernel: the update sweep.  Algorithmic flops per (trajectory, interval) unit of that
        # kernel with a precomputed step propagator: forward apply 8N^2 + L mu-matvecs 8N^2 L
        # + L dots 8N L.  Propagator construction / polynomial order is overhead, not counted.  Per LAUNCH = per rank.
        upd_ms, upd_n = tm["update"]
        bw_ms, bw_n = tm["backward"]
        flops_upd = K_local * NT * (8.0 * N * N * (1 + L) + 8.0 * N * L)
        flops_bw = K_local * NT * (8.0 * N * N)
        avg_upd = upd_ms / max(upd_n, 1)
        avg_bw = bw_ms / max(bw_n, 1)
        achieved = flops_upd / (avg_upd * 1e-3) / 1e12 if upd_n else None
        # HBM bytes of the dominant kernel per launch from the committed PMC passes (tools/profile.sh on this very
        # command; bench.py itself does not run under rocprofv3).  Only quoted for the configuration it was taken on.
        traffic, traffic_src = None, None
        try:
            for name in ("r05_traffic_all.json", "r04_traffic_all.json", "r03_traffic_all.json", "r02_traffic_all.json", "r01_traffic.json"):
                path = os.path.join(ROOT, "profiles", name)
                if not os.path.exists(path):
                    continue
                with open(path) as f:
                    tj = json.load(f)
                for entry in (tj if isinstance(tj, list) else [tj]):
                    if entry.get("config") == args.config and world == 1 and args.traj == 0 and traffic is None:
                        traffic, traffic_src = entry["traffic_bytes_per_launch"], entry["source"]
        except (OSError, ValueError, KeyError):
            pass
        bytes_upd = K_local * NT * 32.0 * N
        mfma_sweep = bool(info.get("matrix_core_sweep", 0))
        kernel_name = ("kr_mf_update" if mfma_sweep else
                       ("kr_rowsplit_pair_sweep" if "kr_rowsplit_pair_sweep" in (plan or "") else "kr_rowsplit_sweep")
                       if info.get("rowsplit_R", 0) else "kr_sweep_update")
        line = {
            "metric": METRIC_FMT % (N, NT, K_global),
            "value": value,
            "unit": "state*timesteps/s",
            "iters_per_sec": args.steps / elapsed,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%s (%s): N=%d %s generator + %d control operator(s), N_T=%d, K=%d trajectories in total "
                                   "(%d on each of %d GPU(s)), J_T_%s, lambda_a=%g; one Krotov iteration per step"
                                   % (args.config, p["name"], N,
                                      "dense Hermitian" if np.allclose(p["H0"], p["H0"].conj().T) else "dense non-Hermitian (Liouvillian)",
                                      L, NT, K_global, K_local, world, p["functional"], p["lam"][0]),
                       "K_global": K_global, "K_per_gpu": K_local,
                       "parallelism": parallelism, "exchange": exchange_used,
                       "engine": info, "plan": plan},
            # bound names the unit the dominant kernel computes on: "mfma" = fp64 matrix cores (v_mfma_f64_4x4x4_4b_f64),
            # "valu_fp64" = fp64 vector FMAs; both have the same 78.6 TFLOP/s peak on MI355X.  What BINDS is latency.
            "roofline": {"bound": "mfma" if mfma_sweep else "valu_fp64", "binding_limit": "latency (N_T dependent steps)",
                         "kernel": kernel_name, "achieved": achieved,
                         "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": (achieved / FP64_PEAK_TFLOPS) if achieved else None, "traffic": traffic,
                         "traffic_unit": "bytes per launch (PMC: 2*FETCH_SIZE + WRITE_SIZE); a PROFILE figure read from the "
                                         "committed rocprofv3 --pmc summary of this command, not measured by this run",
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": bytes_upd,
                         "hbm_achieved_GBs": bytes_upd / (avg_upd * 1e-3) / 1e9 if upd_n else None, "hbm_peak_GBs": HBM_PEAK_GBS,
                         "avg_launch_ms": avg_upd, "launches": upd_n,
                         "algorithmic_flops_per_launch": flops_upd,
                         "backward_kernel": {"avg_launch_ms": avg_bw, "launches": bw_n,
                                             "achieved": (flops_bw / (avg_bw * 1e-3) / 1e12) if bw_n else None},
                         "note": "per launch = this rank's shard; fp64 matrix peak = vector peak on MI355X; achieved counts "
                                 "ALGORITHMIC flops (precomputed-propagator model, SURVEY 8d); the binding limit is sequential "
                                 "latency: N_T dependent steps, each = (1+L) products + one all-to-all sum over the workgroups "
                                 "(+ one hop between GPUs) + (m-1) polynomial rounds (m ~ 7-8)"},
            "J_T": {"before": float(J0), "after": float(J1)},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(wl.config(args.config), log)
        print(json.dumps(line), flush=True)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
