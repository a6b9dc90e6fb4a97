#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X Krotov engine.

A "step" = one full Krotov iteration of the hot path on the headline configuration
(BASELINE.json: N=64 dense Hamiltonian, N_T=1000 intervals, K=256 trajectories, L=2 controls,
fp64 complex), chained in HBM: tau/chi(T) on the device -> backward sweep -> sequential
update+forward sweep (kr_iteration_functional_dev).  Pulses, states, targets and storages are
resident in HBM when the timed region starts.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3|C2|C4]
  N > 1: launched by torch.distributed.run, one rank per GPU; trajectories are sharded
  (K per GPU fixed -> weak scaling), the per-step sum of the L update integrands crosses GPUs
  through in-kernel peer mailboxes over xGMI.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel =
the update sweep, durations from HIP events recorded inside libkrotov_hip on its stream) and
`cpu_baseline` (the plain-C oracle timed on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector = matrix peak (spec; MI355X_MICROARCH.md has no fp64 row)
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3", help="C2 | C3 (headline, default) | C4 | C5 (workloads.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exchange", default="auto", choices=["auto", "mailbox", "rccl"],
                    help="N > 1: how the per-step sum of the L update integrands crosses GPUs: in-kernel peer mailboxes over "
                         "xGMI (default; falls back to rccl if they cannot be wired) or one ncclAllReduce per time step")
    ap.add_argument("--traj", type=int, default=0, help="(testing) override the trajectories per GPU of the config")
    return ap.parse_args()


def cpu_baseline(p, log):
    """Oracle ("port") on a bounded sample of the same workload: same operators and pulses, fewer
    trajectories and time steps, executed the way the reference executes it (one dense expm per
    trajectory per step, serial update sum; use_threads=false -> 1 core).  Unit: state*timesteps/s
    = 2*K*NT / t_iteration, like the GPU metric."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_c as oc
    from krotov_jl_amd import workloads as wl
    N = p["H0"].shape[0]
    K_all, NT_all = p["psi0"].shape[0], len(p["tlist"]) - 1
    nth = oc.max_threads()
    # (name, mode, threads, trajectories, intervals): the threaded variants get enough trajectories to occupy the cores
    variants = [("reference_order_1thread", 0, 1, min(32, K_all), min(64 if N >= 64 else 500, NT_all)),
                ("reference_order_allcores", 0, nth, min(max(32, nth), K_all), min(48 if N >= 64 else 500, NT_all)),
                ("shared_propagator_allcores", 1, nth, min(max(32, nth), K_all), min(200 if N >= 64 else 500, NT_all))]
    out = {}
    for name, mode, nt_, Ks, NTs in variants:
        tl = p["tlist"][:NTs + 1]
        eps, S = p["eps"][:, :NTs], p["S"][:, :NTs]
        psi0, tgt = p["psi0"][:Ks], p["tgt"][:Ks]
        t0 = time.perf_counter()
        psiT, _ = oc.forward(p["H0"], p["Hc"], tl, eps, psi0, store=False, mode=mode, nthreads=nt_)
        t_fw = time.perf_counter() - t0
        _, _, chiT = wl.chi_and_JT(p["functional"], psiT, tgt)
        t0 = time.perf_counter()
        oc.iteration(p["H0"], p["Hc"], tl, eps, S, p["lam"], psi0, chiT, mode=mode, nthreads=nt_)
        t_it = time.perf_counter() - t0
        out[name] = {"value": 2.0 * Ks * NTs / t_it, "iter_s": t_it, "cores": nt_, "K": Ks, "NT": NTs}
        log("cpu %s: %.3f s/iter on sample K=%d NT=%d (fw %.2fs)" % (name, t_it, Ks, NTs, t_fw))
    # "optimised CPU": one LAPACK/Pade expm (scipy.linalg.expm) per step + OpenBLAS ZGEMM over all trajectories, NumPy's
    # thread pool; the update integrand as one batched ZGEMM + einsum (not the reference's serial k loop)
    try:
        out["blas_zgemm_expm_allcores"] = cpu_blas_variant(p, min(256, K_all), min(100 if N >= 64 else 500, NT_all), log)
    except Exception as exc:  # noqa: BLE001 -- the BLAS variant is a courtesy figure
        log("cpu blas variant failed: %r" % (exc,))
    Ks, NTs = out["reference_order_1thread"]["K"], out["reference_order_1thread"]["NT"]
    base = out["reference_order_1thread"]
    return {"value": base["value"], "unit": "state*timesteps/s", "cores": 1, "kind": "port",
            "sample": "same H0/Hc/pulses, K=%d of %d trajectories, first %d of %d intervals; one Krotov "
                      "iteration, dense Pade expm per trajectory per step (reference execution order)"
                      % (Ks, p["psi0"].shape[0], NTs, len(p["tlist"]) - 1),
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
    log("cpu blas_zgemm_expm_allcores: %.3f s/iter on sample K=%d NT=%d" % (t_it, Ks, NTs))
    return {"value": 2.0 * Ks * NTs / t_it, "iter_s": t_it, "cores": os.cpu_count() or 1, "K": Ks, "NT": NTs}


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
    K_local, N = p["psi0"].shape
    L, NT = p["eps"].shape
    K_global = K_local * world
    if world > 1:
        # weak scaling: every rank carries its own K_local trajectories (same operators and pulses,
        # rank-specific states), the functional is normalised by the global K.
        q = wl.dense_problem(N, NT, K_local, L, 64001 + 1000 * rank, p["tlist"][1] - p["tlist"][0], p["lam"][0])
        p["psi0"], p["tgt"] = q["psi0"], q["tgt"]

    parallelism = "trajectories sharded over %d GPU(s); per-step sum of %d doubles" % (world, L)
    exchange_used = "none (one GPU)"
    eng = None
    if world > 1:
        from krotov_jl_amd import _capi as capi

        def all_ok(ok):
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=red_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag.item()) == 1
        want = args.exchange
        if want in ("auto", "mailbox"):
            # peer mailboxes over xGMI (IPC-mapped); every rank must succeed
            ok = True
            try:
                eng = kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K_local, device=local_rank, rank=rank, world=world,
                                   K_global=K_global, exchange=capi.KR_EXCHANGE_MAILBOX,
                                   max_workgroups=int(os.environ.get("KROTOV_BENCH_MAX_WG", "0")))
                krd.connect_peers(eng, dist)
            except Exception as exc:  # noqa: BLE001
                log("rank %d: peer mailbox set-up failed: %r" % (rank, exc))
                ok = False
            if all_ok(ok):
                exchange_used = "in-kernel peer mailboxes over xGMI (hipIpc-mapped), one launch per sweep"
            else:
                if eng is not None:
                    eng.close()
                eng = None
                if want == "mailbox":
                    raise SystemExit("bench.py: --exchange mailbox requested but the peer mailboxes could not be wired on every rank")
                log("falling back to the per-step RCCL all-reduce on all ranks")
                want = "rccl"
        if want == "rccl":
            ok = True
            try:
                eng = kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K_local, device=local_rank, rank=rank, world=world,
                                   K_global=K_global, exchange=capi.KR_EXCHANGE_RCCL)
                ids = [kr.HipEngine.rccl_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(ids, src=0)
                eng.rccl_init(ids[0])
            except Exception as exc:  # noqa: BLE001
                log("rank %d: RCCL set-up failed: %r" % (rank, exc))
                ok = False
            if not all_ok(ok):
                # no silent degradation to independent replicas: a scaling harness must not read that as a scaling point
                raise SystemExit("bench.py: neither peer mailboxes nor the RCCL exchange could be set up on every rank")
            exchange_used = "ncclAllReduce (RCCL) of %d doubles per time step between per-step kernels" % L
    if eng is None:
        eng = kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K_local, device=local_rank, K_global=K_global)
    func = {"sm": 0, "ss": 1, "re": 2}[p["functional"]]
    dev = torch.device("cuda", local_rank)

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
        return a_, b_, wl.chi_and_JT(p["functional"], e.final_states(), p["tgt"], K_global)[1]

    eps_a, eps_b, J0 = prime(eng)
    if world > 1 and "mailboxes" in exchange_used:
        # Trial sweep: the in-kernel mailboxes have never run across real xGMI links in the build environment.  If ANY rank's
        # bounded spin gives up (KR_ERR_TIMEOUT) all ranks move to the RCCL carrier together -- a slower but plain path.
        ok = True
        try:
            dist.barrier()
            eng.iteration_functional_dev(func, eps_a.data_ptr(), eps_b.data_ptr())
            eng.synchronize()
        except Exception as exc:  # noqa: BLE001
            log("rank %d: trial sweep through the peer mailboxes failed: %r" % (rank, exc))
            ok = False
        if all_ok(ok):
            eps_a, eps_b, J0 = prime(eng)      # restart from the guess: the trial iteration is not part of the run
        elif args.exchange == "mailbox":
            raise SystemExit("bench.py: --exchange mailbox requested but the trial sweep through the peer mailboxes failed")
        else:
            log("falling back to the per-step RCCL all-reduce on all ranks")
            eng.close()
            eng = kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K_local, device=local_rank, rank=rank, world=world,
                               K_global=K_global, exchange=capi.KR_EXCHANGE_RCCL)
            ids = [kr.HipEngine.rccl_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            eng.rccl_init(ids[0])
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
    J1 = wl.chi_and_JT(p["functional"], eng.final_states(), p["tgt"], K_global)[1]
    info = eng.info()

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        units = 2.0 * K_local * world * NT     # one backward + one forward state-step per trajectory per interval
        value = units * args.steps / elapsed
        # dominant kernel: the update sweep.  Algorithmic flops per (trajectory, interval) unit of that
        # kernel with a precomputed step propagator: forward apply 8N^2 + L mu-matvecs 8N^2 L
        # + L dots 8N L.  Propagator construction / polynomial order is overhead, not counted.
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
            for name in ("r02_traffic_all.json", "r01_traffic.json"):
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
        kernel_name = "kr_mf_update" if mfma_sweep else ("kr_rowsplit_sweep" if info.get("rowsplit_R", 0) else "kr_sweep_update")
        line = {
            "metric": "Krotov iters/sec + state*timesteps/sec, N=%d nt=%d K=%d traj" % (N, NT, K_local * world),
            "value": value,
            "unit": "state*timesteps/s",
            "iters_per_sec": args.steps / elapsed,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%s (%s): N=%d %s generator + %d control operator(s), N_T=%d, K=%d trajectories per GPU "
                                   "(K_global=%d), J_T_%s, lambda_a=%g; one Krotov iteration per step"
                                   % (args.config, p["name"], N,
                                      "dense Hermitian" if np.allclose(p["H0"], p["H0"].conj().T) else "dense non-Hermitian (Liouvillian)",
                                      L, NT, K_local, K_global, p["functional"], p["lam"][0]),
                       "parallelism": parallelism, "exchange": exchange_used,
                       "engine": info},
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
                         "note": "fp64 matrix peak = vector peak on MI355X; achieved counts ALGORITHMIC flops (precomputed-"
                                 "propagator model, SURVEY 8d); the binding limit is sequential latency: N_T dependent "
                                 "steps, each = (1+L) products + one all-to-all sum over the workgroups + (m-1) Taylor "
                                 "rounds (m ~ 8)"},
            "J_T": {"before": float(J0), "after": float(J1)},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(p, log)
        print(json.dumps(line), flush=True)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
