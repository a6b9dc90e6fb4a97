"""CPU, world_size 2 over gloo: the trajectory sharding used for N > 1 GPUs.  Each rank carries a block
of the trajectories; the only cross-rank quantity of the update sweep is the per-step sum of the L
integrands.  Here the oracle plays the engine and the all-reduce is done by torch.distributed, step by
step, and the result must equal the unsharded sweep."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    import krotov_oracle as o
    from krotov_jl_amd import workloads as wl
    from krotov_jl_amd.distributed import shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = wl.dense_problem(6, 20, 5, 2, 77, 0.1, 2.0)
    K = p["psi0"].shape[0]
    a, b = shard_range(K, rank, world)
    psi0, tgt = p["psi0"][a:b], p["tgt"][a:b]
    H0, Hc, tl, eps, S, lam = p["H0"], p["Hc"], p["tlist"], p["eps"], p["S"], p["lam"]
    psiT, _ = o.forward(H0, Hc, tl, eps, psi0)
    _, _, chiT = wl.chi_and_JT("ss", psiT, tgt, K_global=K)   # local chi with the GLOBAL normalisation
    X = o.backward(H0, Hc, tl, eps, chiT)
    L, NT = eps.shape
    psi = psi0.copy()
    eps_new = eps.copy()
    for n in range(NT):
        du = np.array([sum(np.vdot(X[n, k], Hc[l] @ psi[k]).imag for k in range(b - a)) for l in range(L)])
        t = torch.from_numpy(du)
        dist.all_reduce(t)                                     # the one collective of the path
        for l in range(L):
            eps_new[l, n] = eps[l, n] + S[l, n] / lam[l] * t[l].item()
        U = o.step_propagator(H0, Hc, eps_new[:, n], tl[n + 1] - tl[n])
        psi = psi @ U.T
    if rank == 0:
        q.put(eps_new)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_update_equals_unsharded():
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import krotov_oracle as o
    from krotov_jl_amd import workloads as wl
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    eps_sharded = q.get(timeout=120)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    p = wl.dense_problem(6, 20, 5, 2, 77, 0.1, 2.0)
    psiT, _ = o.forward(p["H0"], p["Hc"], p["tlist"], p["eps"], p["psi0"])
    _, _, chiT = wl.chi_and_JT("ss", psiT, p["tgt"])
    eps_ref = o.iteration(p["H0"], p["Hc"], p["tlist"], p["eps"], p["S"], p["lam"], p["psi0"], chiT)[0]
    assert np.max(np.abs(eps_sharded - eps_ref)) < 1e-13


def test_shard_range_partition():
    from krotov_jl_amd.distributed import shard_range
    for K in (1, 5, 8, 256, 1000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(K, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == K
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
