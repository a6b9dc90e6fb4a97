"""GPU: the N > 1 path of the update sweep -- trajectories sharded over ranks, the per-step sum crossing
ranks through in-kernel peer mailboxes (IPC-mapped, system-scope atomics).  The box has ONE GPU, so the two
ranks share it (each persistent kernel limited to 64 workgroups so both are resident); this exercises the
cross-process mailbox protocol, the IPC wiring and the sharded functional normalisation exactly as the
8-GPU bench uses them.  Result must equal the single-handle, unsharded sweep to 1e-12 (the sum order over
ranks differs from the single-GPU slot order)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem(which):
    from krotov_jl_amd import workloads as wl
    if which == "rowsplit":   # N = 80 -> NP = 128, row-split kernels (several workgroups per trajectory)
        return wl.dense_problem(80, 8, 6, 1, 910, 0.05, 1.0), 6
    if which == "four_ranks":  # world = 4 on the one GPU
        return wl.dense_problem(12, 40, 22, 2, 912, 0.1, 1.0), 22
    if which == "c4_shard":    # the per-GPU shape of BASELINE config 4 on two ranks: N=64, L=2, 64 workgroups per rank (KT = 2)
        return wl.dense_problem(64, 200, 256, 2, 64001, 0.05, 0.05), 256
    if which == "sm_chain":
        return wl.dense_problem(12, 30, 10, 2, 914, 0.1, 0.5), 10
    if which == "skewed":      # as "wide"; rank 0 skips the exchange calibration, rank 1 runs it (different launch counts)
        return wl.dense_problem(20, 30, 256, 2, 913, 0.1, 1.0), 256
    if which == "wide":        # 64 workgroups per rank: calibrated exchange schedule + Hermitian-controls kernel variant
        return wl.dense_problem(20, 30, 256, 2, 913, 0.1, 1.0), 256
    if which == "xcd_local":  # few workgroups per rank, all placed on ONE XCD (exchange in its L2) AND a second rank behind the xmail hop:
        return wl.dense_problem(16, 60, 24, 2, 915, 0.1, 1.0), 24   # the shape of a 32-trajectory shard on each GPU of an 8-GPU run
    if which == "tile_sets":  # N = 64, 600 trajectories per rank on 64 workgroups: four tile sets per workgroup (round 3), partial sums of all
        return wl.dense_problem(64, 8, 1200, 2, 916, 0.05, 0.05), 1200   # sets accumulated in front of the one exchange that crosses the ranks
    if which == "turns":      # N = 280 -> NP = 512: 32 workgroups per trajectory, 2 groups per rank (64 workgroups), 4 trajectories per rank in
        return wl.dense_problem(280, 8, 8, 2, 917, 0.02, 1.0), 8        # two turns (kr_rowsplit_turns_update), known-pulse sweeps in batches
    if which == "pair_sets":  # N = 256, one control, 40 trajectories per rank on 8 groups of 8 workgroups: pair mode with 3 sets per group
        return wl.dense_problem(256, 6, 80, 1, 918, 0.02, 1.0), 80
    if which == "turns256":   # N = 200, two controls, 40 trajectories per rank: 8 groups of 8 workgroups, 5 turns each
        return wl.dense_problem(200, 6, 80, 2, 919, 0.02, 1.0), 80
    if which == "odd_nt":     # odd N_T: consecutive launches must keep alternating the mailbox buffers
        return wl.dense_problem(16, 61, 24, 2, 911, 0.1, 1.0), 24
    return wl.dense_problem(16, 60, 24, 2, 909, 0.1, 1.0), 24


def _worker(rank, world, port, q, which="small"):
    sys.path.insert(0, ROOT)
    max_wg = 48 if world > 2 else 64   # the ranks share ONE GPU: each takes a share of its CUs (kr_config.max_workgroups)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if which == "xcd_local":
        os.environ["KROTOV_HIP_XCD_LOCAL"] = "2"    # force the one-XCD placement although the ranks share the GPU
    if which == "skewed" and rank == 0:
        # kr_create's calibration runs a timing-dependent number of update launches; they must not advance the tag
        # sequence the ranks share (round-1 advisor finding): here rank 0 runs none at all, rank 1 runs 24-30
        os.environ["KROTOV_HIP_POLL_DELAY"] = "8"
    import torch.distributed as dist
    import krotov_jl_amd as kr
    from krotov_jl_amd import workloads as wl
    from krotov_jl_amd.distributed import shard_range, connect_peers
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        p, K = _problem(which)
        a, b = shard_range(K, rank, world)
        eng = kr.HipEngine(p["H0"], p["Hc"], p["tlist"], b - a, device=0, rank=rank, world=world, K_global=K, max_workgroups=max_wg)
        if which == "xcd_local":
            assert "one-XCD placement 1" in eng.plan(), eng.plan()
        if which == "turns":
            assert "kr_rowsplit_turns_update" in eng.plan() and "sets per workgroup 2" in eng.plan(), eng.plan()
        if which == "turns256":
            assert "kr_rowsplit_turns_update" in eng.plan() and "sets per workgroup 5" in eng.plan(), eng.plan()
        if which == "pair_sets":
            assert "kr_rowsplit_pair_sweep" in eng.plan() and "64 workgroups" in eng.plan(), eng.plan()
        connect_peers(eng, dist)
        eng.set_initial_states(p["psi0"][a:b])
        psiT = eng.forward(p["eps"])
        eps = p["eps"].copy()
        for it in range(1 if which == "c4_shard" else 4):
            _, _, chiT = wl.chi_and_JT("ss", psiT, p["tgt"][a:b], K_global=K)
            if it == 0:
                dist.barrier()   # later iterations run unsynchronised: ranks may be a launch apart
            eps, ga, psiT = eng.iteration(eps, chiT, p["S"], p["lam"])
        if which == "sm_chain":   # device-resident chain with J_T_sm: the sum of the taus crosses the ranks through the mailboxes
            import torch
            eng.set_target_states(p["tgt"][a:b])
            eng.set_update_params(p["S"], p["lam"])
            e0 = torch.from_numpy(p["eps"].copy()).cuda()
            e1 = torch.empty_like(e0)
            torch.cuda.synchronize()
            eng.forward_dev(e0.data_ptr())
            dist.barrier()
            for _ in range(3):
                eng.iteration_functional_dev(0, e0.data_ptr(), e1.data_ptr())
                e0, e1 = e1, e0
            eng.synchronize()
            eps, ga, psiT = e0.cpu().numpy(), eng.g_a_int(), eng.final_states()
        q.put((rank, eps, ga, psiT))
        dist.barrier()
        eng.close()
        dist.destroy_process_group()
    except Exception as exc:  # report instead of hanging the parent
        q.put((rank, "ERROR: %r" % (exc,), None, None))
        raise


def _run_ranks(which, world):
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    import krotov_jl_amd as kr
    from krotov_jl_amd import workloads as wl
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, which)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = {}
    for _ in range(world):
        rank, eps, ga, psiT = q.get(timeout=180)
        assert not isinstance(eps, str), eps
        got[rank] = (eps, ga, psiT)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    # both ranks hold the identical updated pulse, bit for bit
    for r in range(1, world):
        assert np.array_equal(got[0][0], got[r][0])
        assert np.array_equal(got[0][1], got[r][1])
    return got


@pytest.mark.parametrize("which", ["small", "odd_nt", "rowsplit", "wide", "skewed", "four_ranks", "xcd_local", "tile_sets", "turns", "pair_sets", "turns256"])
def test_two_ranks_one_gpu_equals_unsharded(which):
    world = 4 if which == "four_ranks" else 2
    sys.path.insert(0, ROOT)
    import krotov_jl_amd as kr
    from krotov_jl_amd import workloads as wl
    got = _run_ranks(which, world)
    # unsharded reference on one handle
    p, K = _problem(which)
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K) as eng:
        eng.set_initial_states(p["psi0"])
        psiT = eng.forward(p["eps"])
        eps = p["eps"].copy()
        for _ in range(4):
            _, _, chiT = wl.chi_and_JT("ss", psiT, p["tgt"])
            eps, ga, psiT = eng.iteration(eps, chiT, p["S"], p["lam"])
    assert np.max(np.abs(got[0][0] - eps)) / np.max(np.abs(eps)) < 1e-12
    assert np.max(np.abs(np.concatenate([got[r][2] for r in range(world)]) - psiT)) < 1e-11
    assert np.max(np.abs(got[0][1] - ga)) / np.max(np.abs(ga)) < 1e-10


def test_c4_shard_shape_two_ranks_vs_unsharded_oracle(oracle_c):
    """BASELINE config 4's shard shape (N=64, L=2, 64 workgroups per rank, two trajectories per team) on two ranks over
    N_T=200 steps: the sharded update sweep (partial sums crossing ranks through the IPC mailboxes) against the UNSHARDED
    C oracle (/root/reference/src/optimize.jl:328-370), not only against the unsharded HIP run."""
    sys.path.insert(0, ROOT)
    from krotov_jl_amd import workloads as wl
    got = _run_ranks("c4_shard", 2)
    p, K = _problem("c4_shard")
    nth = max(1, min(16, os.cpu_count() or 1))
    psiT_o, _ = oracle_c.forward(p["H0"], p["Hc"], p["tlist"], p["eps"], p["psi0"], store=False, mode=1, nthreads=nth)
    _, _, chiT = wl.chi_and_JT("ss", psiT_o, p["tgt"])
    eps_o, ga_o, psiT1_o, _, _ = oracle_c.iteration(p["H0"], p["Hc"], p["tlist"], p["eps"], p["S"], p["lam"], p["psi0"], chiT,
                                                    mode=1, nthreads=nth)
    assert np.max(np.abs(eps_o - p["eps"])) > 1e-4
    assert np.max(np.abs(got[0][0] - eps_o)) / np.max(np.abs(eps_o)) < 1e-10
    assert np.max(np.abs(np.concatenate([got[r][2] for r in range(2)]) - psiT1_o)) < 1e-9
    assert np.max(np.abs(got[0][1] - ga_o)) / np.max(np.abs(ga_o)) < 1e-9


def test_device_chain_j_t_sm_two_ranks():
    """kr_iteration_functional_dev with J_T_sm on two ranks (round 1 refused it for world > 1): every rank needs the sum of
    ALL taus; it travels through the tau slots of the peer mailboxes.  Against the host-chi chain on one unsharded handle."""
    sys.path.insert(0, ROOT)
    import krotov_jl_amd as kr
    from krotov_jl_amd import workloads as wl
    got = _run_ranks("sm_chain", 2)
    p, K = _problem("sm_chain")
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K) as eng:
        eng.set_initial_states(p["psi0"])
        psiT = eng.forward(p["eps"])
        eps = p["eps"].copy()
        for _ in range(3):
            _, _, chiT = wl.chi_and_JT("sm", psiT, p["tgt"])
            eps, ga, psiT = eng.iteration(eps, chiT, p["S"], p["lam"])
    assert np.max(np.abs(got[0][0] - eps)) / np.max(np.abs(eps)) < 1e-10
    assert np.max(np.abs(np.concatenate([got[r][2] for r in range(2)]) - psiT)) < 1e-10
