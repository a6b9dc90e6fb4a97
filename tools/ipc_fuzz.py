"""Developer tool: the N > 1 path as bench.py --gpus N uses it -- one process per rank, mailboxes mapped through IPC -- on random
problems against the C oracle, MANY handles per process in turn (tests/test_gpu_multirank.py creates one handle per process: the pooled
mailboxes that are exported again and the cached IPC mappings of round 4 only show with several).  Two ranks share the one GPU of the
box (kr_config.max_workgroups = 64 each).  Every rank walks the same random shapes: forward sweep + two chained iterations on its
shard; rank 0 gathers both ranks' pulses (they must be identical bit for bit) and compares with the oracle on the whole ensemble.
   python tools/ipc_fuzz.py [COUNT=60] [SEED=1]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, count, seed, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import krotov_jl_amd as kr
    from krotov_jl_amd import workloads as wl
    from krotov_jl_amd.distributed import shard_range, connect_peers
    import oracle_c as oc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(seed)
    bad = 0
    for it in range(count):
        N = int(rng.choice([2, 5, 9, 16, 24, 33, 48, 64, 80, 130]))
        kmax = 256 if N <= 64 else 24
        K = int(rng.integers(2, kmax + 1))
        L = int(rng.choice([1, 2, 2, 3]))
        NT = int(rng.choice([1, 2, 5, 9, 20]))
        p = wl.dense_problem(N, NT, K, L, 5000 + it, float(rng.choice([0.02, 0.05, 0.2])), 2.0)
        tag = ""
        if rng.random() < 0.3:
            p["H0"] = p["H0"] - 1j * 0.05 * np.diag(rng.uniform(0, 1, N))
            tag += " nonherm"
        if rng.random() < 0.3 and NT > 1:
            p["tlist"] = np.concatenate([[0.0], np.cumsum(np.diff(p["tlist"]) * rng.uniform(0.5, 1.5, NT))])
            tag += " grid"
        p["lam"] = rng.uniform(0.5, 8.0, L)
        a, b = shard_range(K, rank, world)
        t0 = time.time()
        try:
            with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], b - a, device=0, rank=rank, world=world, K_global=K, max_workgroups=64) as eng:
                connect_peers(eng, dist)
                eng.set_initial_states(p["psi0"][a:b])
                psiT = eng.forward(p["eps"])
                psiT_o, _ = oc.forward(p["H0"], p["Hc"], p["tlist"], p["eps"], p["psi0"], store=False)
                d0 = np.max(np.abs(psiT - psiT_o[a:b]))
                eps = eps_o = p["eps"]
                d1 = d2 = 0.0
                for _ in range(2):
                    _, _, chiT = wl.chi_and_JT("ss", psiT_o, p["tgt"])
                    guess = eps_o
                    eps, ga, psiT = eng.iteration(guess, chiT[a:b], p["S"], p["lam"])
                    eps_o, ga_o, psiT_o = oc.iteration(p["H0"], p["Hc"], p["tlist"], guess, p["S"], p["lam"], p["psi0"], chiT)[:3]
                    d1 = max(d1, np.max(np.abs(eps - eps_o)) / max(np.max(np.abs(eps_o)), 1e-300))
                    d2 = max(d2, np.max(np.abs(psiT - psiT_o[a:b])))
                info = eng.info()
                dist.barrier()   # nobody destroys its mailbox while a peer's kernel may still post into it
            both = [None] * world
            dist.all_gather_object(both, eps)
            same = all(np.array_equal(both[0], e) for e in both)
            ok = same and d0 < 1e-11 and d1 < 1e-10 and d2 < 1e-10
            if rank == 0:
                print("%-4s N=%4d K=%4d L=%d NT=%2d NP=%4d R=%3d wg=%3d  dpsi %.1e deps %.1e dpsi' %.1e pulses identical on the ranks: %s  %.1fs%s" % (
                    "ok" if ok else "BAD", N, K, L, NT, info["NP"], info["rowsplit_R"], info["workgroups"], d0, d1, d2, same, time.time() - t0, tag), flush=True)
            bad += 0 if ok else 1
        except Exception as exc:  # noqa: BLE001
            bad += 1
            print("FAIL rank %d N=%4d K=%4d L=%d NT=%2d%s: %s" % (rank, N, K, L, NT, tag, str(exc)[:1500]), flush=True)
            break   # the peer is waiting in a collective: stop both
    if rank == 0:
        print("ipc_fuzz: %d problem(s) in %d shapes" % (bad, count), flush=True)
    q.put((rank, bad))
    dist.destroy_process_group()


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 1000)
    procs = [ctx.Process(target=worker, args=(r, 2, port, count, seed, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    bad = 0
    try:
        for _ in range(2):
            bad += q.get(timeout=900)[1]
    except Exception:  # noqa: BLE001
        bad += 1
        for pr in procs:
            if pr.is_alive():
                pr.terminate()
    for pr in procs:
        pr.join(timeout=60)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
