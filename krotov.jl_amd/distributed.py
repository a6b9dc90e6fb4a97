"""Multi-GPU wiring (one process per GPU, trajectories sharded).

The only cross-GPU quantity of the hot path is the per-step sum of the L update integrands
(/root/reference/src/optimize.jl:339-349).  The update kernels exchange it in-kernel through
peer mailboxes over xGMI; torch.distributed (RCCL / gloo) is used here only to hand the mailbox
IPC handles around and for the host-side gathers of taus / final states.
"""
import numpy as np


def shard_range(K_global, rank, world):
    """Block partition of the trajectories (the loops the reference marks @threadsif,
    optimize.jl:182,303,321,360).  Returns (start, stop)."""
    base, rem = divmod(K_global, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def connect_peers(engine, dist):
    """All-gather every rank's mailbox IPC handle and open the peers' mailboxes."""
    mine = engine.ipc_export()
    handles = [None] * dist.get_world_size()
    dist.all_gather_object(handles, mine)
    engine.ipc_connect(handles)
    dist.barrier()


def allgather_states(local, dist, K_global):
    """Gather (K_local, N) blocks from all ranks into (K_global, N) on every rank (host side:
    chi/J_T coupling of J_T_sm, optimize.jl:297-302)."""
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, np.ascontiguousarray(local))
    out = np.concatenate(parts, axis=0)
    assert out.shape[0] == K_global
    return out
