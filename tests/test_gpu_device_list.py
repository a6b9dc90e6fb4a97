"""GPU: ONE handle driving several GPUs from one process (kr_config.n_devices / devices) -- what a Julia host calls
(/root/reference/src/optimize.jl:182,303,321,360 thread over trajectories inside one process).  The box has one GPU, so
the device list names it several times: the shards then share its CUs (each capped to its share automatically, no
KROTOV_HIP_MAX_WG), are wired through plain peer pointers instead of IPC, and run from library-owned threads.  Results must
equal the unsharded oracle; the device-resident functional chain must work for J_T_sm too (the sum of the taus crosses the
shards through the mailboxes)."""
import numpy as np
import pytest

import krotov_jl_amd as kr
from krotov_jl_amd import workloads as wl
from krotov_jl_amd import _capi as capi

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.mark.parametrize("shape", ["small", "c4_shard", "rowsplit", "three", "uneven_generators"])
def test_one_handle_device_list_vs_oracle(oracle_c, shape):
    tg = None
    if shape == "small":
        p, devs = wl.dense_problem(16, 40, 24, 2, 501, 0.1, 1.0), [0, 0]
    elif shape == "c4_shard":      # N=64, L=2, 128 trajectories per shard: matrix-core sweep, 64 workgroups per shard
        p, devs = wl.dense_problem(64, 60, 256, 2, 64001, 0.05, 0.05), [0, 0]
    elif shape == "rowsplit":      # N=80: several workgroups per trajectory
        p, devs = wl.dense_problem(80, 8, 6, 1, 502, 0.05, 1.0), [0, 0]
    elif shape == "three":
        p, devs = wl.dense_problem(12, 30, 22, 2, 503, 0.1, 1.0), [0, 0, 0]
    else:
        p, devs = wl.dense_problem(10, 20, 9, 2, 504, 0.1, 1.0), [0, 0]
    K = p["psi0"].shape[0]
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K, devices=devs) as eng:
        info = eng.info()
        assert info["workgroups"] <= 256 // len(devs)
        eng.set_initial_states(p["psi0"])
        psiT = eng.forward(p["eps"])
        psiT_o, _ = oracle_c.forward(p["H0"], p["Hc"], p["tlist"], p["eps"], p["psi0"], store=False, mode=1, nthreads=8)
        assert np.max(np.abs(psiT - psiT_o)) < 1e-11
        _, _, chiT = wl.chi_and_JT("ss", psiT_o, p["tgt"])
        eps_new, ga, psiT1 = eng.iteration(p["eps"], chiT, p["S"], p["lam"])
        eps_o, ga_o, psiT1_o, X_o, Phi_o = oracle_c.iteration(p["H0"], p["Hc"], p["tlist"], p["eps"], p["S"], p["lam"], p["psi0"], chiT,
                                                             mode=1, nthreads=8)
        assert _rel(eps_new, eps_o) < 1e-10 and _rel(ga, ga_o) < 1e-9 and np.max(np.abs(psiT1 - psiT1_o)) < 1e-9
        NT = p["eps"].shape[1]
        for k in (0, K // 2, K - 1):      # storages are routed to the shard that owns the trajectory
            assert np.max(np.abs(eng.read_storage("bw", k, NT // 2) - X_o[NT // 2, k])) < 1e-11
            assert np.max(np.abs(eng.read_storage("fw", k, NT) - psiT1[k])) == 0.0
        again = eng.iteration(p["eps"], chiT, p["S"], p["lam"])
        assert np.array_equal(again[0], eps_new) and np.array_equal(again[2], psiT1)
        eng.set_target_states(p["tgt"])
        assert np.max(np.abs(eng.taus() - wl.chi_and_JT("ss", psiT1, p["tgt"])[0])) < 1e-11


@pytest.mark.parametrize("fname,fid", [("sm", 0), ("ss", 1), ("re", 2)])
def test_device_functional_chain_on_a_device_list(fname, fid):
    """kr_iteration_functional_dev on a two-shard handle against the host-chi chain on one ordinary handle; J_T_sm needs
    the sum of ALL taus on every shard."""
    import torch
    p = wl.dense_problem(12, 30, 10, 2, 511 + fid, 0.1, 0.5)
    K = 10
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K) as a, kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K, devices=[0, 0]) as b:
        for e in (a, b):
            e.set_initial_states(p["psi0"])
            e.set_target_states(p["tgt"])
        eps = p["eps"].copy()
        psiT = a.forward(eps)
        for _ in range(3):
            _, _, chiT = wl.chi_and_JT(fname, psiT, p["tgt"])
            eps, ga, psiT = a.iteration(eps, chiT, p["S"], p["lam"])
        b.set_update_params(p["S"], p["lam"])
        e0 = torch.from_numpy(p["eps"].copy()).cuda()
        e1 = torch.empty_like(e0)
        torch.cuda.synchronize()
        b.forward_dev(e0.data_ptr())
        for _ in range(3):
            b.iteration_functional_dev(fid, e0.data_ptr(), e1.data_ptr())
            e0, e1 = e1, e0
        b.synchronize()
        assert _rel(e0.cpu().numpy(), eps) < 1e-10
        assert np.max(np.abs(b.final_states() - psiT)) < 1e-10
        assert _rel(b.g_a_int(), ga) < 1e-9


def test_device_list_errors():
    p = wl.dense_problem(8, 5, 6, 1, 520, 0.1, 1.0)
    with pytest.raises(kr.KrotovHipError) as ei:    # RCCL refuses one device twice
        kr.HipEngine(p["H0"], p["Hc"], p["tlist"], 6, devices=[0, 0], exchange=capi.KR_EXCHANGE_RCCL)
    assert ei.value.code == capi.KR_ERR_UNSUPPORTED
    with pytest.raises(kr.KrotovHipError):
        kr.HipEngine(p["H0"], p["Hc"], p["tlist"], 6, devices=[0, 99])
    with pytest.raises(kr.KrotovHipError) as ei:    # repeated ordinals are a test facility, bounded by the hardware queues
        kr.HipEngine(p["H0"], p["Hc"], p["tlist"], 6, devices=[0, 0, 0, 0])
    assert ei.value.code == capi.KR_ERR_UNSUPPORTED
    with pytest.raises(kr.KrotovHipError):
        kr.HipEngine(p["H0"], p["Hc"], p["tlist"], 6, devices=[0, 0], rank=1, world=2)
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], 6, devices=[0, 0]) as eng:
        with pytest.raises(kr.KrotovHipError):
            eng.ipc_export()                        # a device list wires itself
        with pytest.raises(kr.KrotovHipError):
            eng.read_storage("bw", 6, 0)


def test_max_workgroups_is_honoured(oracle_c):
    """kr_config.max_workgroups: a handle that shares its GPU with others declares its share of the CUs; the plan packs
    more trajectories per workgroup (or moves to the generic kernels) instead of launching more persistent workgroups."""
    p = wl.dense_problem(16, 20, 40, 1, 530, 0.1, 1.0)
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], 40, max_workgroups=8) as eng:
        info = eng.info()
        assert info["workgroups"] <= 8 and info["generic_update"] == 0
        eng.set_initial_states(p["psi0"])
        psiT = eng.forward(p["eps"])
        _, _, chiT = wl.chi_and_JT("ss", psiT, p["tgt"])
        eps_new, ga, psiT1 = eng.iteration(p["eps"], chiT, p["S"], p["lam"])
    eps_o = oracle_c.iteration(p["H0"], p["Hc"], p["tlist"], p["eps"], p["S"], p["lam"], p["psi0"], chiT)[0]
    assert _rel(eps_new, eps_o) < 1e-10
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], 40, max_workgroups=2) as eng:   # 40 trajectories on 2 workgroups: generic kernels
        assert eng.info()["generic_update"] == 1
        eng.set_initial_states(p["psi0"])
        eng.forward(p["eps"])
        assert _rel(eng.iteration(p["eps"], chiT, p["S"], p["lam"])[0], eps_o) < 1e-10


def test_a_late_shard_times_out_with_a_report_and_the_handle_recovers(oracle_c):
    """Round 4: the shards of a device list wait for each other inside their kernels, bounded by kr_config.spin_timeout_ms.  A shard
    that starts 400 ms late (KROTOV_HIP_SHARD_SKEW_MS, a test facility) against a bound of 100 ms: KR_ERR_TIMEOUT, and the message
    says who was late -- every shard's host timeline, the tags its mailbox between GPUs holds against the tag it expected (the early
    shard holds its own partial sums and not the peer's), and how many partial sums each replica of the local mailbox holds (all of
    them: the loss was between the GPUs).  The same problem on a fresh handle without the skew is exact afterwards."""
    p = wl.dense_problem(12, 6, 20, 2, 541, 0.1, 1.0)
    K = 20
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K, devices=[0, 0], spin_timeout_ms=100, options={"SHARD_SKEW_MS": 400}) as eng:
        eng.set_initial_states(p["psi0"])
        psiT = eng.forward(p["eps"])                  # known-pulse sweeps do not wait for anybody
        _, _, chiT = wl.chi_and_JT("ss", psiT, p["tgt"])
        with pytest.raises(kr.KrotovHipError) as ei:
            eng.iteration(p["eps"], chiT, p["S"], p["lam"])
        msg = str(ei.value)
        assert ei.value.code == capi.KR_ERR_TIMEOUT
        assert "shard 0 of 2" in msg and "gave up at step 0" in msg and "timelines [ms]" in msg and "shard 1 launch" in msg
        # (round 5: the tags of a group start at a base of its own, so that a recycled mailbox never sees a tag it waits for)
        import re
        m = re.search(r"expected tag (\d+) in buffer (\d)", msg)
        assert m, msg
        tag, buf = m.group(1), m.group(2)
        assert int(buf) == int(tag) & 1
        assert "[buf %s rank 0: %s %s %s %s] [buf %s rank 1: 0 0 0 0]" % (buf, tag, tag, tag, tag, buf) in msg, msg
        assert "partial sums of that step in the mailbox replicas (slot stride 2 words): " + " ".join([str(eng.info()["workgroups"])] * 8) in msg, msg
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K, devices=[0, 0]) as eng:
        eng.set_initial_states(p["psi0"])
        eng.forward(p["eps"])
        eps_new = eng.iteration(p["eps"], chiT, p["S"], p["lam"])[0]
    eps_o = oracle_c.iteration(p["H0"], p["Hc"], p["tlist"], p["eps"], p["S"], p["lam"], p["psi0"], chiT)[0]
    assert _rel(eps_new, eps_o) < 1e-10
