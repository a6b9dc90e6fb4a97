"""GPU tests of round 5: the integrity guards of the N <= 64 kernels (pulse coherence words, mailbox self-test), the teardown of a
device-list handle after a failed asynchronous call, the known-pulse sweeps in blocks of time."""
import numpy as np
import pytest

import krotov_jl_amd as kr
from krotov_jl_amd import workloads as wl
from krotov_jl_amd import _capi as capi

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.mark.parametrize("shape", [(64, 12, 256, 2), (64, 10, 600, 2), (16, 20, 32, 1), (8, 9, 700, 3), (100, 6, 300, 3)],
                         ids=["c3_vector_herm", "matrix_cores", "c2_one_xcd", "kt4_lmax4", "streamed"])
def test_pulse_coherence_words_find_a_planted_fault(shape, oracle_c):
    """optimize.jl:351-358 has ONE pulse value per step.  In the persistent kernels every workgroup forms it for itself from the partial
    sums IT gathered; each folds its N_T x L values into a 64-bit word and kr_pulse_check compares the P words after the sweep (the
    headline path had no such check until round 5).  Clean: the oracle's pulses.  PULSE_FAULT=1: workgroup 1 folds one value with a
    flipped bit -> KR_ERR_INTERNAL naming the workgroup instead of a silently diverged sweep; the handle works again afterwards."""
    N, NT, K, L = shape
    p = wl.dense_problem(N, NT, K, L, 5101, 0.05, 2.0)
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K) as eng:
        info = eng.info()
        assert info["generic_update"] == 0 and info["rowsplit_R"] == 0 and info["workgroups"] >= 2, eng.plan()
        assert "per-step totals of all workgroups compared on" in eng.plan() and "mailbox self-test passed" in eng.plan(), eng.plan()
        eng.set_initial_states(p["psi0"])
        psiT = eng.forward(p["eps"])
        _, _, chiT = wl.chi_and_JT(p["functional"], psiT, p["tgt"])
        eps_new = eng.iteration(p["eps"], chiT, p["S"], p["lam"])[0]
    eps_o = oracle_c.iteration(p["H0"], p["Hc"], p["tlist"], p["eps"], p["S"], p["lam"], p["psi0"], chiT, mode=1, nthreads=8)[0]
    assert _rel(eps_new, eps_o) < 1e-10
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K, options={"PULSE_FAULT": 1}) as eng:
        eng.set_initial_states(p["psi0"])
        eng.forward(p["eps"])
        with pytest.raises(kr.KrotovHipError) as ei:
            eng.iteration(p["eps"], chiT, p["S"], p["lam"])
        assert ei.value.code == capi.KR_ERR_INTERNAL and "gathered other totals than the reference workgroup" in str(ei.value), str(ei.value)
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K, options={"PULSE_CHECK": 0, "PULSE_FAULT": 1}) as eng:   # check off: nothing posted, nothing found
        assert "per-step totals of all workgroups compared n/a" in eng.plan()
        eng.set_initial_states(p["psi0"])
        eng.forward(p["eps"])
        assert np.array_equal(eng.iteration(p["eps"], chiT, p["S"], p["lam"])[0], eps_new)


def test_mailbox_self_test_sets_a_bad_block_aside_or_fails_kr_create(oracle_c):
    """kr_create runs the exchange's own store / poll pattern on every freshly allocated mailbox (kr_mail_selftest).  MAIL_FAULT=1: the
    first block 'loses' a word -> it is set aside, the second block passes, the plan says so and the results are the oracle's.
    MAIL_FAULT=2: every block fails -> KR_ERR_INTERNAL from kr_create after three attempts, no handle."""
    p = wl.dense_problem(32, 15, 150, 2, 5202, 0.05, 2.0)
    K = 150
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K, options={"MAIL_FAULT": 1}) as eng:
        assert "mailbox self-test passed after setting blocks aside" in eng.plan(), eng.plan()
        eng.set_initial_states(p["psi0"])
        psiT = eng.forward(p["eps"])
        _, _, chiT = wl.chi_and_JT("ss", psiT, p["tgt"])
        eps_new = eng.iteration(p["eps"], chiT, p["S"], p["lam"])[0]
    eps_o = oracle_c.iteration(p["H0"], p["Hc"], p["tlist"], p["eps"], p["S"], p["lam"], p["psi0"], chiT, mode=1, nthreads=8)[0]
    assert _rel(eps_new, eps_o) < 1e-10
    with pytest.raises(kr.KrotovHipError) as ei:
        kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K, options={"MAIL_FAULT": 2})
    assert ei.value.code == capi.KR_ERR_INTERNAL and "failed their self-test" in str(ei.value), str(ei.value)
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K, options={"MAIL_SELFTEST": 0}) as eng:
        assert "mailbox self-test off" in eng.plan()
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K, options={"UC_POOL": 0}) as eng:
        assert "WARNING: UC_POOL=0 (diagnostics only)" in eng.plan()


def test_destroy_right_after_a_failed_asynchronous_call_on_a_device_list(oracle_c):
    """Round 5 (the teardown path behind the GPU fault on record): the chain calls are asynchronous, so a device-list handle
    can be destroyed while its shards' persistent kernels are still waiting for each other.  Shard 1 starts 300 ms late against a spin
    bound of 100 ms: shard 0's kernel gives up alone, shard 1's runs -- and stores into shard 0's mailbox -- long after.  kr_destroy
    must wait for EVERY shard's stream before it releases ANY shard's memory; a fresh handle of the same shape is exact afterwards."""
    import torch
    p = wl.dense_problem(12, 6, 20, 2, 5303, 0.1, 1.0)
    K = 20
    eng = kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K, devices=[0, 0], spin_timeout_ms=100, options={"SHARD_SKEW_MS": 300})
    eng.set_initial_states(p["psi0"])
    eng.set_target_states(p["tgt"])
    eng.set_update_params(p["S"], p["lam"])
    e0 = torch.from_numpy(p["eps"].copy()).cuda()
    e1 = torch.empty_like(e0)
    torch.cuda.synchronize()
    eng.forward_dev(e0.data_ptr())
    eng.iteration_functional_dev(1, e0.data_ptr(), e1.data_ptr())    # returns with the kernels in flight (and about to time out)
    eng.close()                                                       # ... and the handle goes at once
    torch.cuda.synchronize()
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K, devices=[0, 0]) as eng:
        eng.set_initial_states(p["psi0"])
        psiT = eng.forward(p["eps"])
        _, _, chiT = wl.chi_and_JT("ss", psiT, p["tgt"])
        eps_new = eng.iteration(p["eps"], chiT, p["S"], p["lam"])[0]
    eps_o = oracle_c.iteration(p["H0"], p["Hc"], p["tlist"], p["eps"], p["S"], p["lam"], p["psi0"], chiT)[0]
    assert _rel(eps_new, eps_o) < 1e-10
    # a synchronous call that fails returns with every shard idle, and the handle itself recovers
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K, devices=[0, 0], spin_timeout_ms=100, options={"SHARD_SKEW_MS": 300}) as eng:
        eng.set_initial_states(p["psi0"])
        eng.forward(p["eps"])
        with pytest.raises(kr.KrotovHipError) as ei:
            eng.iteration(p["eps"], chiT, p["S"], p["lam"])
        assert ei.value.code == capi.KR_ERR_TIMEOUT


def test_rccl_handles_in_turn_share_one_communicator():
    """Handles with exchange = RCCL created and destroyed in turn used to create and destroy a communicator each (RCCL then frees its
    device buffers: the pattern behind lost stores on this platform).  The communicator now goes back to a process-wide cache and the
    next handle of the same shape takes it; results are identical from handle to handle."""
    p = wl.dense_problem(10, 8, 12, 2, 5404, 0.1, 1.0)
    outs = []
    for _ in range(3):
        with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], 12, exchange=capi.KR_EXCHANGE_RCCL) as eng:
            try:
                eng.rccl_init(eng.rccl_unique_id())
            except kr.KrotovHipError as e:
                if e.code == capi.KR_ERR_UNSUPPORTED:
                    pytest.skip("librccl not available: " + str(e))
                raise
            eng.set_initial_states(p["psi0"])
            psiT = eng.forward(p["eps"])
            _, _, chiT = wl.chi_and_JT("ss", psiT, p["tgt"])
            outs.append(eng.iteration(p["eps"], chiT, p["S"], p["lam"])[0])
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def _nonherm(p, rng, N):
    p["H0"] = p["H0"] - 1j * 0.05 * np.diag(rng.uniform(0, 1, N)) + 0.02 * rng.normal(size=(N, N))


@pytest.mark.parametrize("case", ["c3_shape", "ragged_nonherm_grid", "n40_k70", "two_generators", "matrix_cores_k600", "few_k3", "blocks_of_5", "blocks_of_8_chain", "blocks_of_4_tree", "blocks_of_32",
                                  "c2_shape", "small_two_generators_ragged", "n32_k300_three_controls", "n20_blocks_of_3", "n5_one_trajectory"])
def test_known_pulse_sweeps_in_blocks_of_time_vs_oracle(case, oracle_c):
    """Round 5, kr_device_tblock.h: step propagators -> block propagators (matrix cores) -> sequential sweep over the block boundaries ->
    all blocks in parallel.  Forward storage (optimize.jl:254,263: slot n + 1), backward storage through the TRUE adjoint (optimize.jl:
    307,315: slot n; workspace.jl:69) and the final states against the C oracle at the tolerances of the sequential sweeps (states
    1e-11), for: C3's shape on a shorter grid, a ragged last block with a dissipative drift on a non-uniform grid, N < 64 with a
    trajectory count that fills neither a wave nor a workgroup, per-trajectory generators, the shapes whose sequential kernels run on
    the matrix cores, three trajectories, other block lengths, the block propagators as a tree of pairwise products (default, blocks of
    16) or as a chain per block.  The sequential sweep (TBLOCK=0) must agree to rounding.
    N <= 32 (kr_device_prop.h: kr_tb_products_small, kr_sweep_apply<.., TB>): C2's shape, per-trajectory generators on a ragged non-uniform
    grid with a dissipative drift, several trajectories per workgroup, short blocks, a single trajectory."""
    rng = np.random.default_rng(5500)
    opts, tg, G = {}, None, 1
    if case == "c3_shape":
        N, NT, K, L = 64, 100, 256, 2
    elif case == "ragged_nonherm_grid":
        N, NT, K, L = 64, 77, 40, 2
    elif case == "n40_k70":
        N, NT, K, L = 40, 90, 70, 1
    elif case == "two_generators":
        N, NT, K, L = 48, 70, 37, 2
    elif case == "matrix_cores_k600":
        N, NT, K, L = 64, 66, 600, 2
    elif case == "few_k3":
        N, NT, K, L = 64, 130, 3, 1
    elif case == "blocks_of_5":
        N, NT, K, L = 33, 64, 20, 2
        opts["TBLOCK"] = 5
    elif case == "blocks_of_8_chain":      # (the default is a tree of pairwise products over blocks of 16 steps: here the chain of products per block)
        N, NT, K, L = 64, 67, 24, 1
        opts.update(TBLOCK=8, TBLOCK_TREE=0)
    elif case == "blocks_of_4_tree":
        N, NT, K, L = 50, 71, 130, 2
        opts["TBLOCK"] = 4
    elif case == "blocks_of_32":
        N, NT, K, L = 64, 150, 24, 1
        opts["TBLOCK"] = 32
    elif case == "c2_shape":
        N, NT, K, L = 16, 500, 32, 1
    elif case == "small_two_generators_ragged":
        N, NT, K, L = 7, 131, 19, 2
        opts["TBLOCK"] = 6
    elif case == "n32_k300_three_controls":
        N, NT, K, L = 32, 150, 300, 3
    elif case == "n20_blocks_of_3":
        N, NT, K, L = 20, 40, 9, 2
        opts.update(TBLOCK=3, TBLOCK_MIN_NT=16)
    else:
        N, NT, K, L = 5, 200, 1, 1
    p = wl.dense_problem(N, NT, K, L, 5501, 0.05, 2.0)
    if case in ("ragged_nonherm_grid", "blocks_of_5", "small_two_generators_ragged"):
        _nonherm(p, rng, N)
        p["tlist"] = np.concatenate([[0.0], np.cumsum(np.diff(p["tlist"]) * rng.uniform(0.5, 1.5, NT))])
    H0, Hc = p["H0"], p["Hc"]
    kw = {}
    if case in ("two_generators", "small_two_generators_ragged"):
        G = 2
        H0 = np.stack([p["H0"], 0.8 * p["H0"]])
        Hc = np.stack([p["Hc"], 1.2 * p["Hc"]])
        tg = rng.integers(0, 2, K).astype(np.int32)
        kw["traj_generator"] = tg
    outs = {}
    for name, o in (("blocks", dict(opts)), ("sequential", dict(opts, TBLOCK=0))):
        with kr.HipEngine(H0, Hc, p["tlist"], K, options=o, **kw) as eng:
            assert eng.info()["precomputed_propagators"] == 1, eng.plan()
            assert ("time blocks" in eng.plan()) == (name == "blocks") and ("on the matrix cores" in eng.plan()) == (name == "blocks" and N > 32), eng.plan()
            eng.set_initial_states(p["psi0"])
            psiT = eng.forward(p["eps"])
            fw = np.stack([np.stack([eng.read_storage("fw", k, n) for k in (0, K // 2, K - 1)]) for n in range(NT + 1)])
            chiT = wl.chi_and_JT("ss", psiT, p["tgt"])[2]
            eng.backward(p["eps"], chiT)
            bw = np.stack([np.stack([eng.read_storage("bw", k, n) for k in (0, K // 2, K - 1)]) for n in range(NT + 1)])
            outs[name] = (psiT, fw, bw)
            if name == "blocks":
                assert np.array_equal(fw[NT], psiT[[0, K // 2, K - 1]])     # the storage's last slot IS the final state (optimize.jl:263)
                assert np.array_equal(bw[NT], chiT[[0, K // 2, K - 1]]) and np.array_equal(fw[0], p["psi0"][[0, K // 2, K - 1]])
    # oracle, generator by generator
    psiT_o = np.empty((K, N), complex)
    Phi_o = np.empty((NT + 1, K, N), complex)
    X_o = np.empty((NT + 1, K, N), complex)
    chiT = wl.chi_and_JT("ss", outs["blocks"][0], p["tgt"])[2]
    for g in range(G):
        m = np.ones(K, bool) if tg is None else (tg == g)
        if not m.any():
            continue
        h0, hc = (H0, Hc) if G == 1 else (H0[g], Hc[g])
        f, st = oracle_c.forward(h0, hc, p["tlist"], p["eps"], p["psi0"][m], store=True, mode=1, nthreads=8)
        psiT_o[m], Phi_o[:, m] = f, st
        X_o[:, m] = oracle_c.backward(h0, hc, p["tlist"], p["eps"], chiT[m], mode=1, nthreads=8)
    ks = [0, K // 2, K - 1]
    for name in ("blocks", "sequential"):
        psiT, fw, bw = outs[name]
        assert np.max(np.abs(psiT - psiT_o)) < 1e-11, name
        assert np.max(np.abs(fw - Phi_o[:, ks])) < 1e-11, name
        assert np.max(np.abs(bw - X_o[:, ks])) < 1e-11, name
    assert np.max(np.abs(outs["blocks"][1] - outs["sequential"][1])) < 1e-12 and np.max(np.abs(outs["blocks"][2] - outs["sequential"][2])) < 1e-12


def test_blocks_of_time_without_forward_storage_and_inside_an_iteration(oracle_c):
    """KR_NO_STORE_FW: the initial forward sweep only needs its final states -- block propagators and the sweep over the boundaries, no
    parallel pass; and a whole iteration (backward sweep in blocks of time feeding the update sweep) against the oracle."""
    p = wl.dense_problem(64, 96, 64, 2, 5602, 0.05, 2.0)
    K = 64
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K, store_fw=False) as eng:
        eng.set_initial_states(p["psi0"])
        psiT = eng.forward(p["eps"])
        psiT_o, _ = oracle_c.forward(p["H0"], p["Hc"], p["tlist"], p["eps"], p["psi0"], store=False, mode=1, nthreads=8)
        assert np.max(np.abs(psiT - psiT_o)) < 1e-11
        _, _, chiT = wl.chi_and_JT("ss", psiT_o, p["tgt"])
        eps_new, ga, psiT1 = eng.iteration(p["eps"], chiT, p["S"], p["lam"])
    eps_o, ga_o, psiT1_o, _, _ = oracle_c.iteration(p["H0"], p["Hc"], p["tlist"], p["eps"], p["S"], p["lam"], p["psi0"], chiT, mode=1, nthreads=8)
    assert _rel(eps_new, eps_o) < 1e-10 and _rel(ga, ga_o) < 1e-9 and np.max(np.abs(psiT1 - psiT1_o)) < 1e-9


@pytest.mark.parametrize("shape", [(300, 4, 20, 1), (130, 5, 40, 1)], ids=["batched_known_sweeps_np512", "pair_np256"])
def test_rowsplit_coherence_check_on_another_tile_and_on_the_batch_path(shape):
    """The planted fault of round 4 sat in workgroup 1 (tile 0) of a single launch.  RS_FAULT=w now names the workgroup: w = R + 1 is a
    member of tile 1, and with more trajectories than one launch carries the known-pulse sweeps run in BATCHES of launches, each with
    its own check over its own tiles (rowsplit_postlaunch) -- the fault must be found there too, with the tile in the message."""
    N, NT, K, L = shape
    p = wl.dense_problem(N, NT, K, L, 5701, 0.02, 2.0)
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K) as eng:
        R = eng.info()["rowsplit_R"]
        assert R > 1, eng.plan()
        eng.set_initial_states(p["psi0"])
        eng.forward(p["eps"])
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K, options={"RS_FAULT": R + 1}) as eng:
        eng.set_initial_states(p["psi0"])
        with pytest.raises(kr.KrotovHipError) as ei:
            eng.forward(p["eps"])
        assert ei.value.code == capi.KR_ERR_INTERNAL and "share trajectory tile 1 " in str(ei.value), str(ei.value)


@pytest.mark.parametrize("shape", [(100, 6, 6, 2), (300, 4, 12, 2)], ids=["single_np128", "turns_np512"])
def test_rowsplit_update_sweeps_compare_the_workgroups_totals(shape, oracle_c):
    """The row-split update kernels form eps_new per workgroup as well; their coherence words compare the copies of a trajectory's STATE
    inside a group, which a wrong total in one member does not disturb.  Round 5: they log the totals they gathered like the N <= 64
    kernels (kr_pulse_finish, comparison only -- these kernels still write eps_new themselves; not the pair mode, whose kernel sits at
    the register limit: either placement of the store cost C5 1.4 %).  Clean run against the oracle; with
    PULSE_FAULT workgroup 1 logs a wrong total at step 1 and the call fails with KR_ERR_INTERNAL."""
    N, NT, K, L = shape
    p = wl.dense_problem(N, NT, K, L, 5801, 0.02, 2.0)
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K) as eng:
        info = eng.info()
        assert info["rowsplit_R"] > 1 and info["generic_update"] == 0, eng.plan()
        assert "per-step totals of all workgroups compared on" in eng.plan(), eng.plan()
        eng.set_initial_states(p["psi0"])
        psiT = eng.forward(p["eps"])
        _, _, chiT = wl.chi_and_JT(p["functional"], psiT, p["tgt"])
        eps_new = eng.iteration(p["eps"], chiT, p["S"], p["lam"])[0]
    eps_o = oracle_c.iteration(p["H0"], p["Hc"], p["tlist"], p["eps"], p["S"], p["lam"], p["psi0"], chiT, mode=1, nthreads=8)[0]
    assert _rel(eps_new, eps_o) < 1e-10
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K, options={"PULSE_FAULT": 1}) as eng:
        eng.set_initial_states(p["psi0"])
        eng.forward(p["eps"])
        with pytest.raises(kr.KrotovHipError) as ei:
            eng.iteration(p["eps"], chiT, p["S"], p["lam"])
        assert ei.value.code == capi.KR_ERR_INTERNAL and "gathered other totals than the reference workgroup" in str(ei.value), str(ei.value)
