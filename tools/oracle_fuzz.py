"""Developer tool: random problems DIRECTLY against the oracle (tools/plan_fuzz.py compares the plans with the generic kernels, which
tests/test_gpu_generic.py pins to the oracle; this one skips the middle man).  Random N, K, L, N_T, Hermitian or dissipative drifts,
one or two generators with random assignment, absent control derivatives (control_derivs === nothing, optimize.jl:344), non-uniform
grids, all three functionals, random update shapes and lambda_a per control: one forward sweep + two chained iterations on the HIP
engine and on the NumPy oracle (exact step propagators), states to 1e-11, pulses to 1e-10 relative, the UPDATE itself (delta eps, which
the pulses' rounding hides) to 1e-8 of its largest value, the running integral to 1e-9.
   python tools/oracle_fuzz.py [COUNT=150] [SEED=1]        (ORACLE_FUZZ_NMAX=130: state dimensions up to 130 only;
    ORACLE_FUZZ_DEVLIST=1: every handle drives two shards on the one GPU (kr_config.devices = {0, 0}), = 3: three;
    ORACLE_FUZZ_DEVCHAIN=1: the two iterations also as one device-resident chain (kr_iteration_functional_dev);
    ORACLE_FUZZ_KBIG=1: most shapes at N <= 64 with 300 ... 5000 trajectories;
    ORACLE_FUZZ_METHODS=1: most shapes with prop_method = Chebyshev / Newton or the generic Taylor kernels forced;
    ORACLE_FUZZ_HARD=1: half the shapes with ||H dt|| up to ~6 -- sub-stepping --, many with lambda_a / 50 -- large updates;
    ORACLE_FUZZ_TBLOCK=1: known-pulse sweeps in blocks of 2-5 time steps wherever the plan allows them (32 < N <= 64, N_T >= 2 S);
    ORACLE_FUZZ_NS=33,48,64: draw N from this list only)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import krotov_jl_amd as kr  # noqa: E402
from krotov_jl_amd import workloads as wl  # noqa: E402
import krotov_oracle as o  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for it in range(count):
    NS = [1, 2, 3, 5, 8, 9, 16, 17, 24, 32, 33, 48, 64, 65, 80, 100, 128, 130, 160, 200, 256, 257, 300]
    if os.environ.get("ORACLE_FUZZ_NMAX"):   # (the oracle's exact propagators cost seconds per shape at N >= 200: the test suite stays below)
        NS = [n for n in NS if n <= int(os.environ["ORACLE_FUZZ_NMAX"])]
    if os.environ.get("ORACLE_FUZZ_NS"):
        NS = [int(x) for x in os.environ["ORACLE_FUZZ_NS"].split(",")]
    if os.environ.get("ORACLE_FUZZ_NMIN"):   # (large state dimensions only: the row-split kernels of N > 128)
        NS = [n for n in NS if n >= int(os.environ["ORACLE_FUZZ_NMIN"])]
    N = int(rng.choice(NS))
    kmax = 300 if N <= 64 else (64 if N <= 130 else (24 if N <= 256 else 6))
    K = int(rng.integers(1, kmax + 1))
    if os.environ.get("ORACLE_FUZZ_KBIG") and N <= 64 and rng.random() < 0.6:   # many trajectories per GPU: several per workgroup, tile sets in turn
        K = int(rng.integers(300, 5001))                                        # (N = 64: matrix-core sweep), beyond 4096 tiles the generic kernels
    L = int(rng.choice([1, 1, 2, 2, 3, 4]))
    NT = int(rng.choice([1, 2, 5, 9, 34]))
    if N > 256 and L > 2:
        L = 2
    p = wl.dense_problem(N, NT, K, L, 7000 + it, float(rng.choice([0.02, 0.05, 0.2])), 2.0)
    tag = ""
    kw = {}
    if os.environ.get("ORACLE_FUZZ_HARD") and rng.random() < 0.5:   # ||H dt|| up to ~6: sub-stepped steps; with a small lambda_a also updates
        hs = float(rng.choice([5.0, 30.0]))                          # that leave the margin of the economized polynomial (Taylor fallback)
        p["H0"] = p["H0"] * hs
        tag += " x%g" % hs
    if rng.random() < 0.3:
        p["H0"] = p["H0"] - 1j * float(rng.choice([0.003, 0.05, 0.3])) * np.diag(rng.uniform(0, 1, N)) + (0.02 * rng.normal(size=(N, N)) if rng.random() < 0.5 else 0.0)
        tag += " nonherm"
    G = 1
    tg = np.zeros(K, np.int32)
    H0s, Hcs = p["H0"][None], p["Hc"][None]
    if rng.random() < 0.25 and K >= 2 and N <= 256:
        G = 2
        H0s = np.stack([p["H0"], 0.8 * p["H0"]])
        Hcs = np.stack([p["Hc"], 1.2 * p["Hc"]])
        tg = rng.integers(0, 2, K).astype(np.int32)
        kw["traj_generator"] = tg
        tag += " G=2"
    pres = np.ones((G, L), np.uint8)
    if L >= 2 and rng.random() < 0.3:
        pres[int(rng.integers(0, G)), int(rng.integers(0, L))] = 0
        kw["hc_present"] = pres
        tag += " mask"
    if rng.random() < 0.3 and NT > 1:
        p["tlist"] = np.concatenate([[0.0], np.cumsum(np.diff(p["tlist"]) * rng.uniform(0.5, 1.5, NT))])
        tag += " grid"
    p["S"] = p["S"] * rng.uniform(0.3, 1.0, (L, 1))
    p["lam"] = rng.uniform(0.5, 8.0, L)
    if os.environ.get("ORACLE_FUZZ_HARD") and rng.random() < 0.4:
        # large updates -- but inside the stability range of the sequential update: the step eps_n -> psi(t_n+1) -> Delta u_(n+1) is a
        # feedback loop whose gain per step is ~ (S / lambda_a) ||mu||^2 dt; beyond ~1 rounding noise grows geometrically along the
        # grid and ANY two implementations part ways (seen with lambda_a = 0.024, dt = 0.2: the persistent and the generic kernels
        # agreed to 1e-5 only, both O(1) from the oracle -- an ill-conditioned problem, not a defect).  Keep the gain below 1/3.
        dtm = float(np.max(np.diff(p["tlist"])))
        mu2 = np.array([max(np.linalg.norm(Hcs[g][l], 2) ** 2 for g in range(G)) for l in range(L)])
        p["lam"] = np.maximum(p["lam"] * 0.02, 3.0 * dtm * np.max(p["S"], axis=1) * mu2)
        tag += " smalllam"
    fname = str(rng.choice(["sm", "ss", "re"]))
    if os.environ.get("ORACLE_FUZZ_ONLY") and it != int(os.environ["ORACLE_FUZZ_ONLY"]):   # (replay ONE shape of a run: the draws above are consumed all the same)
        continue
    if it < int(os.environ.get("ORACLE_FUZZ_FROM", "0")):   # (skip the first shapes of a run, draws consumed)
        continue
    t0 = time.time()
    try:
        # oracle: per-trajectory generators by groups (the oracle's routines take one generator)
        def o_forward(eps):
            out = np.empty((K, N), complex)
            for g in range(G):
                m = tg == g
                if m.any():
                    out[m] = o.forward(H0s[g], Hcs[g], p["tlist"], eps, p["psi0"][m], store=False)[0]
            return out

        def o_iteration(eps, chiT):
            X = np.empty((NT + 1, K, N), complex)
            for g in range(G):
                m = tg == g
                if m.any():
                    X[:, m] = o.backward(H0s[g], Hcs[g], p["tlist"], eps, chiT[m])
            psi = p["psi0"].copy()
            eps_new = eps.copy()
            ga = np.zeros(L)
            for n in range(NT):
                dt = p["tlist"][n + 1] - p["tlist"][n]
                for l in range(L):
                    du = 0.0
                    for k in range(K):
                        if pres[tg[k], l]:
                            du += np.vdot(X[n, k], Hcs[tg[k], l] @ psi[k]).imag
                    alpha = p["S"][l, n] / p["lam"][l]
                    eps_new[l, n] = eps[l, n] + alpha * du
                    ga[l] += alpha * du * du * dt
                for g in range(G):
                    m = tg == g
                    if m.any():
                        psi[m] = psi[m] @ o.step_propagator(H0s[g], Hcs[g], eps_new[:, n], dt).T
            return eps_new, ga, psi

        Harg = (H0s[0], Hcs[0]) if G == 1 else (H0s, Hcs)
        if os.environ.get("ORACLE_FUZZ_METHODS") and rng.random() < 0.7:
            # kr_config.prop_method (the reference's prop_method): Chebyshev (Hermitian generators only), Newton at Leja points, or the
            # generic multi-pass kernels with Taylor steps -- all three on the generic kernels
            m = str(rng.choice(["cheby", "newton", "generic"])) if "nonherm" not in tag else str(rng.choice(["newton", "generic"]))
            if m == "generic":
                kw["options"] = {"GENERIC": 1}
            else:
                kw["prop_method"] = kr._capi.KR_PROP_CHEBY if m == "cheby" else kr._capi.KR_PROP_NEWTON
            tag += " " + m
        if os.environ.get("ORACLE_FUZZ_TBLOCK") and "options" not in kw:
            # the known-pulse sweeps in blocks of time (32 < N <= 64) also on the fuzzer's short grids: blocks of 2-5 steps from 4 intervals on
            kw["options"] = {"TBLOCK": int(rng.choice([2, 3, 4, 5])), "TBLOCK_MIN_NT": 4}
            tag += " tblock%d" % kw["options"]["TBLOCK"]
        if os.environ.get("ORACLE_FUZZ_DEVLIST") and K >= 2:
            # ONE handle with a device list that names the GPU twice: two shards (library threads, peer-pointer mailboxes) share it --
            # the inter-GPU exchange of the update sweep on random shapes
            kw["devices"] = [0, 0, 0] if (os.environ["ORACLE_FUZZ_DEVLIST"] == "3" and K >= 3) else [0, 0]   # (= 3: three shards)
            tag += " devlist%d" % len(kw["devices"])
        with kr.HipEngine(Harg[0], Harg[1], p["tlist"], K, **kw) as eng:
            eng.set_initial_states(p["psi0"])
            psiT = eng.forward(p["eps"])
            psiT_o = o_forward(p["eps"])
            d0 = np.max(np.abs(psiT - psiT_o))
            eps, eps_o = p["eps"], p["eps"]
            d1 = d2 = d3 = d4 = 0.0
            for _ in range(2):
                _, _, chiT = wl.chi_and_JT(fname, psiT_o, p["tgt"])
                guess = eps_o
                eps, ga, psiT = eng.iteration(guess, chiT, p["S"], p["lam"])
                eps_o, ga_o, psiT_o = o_iteration(guess, chiT)
                d1 = max(d1, np.max(np.abs(eps - eps_o)) / max(np.max(np.abs(eps_o)), 1e-300))
                # the UPDATE itself (the pulses hide it: delta eps is a small correction of eps): relative to the largest update
                # (an update below 1e-9 of the pulse is rounding noise -- N = 1: Im<chi|H|psi> of a scalar system -- and is not compared)
                if np.max(np.abs(eps_o - guess)) > 1e-9 * np.max(np.abs(guess)):
                    d4 = max(d4, np.max(np.abs((eps - guess) - (eps_o - guess))) / np.max(np.abs(eps_o - guess)))
                d2 = max(d2, np.max(np.abs(psiT - psiT_o)))
                d3 = max(d3, np.max(np.abs(ga - ga_o)) / max(np.max(np.abs(ga_o)), 1e-300) if np.max(np.abs(ga_o)) > 1e-20 else 0.0)
            info = eng.info()
        d5 = 0.0
        if os.environ.get("ORACLE_FUZZ_DEVCHAIN"):
            # the same two iterations as ONE device-resident chain (tau, J_T and chi(T) formed on the device, pulses ping-pong in device
            # memory: kr_forward_dev / kr_iteration_functional_dev): must land on the oracle's chain as well
            import torch
            with kr.HipEngine(Harg[0], Harg[1], p["tlist"], K, **kw) as e2:
                e2.set_initial_states(p["psi0"])
                e2.set_target_states(p["tgt"])
                e2.set_update_params(p["S"], p["lam"])
                ea = torch.from_numpy(np.ascontiguousarray(p["eps"])).cuda()
                eb = torch.empty_like(ea)
                torch.cuda.synchronize()
                e2.forward_dev(ea.data_ptr())
                for _ in range(2):
                    e2.iteration_functional_dev({"sm": 0, "ss": 1, "re": 2}[fname], ea.data_ptr(), eb.data_ptr())
                    ea, eb = eb, ea
                e2.synchronize()
                d5 = max(np.max(np.abs(ea.cpu().numpy() - eps_o)) / max(np.max(np.abs(eps_o)), 1e-300), np.max(np.abs(e2.final_states() - psiT_o)) * 0.1)
            tag += " devchain %.1e" % d5
        ok = d0 < 1e-11 and d1 < 1e-10 and d2 < 1e-10 and d3 < 1e-9 and d4 < 1e-8 and d5 < 1e-10
        print("%-4s N=%4d K=%4d L=%d NT=%2d J_T_%s NP=%4d R=%3d wg=%3d genK=%d genU=%d  dpsi %.1e deps %.1e dupd %.1e dpsi' %.1e dga %.1e  %.1fs%s" % (
            "ok" if ok else "BAD", N, K, L, NT, fname, info["NP"], info["rowsplit_R"], info["workgroups"], info["generic_known"], info["generic_update"],
            d0, d1, d4, d2, d3, time.time() - t0, tag), flush=True)
        bad += 0 if ok else 1
    except Exception as exc:  # noqa: BLE001
        if "needs distinct devices" in str(exc):   # a shard on the generic kernels exchanges through RCCL: not with one GPU listed twice
            print("skip N=%4d K=%4d L=%d NT=%2d%s: %s" % (N, K, L, NT, tag, str(exc)[:120]), flush=True)
            continue
        bad += 1
        print("FAIL N=%4d K=%4d L=%d NT=%2d%s: %s" % (N, K, L, NT, tag, str(exc)[:1500]), flush=True)
print("oracle_fuzz: %d problem(s) in %d shapes" % (bad, count))
sys.exit(1 if bad else 0)
