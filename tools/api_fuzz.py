"""Developer tool: random SEQUENCES of calls on one handle against the C oracle (tools/oracle_fuzz.py always does forward + two
iterations): kr_forward, kr_backward (+ reads of the backward storage), kr_update_forward on the stored co-states -- possibly several
times on the same backward sweep, with other pulses and update shapes --, kr_iteration, reads of the forward storage the last update
sweep wrote, kr_final_states, kr_g_a_int, kr_set_initial_states with other states in the middle.  State that one call leaves for the
next (the stored co-states, the tag sequence of the exchange, the resident pulses, sticky status words) is what this looks at.
   python tools/api_fuzz.py [COUNT=60] [SEED=1]        (ORACLE_FUZZ_DEVLIST=1: two shards on the one GPU per handle)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import krotov_jl_amd as kr  # noqa: E402
from krotov_jl_amd import workloads as wl  # noqa: E402
import oracle_c as oc  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for it in range(count):
    N = int(rng.choice([2, 5, 9, 16, 24, 33, 48, 64, 80, 130, 140]))
    kmax = 200 if N <= 64 else 20
    K = int(rng.integers(2, kmax + 1))
    L = int(rng.choice([1, 2, 2, 3]))
    NT = int(rng.choice([1, 2, 5, 9, 20]))
    p = wl.dense_problem(N, NT, K, L, 3000 + it, float(rng.choice([0.02, 0.05, 0.2])), 2.0)
    tag = ""
    if rng.random() < 0.3:
        p["H0"] = p["H0"] - 1j * 0.05 * np.diag(rng.uniform(0, 1, N))
        tag += " nonherm"
    if rng.random() < 0.3 and NT > 1:
        p["tlist"] = np.concatenate([[0.0], np.cumsum(np.diff(p["tlist"]) * rng.uniform(0.5, 1.5, NT))])
        tag += " grid"
    kw = {}
    if os.environ.get("ORACLE_FUZZ_DEVLIST"):
        kw["devices"] = [0, 0]
        tag += " devlist"
    H0, Hc, tl = p["H0"], p["Hc"], p["tlist"]
    t0 = time.time()
    trace = []
    worst = 0.0

    def check(what, got, want, tol, rel=False):
        global worst
        d = np.max(np.abs(np.asarray(got) - np.asarray(want)))
        if rel:
            d /= max(np.max(np.abs(want)), 1e-300)
        worst = max(worst, d / tol)
        if not d < tol:
            raise AssertionError("%s off by %.2e (tolerance %.0e) after [%s]" % (what, d, tol, " ".join(trace)))

    try:
        with kr.HipEngine(H0, Hc, tl, K, **kw) as eng:
            psi0 = p["psi0"]
            eng.set_initial_states(psi0)
            eps = p["eps"]                      # the pulses the next call takes as its guess
            X = None                            # oracle copy of the co-states the handle has stored, and the pulses they belong to
            Phi = None                          # oracle copy of the forward storage of the last update sweep
            psiT_o = None
            for step in range(int(rng.integers(5, 11))):
                ops = ["forward", "backward", "iteration", "newstates"]
                if X is not None:
                    ops += ["update", "update", "read_bw"]
                if Phi is not None:
                    ops += ["read_fw", "final", "ga"]
                op = str(rng.choice(ops))
                trace.append(op)
                S = p["S"] * rng.uniform(0.3, 1.0, (L, 1))
                lam = rng.uniform(0.5, 8.0, L)
                if op == "forward":
                    psiT = eng.forward(eps)
                    psiT_o = oc.forward(H0, Hc, tl, eps, psi0, store=False)[0]
                    check("forward states", psiT, psiT_o, 1e-11)
                elif op == "newstates":       # other initial states: the stored co-states stay valid, the forward storage does not describe them
                    psi0 = np.roll(psi0, 1, axis=0) * np.exp(1j * rng.uniform(0, 6.28))
                    eng.set_initial_states(psi0)
                    Phi = None
                elif op == "backward":
                    if psiT_o is None:
                        psiT_o = oc.forward(H0, Hc, tl, eps, psi0, store=False)[0]
                    chiT = wl.chi_and_JT(str(rng.choice(["sm", "ss", "re"])), psiT_o, p["tgt"])[2]
                    eng.backward(eps, chiT)
                    X = oc.backward(H0, Hc, tl, eps, chiT)
                    X_eps = eps
                elif op == "read_bw":
                    k, n = int(rng.integers(0, K)), int(rng.integers(0, NT + 1))
                    check("backward storage [%d, %d]" % (k, n), eng.read_storage("bw", k, n), X[n, k], 1e-11)
                elif op == "update":          # on the co-states of the last backward sweep -- guess pulses = the pulses they were propagated under
                    e2, ga, psiT = eng.update_forward(X_eps, S, lam)
                    e2_o, ga_o, psiT_o2, Phi = oc.update_forward(H0, Hc, tl, X_eps, S, lam, psi0, X)
                    check("updated pulses", e2, e2_o, 1e-10, rel=True)
                    check("final states of the update", psiT, psiT_o2, 1e-10)
                    if np.max(np.abs(ga_o)) > 1e-20:
                        check("running integral", ga, ga_o, 1e-9 * np.max(np.abs(ga_o)))
                    ga_last, psiT_last = ga_o, psiT_o2
                    eps, psiT_o = e2_o, psiT_o2
                elif op == "iteration":
                    if psiT_o is None:
                        psiT_o = oc.forward(H0, Hc, tl, eps, psi0, store=False)[0]
                    chiT = wl.chi_and_JT(str(rng.choice(["sm", "ss", "re"])), psiT_o, p["tgt"])[2]
                    e2, ga, psiT = eng.iteration(eps, chiT, S, lam)
                    e2_o, ga_o, psiT_o2, X, Phi = oc.iteration(H0, Hc, tl, eps, S, lam, psi0, chiT)
                    X_eps = eps
                    check("updated pulses", e2, e2_o, 1e-10, rel=True)
                    check("final states of the iteration", psiT, psiT_o2, 1e-10)
                    ga_last, psiT_last = ga_o, psiT_o2
                    eps, psiT_o = e2_o, psiT_o2
                elif op == "read_fw":
                    k, n = int(rng.integers(0, K)), int(rng.integers(0, NT + 1))
                    check("forward storage [%d, %d]" % (k, n), eng.read_storage("fw", k, n), Phi[n, k], 1e-10)
                elif op == "final":
                    check("kr_final_states", eng.final_states(), psiT_last, 1e-10)
                elif op == "ga":
                    if np.max(np.abs(ga_last)) > 1e-20:
                        check("kr_g_a_int", eng.g_a_int(), ga_last, 1e-9 * np.max(np.abs(ga_last)))
            info = eng.info()
        print("ok   N=%4d K=%4d L=%d NT=%2d NP=%4d R=%3d wg=%3d  worst / tolerance %.1e  %.1fs%s  [%s]" % (
            N, K, L, NT, info["NP"], info["rowsplit_R"], info["workgroups"], worst, time.time() - t0, tag, " ".join(trace)), flush=True)
    except Exception as exc:  # noqa: BLE001
        bad += 1
        print("FAIL N=%4d K=%4d L=%d NT=%2d%s: %s" % (N, K, L, NT, tag, str(exc)[:1500]), flush=True)
print("api_fuzz: %d problem(s) in %d shapes" % (bad, count))
sys.exit(1 if bad else 0)
