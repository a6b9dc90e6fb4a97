"""Developer tool: cross-check of the PLANS.  Random shapes (N, K, L) around the boundaries between the kernel families run one
Krotov iteration on the plan kr_create picks and on the generic multi-pass kernels (KROTOV_HIP_GENERIC=1, independent code);
pulses and final states must agree to 1e-10.  Catches hangs (bounded spins -> error), layout slips and wrong plan limits.
   python tools/plan_fuzz.py [COUNT=60] [SEED=1]        (FUZZ_MAXWG=64,100: draw kr_config.max_workgroups from that list;
    FUZZ_NMAX=64: small state dimensions with up to 4200 trajectories)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import krotov_jl_amd as kr
from krotov_jl_amd import workloads as wl

count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
NS = [9, 16, 33, 64, 65, 100, 128, 130, 200, 256, 257, 300, 400, 512, 513, 700, 1024]
KS = [1, 2, 3, 5, 8, 9, 16, 17, 31, 32, 33, 64, 65, 100, 128, 130, 250, 256, 257, 300, 520]
if os.environ.get("FUZZ_NMAX"):          # e.g. FUZZ_NMAX=64: the register-resident / matrix-core kernels with many trajectories
    NS = [n for n in NS if n <= int(os.environ["FUZZ_NMAX"])] + [8, 12, 20, 32, 48, 64]
    KS = KS + [600, 1024, 1100, 2048, 2100, 4096, 4200]
bad = 0
for it in range(count):
    N = int(rng.choice(NS)); K = int(rng.choice(KS)); L = int(rng.choice([1, 1, 2, 2, 3, 4])); NT = int(rng.choice([3, 4, 34]))
    if N * K * (NT + 1) > 40e6 or (N > 512 and K > 64):
        continue
    p = wl.dense_problem(N, NT, K, L, 5000 + it, 0.03, 2.0)
    kw = {}
    tag = ""
    if os.environ.get("FUZZ_MAXWG"):     # a share of the GPU (several ranks or handles per device): the plans change with it
        kw["max_workgroups"] = int(rng.choice([int(x) for x in os.environ["FUZZ_MAXWG"].split(",")]))
        tag += " maxwg=%d" % kw["max_workgroups"]
    if rng.random() < 0.25:      # a drift that is not Hermitian (weak or strong dissipation)
        p["H0"] = p["H0"] - 1j * float(rng.choice([0.003, 0.2])) * np.diag(rng.uniform(0, 1, N))
        tag += " nonherm"
    if rng.random() < 0.25 and K >= 2:   # two generators, trajectories assigned at random
        p["H0"] = np.stack([p["H0"], 0.9 * p["H0"]])
        p["Hc"] = np.stack([p["Hc"], 1.1 * p["Hc"]])
        kw["traj_generator"] = rng.integers(0, 2, K).astype(np.int32)
        tag += " G=2"
    if os.environ.get("FUZZ_EXTRA"):     # FUZZ_EXTRA=1: also masks (control_derivs === nothing for one control) and a looser tolerance
        if L >= 2 and rng.random() < 0.4:
            G_ = p["H0"].shape[0] if p["H0"].ndim == 3 else 1
            pres = np.ones((G_, L), np.uint8)
            pres[int(rng.integers(0, G_)), int(rng.integers(0, L))] = 0
            kw["hc_present"] = pres
            tag += " mask"
        if rng.random() < 0.3:
            kw["prop_tol"] = 1e-11
            tag += " tol"
    if rng.random() < 0.25:      # a non-uniform time grid
        dts = np.diff(p["tlist"]) * rng.uniform(0.6, 1.4, NT)
        p["tlist"] = np.concatenate([[0.0], np.cumsum(dts)])
        tag += " grid"
    res = {}
    t0 = time.time()
    try:
        for gen in ["", "1"]:
            if gen:
                os.environ["KROTOV_HIP_GENERIC"] = gen
            else:
                os.environ.pop("KROTOV_HIP_GENERIC", None)
            with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K, **kw) as eng:
                eng.set_initial_states(p["psi0"])
                psiT = eng.forward(p["eps"])
                _, _, chiT = wl.chi_and_JT("ss", psiT, p["tgt"])
                eps_new, ga, psiT1 = eng.iteration(p["eps"], chiT, p["S"], p["lam"])
                if not gen:
                    info = eng.info()
            res[gen] = (psiT, eps_new, psiT1)
        d0 = np.max(np.abs(res[""][0] - res["1"][0]))
        d1 = np.max(np.abs(res[""][1] - res["1"][1])) / max(np.max(np.abs(res["1"][1])), 1e-300)
        d2 = np.max(np.abs(res[""][2] - res["1"][2]))
        lim = 1e-10 if "prop_tol" not in kw else 1e-8   # (a looser truncation bound: the two plans' polynomials differ by up to it per step)
        ok = d0 < lim and d1 < lim and d2 < lim and np.max(np.abs(res[""][1] - p["eps"])) > 0
        print("%-4s N=%4d K=%4d L=%d NT=%2d  NP=%4d R=%3d wg=%3d Kp=%4d genK=%d genU=%d  dpsi %.1e deps %.1e dpsi1 %.1e  %.1fs%s" % (
            "ok" if ok else "BAD", N, K, L, NT, info["NP"], info["rowsplit_R"], info["workgroups"], info["K_padded"], info["generic_known"],
            info["generic_update"], d0, d1, d2, time.time() - t0, tag), flush=True)
        bad += 0 if ok else 1
    except Exception as exc:  # noqa: BLE001
        bad += 1
        print("FAIL N=%4d K=%4d L=%d NT=%2d: %s" % (N, K, L, NT, str(exc)[:160]), flush=True)
os.environ.pop("KROTOV_HIP_GENERIC", None)
print("plan_fuzz: %d problem(s)" % bad)
sys.exit(1 if bad else 0)
