"""developer tool: the sequence of handles of tests/test_gpu_parity.py::test_rowsplit_one_control_instantiation, repeated, every result
compared with an exact reference -- a result that depends on what the PREVIOUS handle left behind (recycled device memory, caches)
cannot show when one problem is run over and over, it needs different problems in turn.  usage: python tools/diag_rowsplit_sequence.py [reps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import krotov_jl_amd as kr  # noqa: E402
from krotov_jl_amd import workloads as wl  # noqa: E402
from diag_rowsplit_case import reference  # noqa: E402


def iterate(p):
    K = p["psi0"].shape[0]
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K) as eng:
        eng.set_initial_states(p["psi0"])
        psiT = eng.forward(p["eps"])
        _, _, chiT = wl.chi_and_JT(p["functional"], psiT, p["tgt"])
        out = eng.iteration(p["eps"], chiT, p["S"], p["lam"])
        return psiT, out, eng.plan()


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    cases = [wl.dense_problem(100, 40, 6, 1, 191, 0.02, 2.0), wl.dense_problem(256, 36, 24, 1, 192, 0.01, 2.0),
             wl.dense_problem(130, 35, 40, 1, 193, 0.02, 5.0)]
    cases[2]["H0"] = cases[2]["H0"] - 0.03j * np.diag(np.random.default_rng(5).uniform(0, 1, 130))
    refs = [reference(p) for p in cases]
    print("references done", flush=True)
    worst = {}
    nbad = 0
    loud = {}
    # DIAG_POISON=n: first n device-list handles of mixed shapes (two shards on the one GPU: uncached mailboxes between the shards, big-grid
    # uncached and small-grid ordinary mailboxes), created, used once and destroyed -- what tests/test_gpu_device_list.py leaves behind in
    # the test process before tests/test_gpu_parity.py runs (with KROTOV_HIP_UC_POOL=0 their uncached blocks are freed: DESIGN.md)
    rng = np.random.default_rng(11)
    for i in range(int(os.environ.get("DIAG_POISON", "0"))):
        N = int(rng.choice([9, 16, 33, 48, 64])); K = int(rng.choice([8, 40, 107, 130, 256])); L = int(rng.choice([1, 2]))
        q = wl.dense_problem(N, 5, K, L, 900 + i, 0.05, 2.0)
        try:
            with kr.HipEngine(q["H0"], q["Hc"], q["tlist"], K, devices=[0, 0], spin_timeout_ms=300) as eng:
                eng.set_initial_states(q["psi0"])
                psiT = eng.forward(q["eps"])
                _, _, chiT = wl.chi_and_JT("ss", psiT, q["tgt"])
                eng.iteration(q["eps"], chiT, q["S"], q["lam"])
        except kr.KrotovHipError as exc:
            print("poison handle %d (N=%d K=%d): error %d" % (i, N, K, exc.code), flush=True)
    for r in range(reps):
        for ci, p in enumerate(cases):
            for mode in ["0", "1"]:
                os.environ["KROTOV_HIP_RS_LMAX1"] = mode
                try:
                    psiT, (eps_new, ga, psiT1), plan = iterate(p)
                except kr.KrotovHipError as exc:     # loud failures (bounded spins under a loaded GPU, integrity checks) are counted, not hidden
                    loud[exc.code] = loud.get(exc.code, 0) + 1
                    if loud[exc.code] <= 3:
                        print("rep %d case %d mode %s: error %d: %s" % (r, ci, mode, exc.code, str(exc)[:200]), flush=True)
                    continue
                e_fw = np.max(np.abs(psiT - refs[ci]["psiT"]))
                e_eps = np.max(np.abs(eps_new - refs[ci]["eps_new"])) / np.max(np.abs(refs[ci]["eps_new"]))
                key = (ci, mode)
                worst[key] = max(worst.get(key, 0.0), e_eps)
                if e_eps > 1e-10 or e_fw > 1e-11:
                    nbad += 1
                    d = (eps_new - refs[ci]["eps_new"])[0]
                    print("rep %d case %d mode %s: fw %.2e eps %.2e | d[:6] %s d[-6:] %s | %s" % (r, ci, mode, e_fw, e_eps, d[:6], d[-6:], plan), flush=True)
    print("worst relative pulse error per (case, mode):", {k: "%.1e" % v for k, v in worst.items()}, "silently wrong runs:", nbad,
          "loud failures by status code:", loud, flush=True)


if __name__ == "__main__":
    main()
