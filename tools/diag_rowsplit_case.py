"""developer tool: one row-split shape run many times under several option sets, every run compared with an exact reference
(scipy expm per step, shared by the trajectories), so that a run- or placement-dependent result shows up as an outlier and names
the sweep it comes from.  usage: python tools/diag_rowsplit_case.py [N NT K L seed dt lam] [reps]
"""
import os
import sys
import time

import numpy as np
import scipy.linalg as sla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import krotov_jl_amd as kr  # noqa: E402
from krotov_jl_amd import workloads as wl  # noqa: E402


def reference(p):
    H0, Hc, tl, eps = p["H0"], p["Hc"], p["tlist"], p["eps"]
    K, N = p["psi0"].shape
    L, NT = eps.shape
    dts = np.diff(tl)

    def U(e, n):
        return sla.expm(-1j * (H0 + np.tensordot(e, Hc, 1)) * dts[n])
    Ug = [U(eps[:, n], n) for n in range(NT)]
    psi = p["psi0"].T.copy()
    for n in range(NT):
        psi = Ug[n] @ psi
    psiT = psi.T.copy()
    _, _, chiT = wl.chi_and_JT(p["functional"], psiT, p["tgt"])
    X = np.empty((NT + 1, K, N), complex)
    X[NT] = chiT
    chi = chiT.T.copy()
    for n in range(NT - 1, -1, -1):
        chi = Ug[n].conj().T @ chi
        X[n] = chi.T
    psi = p["psi0"].T.copy()
    eps_new = eps.copy()
    for n in range(NT):
        du = np.array([np.sum(np.einsum("ik,ik->k", X[n].T.conj(), Hc[l] @ psi).imag) for l in range(L)])
        eps_new[:, n] = eps[:, n] + p["S"][:, n] / p["lam"] * du
        psi = U(eps_new[:, n], n) @ psi
    return dict(psiT=psiT, chiT=chiT, X0=X[0], eps_new=eps_new, psiT1=psi.T.copy())


def run(p, ref, options):
    K = p["psi0"].shape[0]
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K, options=options) as eng:
        eng.set_initial_states(p["psi0"])
        psiT = eng.forward(p["eps"])
        eps_new, ga, psiT1 = eng.iteration(p["eps"], ref["chiT"], p["S"], p["lam"])
        X0 = np.array([eng.read_storage("bw", k, 0) for k in range(K)])
        info = eng.info()
    return (np.max(np.abs(psiT - ref["psiT"])), np.max(np.abs(X0 - ref["X0"])) / np.max(np.abs(ref["X0"])),
            np.max(np.abs(eps_new - ref["eps_new"])) / np.max(np.abs(ref["eps_new"])), np.max(np.abs(psiT1 - ref["psiT1"])), info, eps_new)


def main():
    args = sys.argv[1:]
    shape = (256, 36, 24, 1, 192, 0.01, 2.0)
    reps = 40
    if len(args) >= 7:
        shape = (int(args[0]), int(args[1]), int(args[2]), int(args[3]), int(args[4]), float(args[5]), float(args[6]))
        args = args[7:]
    if args:
        reps = int(args[0])
    p = wl.dense_problem(*shape)
    t0 = time.time()
    ref = reference(p)
    print("shape %r reference in %.1f s" % (shape, time.time() - t0), flush=True)
    variants = [{}, {"RS_LMAX1": 0}, {"XCD_GROUP": 0}, {"XCD_GROUP": 0, "RS_LMAX1": 0}, {"ECON": 0}, {"ECON": 0, "RS_LMAX1": 0},
                {"GATHER_DELAY_LOCAL": 0}, {"GATHER_DELAY_LOCAL": 0, "RS_LMAX1": 0}, {"GATHER_DELAY": 0, "XCD_GROUP": 0},
                {"RS_POLL_DELAY": -1}, {"RS_POLL_DELAY": 0}]
    extra = os.environ.get("DIAG_VARIANTS")
    if extra:   # "A=1,B=2;C=3"
        variants = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in v.split(",") if kv) for v in extra.split(";")]
    for opts in variants:
        errs = []
        first = None
        nbit = 0
        for r in range(reps):
            e = run(p, ref, opts)
            errs.append(e[:4])
            if first is None:
                first = e[5]
                plan = e[4]
            elif not np.array_equal(first, e[5]):
                nbit += 1
        errs = np.array(errs)
        bad = int(np.sum(errs[:, 2] > 1e-10))
        print("%-44s R=%d LMAX=%d | fw psi(T) max %.2e | bw X[0] max %.2e | eps_new med %.2e max %.2e | psi(T)' max %.2e | outliers %d/%d, runs != first run bitwise %d"
              % (opts, plan["rowsplit_R"], plan["LMAX"], errs[:, 0].max(), errs[:, 1].max(), np.median(errs[:, 2]), errs[:, 2].max(), errs[:, 3].max(), bad, reps, nbit), flush=True)
        if bad:
            for r in np.nonzero(errs[:, 2] > 1e-10)[0][:5]:
                print("    run %d: fw %.2e bw %.2e eps %.2e psiT' %.2e" % (r, *errs[r]), flush=True)


if __name__ == "__main__":
    main()
