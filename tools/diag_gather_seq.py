"""developer tool: run row-split shapes on libkrotov_hip_diag.so built with DIAG_DEFS=-DKR_GATHER_SEQ (tools/build_diag.sh kr_rs256 kr_rs128):
every all-gather element carries its gather counter in bits 15..1, and the kernels count elements that the one-bit validity accepted
although they belong to an earlier use of the buffer.  Results are perturbed at 2^-37 by the counter bits -- only the count matters.
usage: KROTOV_HIP_LIB=krotov.jl_amd/lib/libkrotov_hip_diag.so python tools/diag_gather_seq.py [reps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import krotov_jl_amd as kr  # noqa: E402
from krotov_jl_amd import workloads as wl  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    shapes = [(256, 36, 24, 1, 192, 0.01, 2.0), (100, 40, 6, 1, 191, 0.02, 2.0), (130, 35, 40, 1, 193, 0.02, 5.0), (256, 60, 24, 2, 7, 0.02, 2.0),
              (200, 50, 60, 2, 8, 0.02, 2.0), (128, 50, 128, 2, 9, 0.02, 2.0), (256, 40, 8, 2, 10, 0.02, 2.0)]
    total = 0
    for shape in shapes:
        p = wl.dense_problem(*shape)
        K = p["psi0"].shape[0]
        for opts in [{"STAMPS": 1}, {"STAMPS": 1, "RS_LMAX1": 0}, {"STAMPS": 1, "XCD_GROUP": 0}, {"STAMPS": 1, "RS_PAIR": 0, "RS_LMAX1": 0}]:
            events = 0
            first = None
            for r in range(reps):
                with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K, options=opts) as eng:
                    eng.set_initial_states(p["psi0"])
                    psiT = eng.forward(p["eps"])
                    _, _, chiT = wl.chi_and_JT(p["functional"], psiT, p["tgt"])
                    eng.iteration(p["eps"], chiT, p["S"], p["lam"])
                    st = eng.debug_stamps()
                    plan = eng.plan()
                    if st[4]:
                        events += st[4]
                        if first is None:
                            first = eng.debug_arrivals(1)[0].ravel()[:4]
            total += events
            print("%r %r: stale elements accepted: %d in %d runs %s | %s" % (shape, opts, events, reps, "" if first is None else "last event (block, counter, element, seq re|im): %r" % (list(map(int, first)),),
                                                                            plan.split(": update sweep = ")[1][:60]), flush=True)
    print("TOTAL stale elements accepted:", total)


if __name__ == "__main__":
    main()
