"""developer tool: run shapes of every kernel family on a library built with -DKR_GATHER_SEQ
  (DIAG_TAG=seq DIAG_DEFS=-DKR_GATHER_SEQ tools/build_diag.sh kr_np64 kr_np32 kr_np16 kr_np8 kr_mf64_l2 kr_rs256 kr_rs128):
every all-gather element of the row-split kernels carries its gather counter in bits 15..1, every partial sum of the per-step exchange
its step index, and the kernels count what the one-bit validity accepted although it belongs to an earlier use of the buffer.
Results are perturbed at 2^-37 by the counter bits -- only the counts matter.
usage: KROTOV_HIP_LIB=$PWD/krotov.jl_amd/lib/libkrotov_hip_seq.so python tools/diag_gather_seq.py [reps]"""
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
              (200, 50, 60, 2, 8, 0.02, 2.0), (128, 50, 128, 2, 9, 0.02, 2.0), (256, 40, 8, 2, 10, 0.02, 2.0),
              # the per-step exchange of the register-resident and matrix-core kernels: C3's, C4's and C2's shapes, N = 32, N = 8
              (64, 300, 256, 2, 11, 0.05, 5.0), (64, 200, 1024, 2, 12, 0.05, 5.0), (16, 500, 32, 1, 13, 0.1, 1.0), (32, 300, 256, 2, 14, 0.05, 5.0),
              (8, 300, 512, 1, 15, 0.1, 2.0), (64, 200, 2048, 2, 16, 0.05, 5.0)]
    total = 0
    for shape in shapes:
        p = wl.dense_problem(*shape)
        K = p["psi0"].shape[0]
        small = shape[0] <= 64
        optsets = ([{"STAMPS": 1}, {"STAMPS": 1, "POLL_DELAY": -1}, {"STAMPS": 1, "POLL_DELAY": 8}, {"STAMPS": 1, "HERM": 0}] if small else
                   [{"STAMPS": 1}, {"STAMPS": 1, "RS_LMAX1": 0}, {"STAMPS": 1, "XCD_GROUP": 0}, {"STAMPS": 1, "RS_PAIR": 0, "RS_LMAX1": 0}])
        for opts in optsets:
            events = 0
            xevents = 0
            first = None
            for r in range(reps):
                with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K, options=opts) as eng:
                    eng.set_initial_states(p["psi0"])
                    psiT = eng.forward(p["eps"])
                    _, _, chiT = wl.chi_and_JT(p["functional"], psiT, p["tgt"])
                    eng.iteration(p["eps"], chiT, p["S"], p["lam"])
                    st = eng.debug_stamps()
                    arr = eng.debug_arrivals(1)[0].ravel()      # debug words 8..15: [0..3] last all-gather event, [4] = exchange count, [5] its last event
                    plan = eng.plan()
                    if st[4] and not small:                     # (word 4 is a time stamp in the N <= 64 kernels)
                        events += st[4]
                        if first is None:
                            first = arr[:4]
                    xevents += int(arr[4])
            total += events + xevents
            print("%r %r: stale all-gather elements accepted: %d, stale partial sums accepted in the per-step exchange: %d, in %d runs %s | %s"
                  % (shape, opts, events, xevents, reps, "" if first is None else "last event (block, counter, element, seq re|im): %r" % (list(map(int, first)),),
                     plan.split(": update sweep = ")[1][:60]), flush=True)
    print("TOTAL stale values accepted:", total)
    # positive control: with RS_FAULT=1 workgroup 1 of a diagnostics build labels its partial sum of every 16th step as four steps old
    p = wl.dense_problem(64, 100, 256, 2, 11, 0.05, 5.0)
    with kr.HipEngine(p["H0"], p["Hc"], p["tlist"], 256, options={"STAMPS": 1, "RS_FAULT": 1, "POLL_DELAY": 8}) as eng:
        eng.set_initial_states(p["psi0"])
        psiT = eng.forward(p["eps"])
        _, _, chiT = wl.chi_and_JT(p["functional"], psiT, p["tgt"])
        eng.iteration(p["eps"], chiT, p["S"], p["lam"])
        n = int(eng.debug_arrivals(1)[0].ravel()[4])
    print("positive control (one mislabelled slot in 6 of 100 steps, 256 pollers): %d events counted -- the detector %s" % (n, "works" if n > 0 else "DOES NOT FIRE"))


if __name__ == "__main__":
    main()
