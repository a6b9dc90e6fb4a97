#!/usr/bin/env python3
"""Diagnostic (needs a library built with -DKR_ARRIVALS, see tools/build_diag.sh): when do the workgroups publish their partial sums
and complete the exchange in the middle of an update sweep?  Separates arrival skew from the exchange itself.
    KROTOV_HIP_LIB=.../libkrotov_hip_diag.so KROTOV_HIP_STAMPS=1 python tools/exchange_arrivals.py C3"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import krotov_jl_amd as kr
from krotov_jl_amd import workloads as wl

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
p = wl.config(cfg)
K, N = p["psi0"].shape
eng = kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K)
eng.set_initial_states(p["psi0"])
psiT = eng.forward(p["eps"])
_, _, chiT = wl.chi_and_JT(p["functional"], psiT, p["tgt"])
for _ in range(3):
    eng.iteration(p["eps"], chiT, p["S"], p["lam"])
info = eng.info()
P = info["workgroups"]
a = eng.debug_arrivals(P).astype(np.int64)
print(cfg, {k: info[k] for k in ("workgroups", "exchange_poll_delay", "hermitian_controls_path", "matrix_core_sweep")})
tick = 10.0  # ns
for e in range(2):
    pub, wend, done, passes = a[:, e, 0], a[:, e, 1], a[:, e, 2], (a[:, e, 3] & 255).copy()
    last_pass, wend_to_sum = (a[:, e, 3] >> 8) & 0xFFFFFF, (a[:, e, 3] >> 32) & 0xFFFFFF
    early = passes >= 100
    passes[early] -= 101   # an early hit needs no register pass
    print("   early pass (LDS-DMA at publish time) hit in %d of %d workgroups; by XCD: %s" % (early.sum(), P, " ".join("%2d" % early[np.arange(P) % 8 == x].sum() for x in range(8))))
    if not pub.any():
        print("no arrival stamps: library not built with -DKR_ARRIVALS?")
        break
    t0 = pub.min()
    print("exchange %d: publish spread %.0f ns (median %+.0f, last %+.0f); window work %.0f ns (median); completion after LAST publish: "
          "min %.0f median %.0f max %.0f ns; after OWN publish: median %.0f ns; poll passes median %.1f max %d"
          % (e, (pub.max() - t0) * tick, (np.median(pub) - t0) * tick, (pub.max() - t0) * tick, np.median(wend - pub) * tick,
             (done.min() - pub.max()) * tick, (np.median(done) - pub.max()) * tick, (done.max() - pub.max()) * tick,
             np.median(done - pub) * tick, np.median(passes) + 1, passes.max() + 1))
    xcd = np.arange(P) % 8
    print("   publish offset by XCD (ns, median): " + " ".join("%4.0f" % ((np.median(pub[xcd == x]) - t0) * tick) for x in range(8)))
    print("   window work by XCD (ns, median):        " + " ".join("%4.0f" % (np.median((wend - pub)[xcd == x]) * tick) for x in range(8)))
    print("   poll (window end -> done) by XCD (ns):   " + " ".join("%4.0f" % (np.median((done - wend)[xcd == x]) * tick) for x in range(8)))
    print("   poll passes by XCD (median):             " + " ".join("%4.1f" % (np.median(passes[xcd == x]) + 1) for x in range(8)))
    print("   LAST pass, loads + accumulate (cycles):  " + " ".join("%4.0f" % np.median(last_pass[xcd == x]) for x in range(8)))
    print("   window end -> sums done (cycles):        " + " ".join("%4.0f" % np.median(wend_to_sum[xcd == x]) for x in range(8)))
    print("   completion offset by XCD (ns, median): " + " ".join("%4.0f" % ((np.median(done[xcd == x]) - t0) * tick) for x in range(8)))
if a[:, 1, 0].any():
    xcd = np.arange(P) % 8
    print("own cycle outside the exchange (done of exchange 0 -> publish of exchange 1) by XCD (ns, median): "
          + " ".join("%4.0f" % (np.median((a[:, 1, 0] - a[:, 0, 2])[xcd == x]) * tick) for x in range(8)))
    print("step length (publish to publish): median %.0f ns" % (np.median(a[:, 1, 0] - a[:, 0, 0]) * tick))
