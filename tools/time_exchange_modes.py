"""Developer tool: the update sweep of a config under the three exchange carriers on ONE GPU (world = 1, so the exchange
itself is trivial: what is timed is the per-step launch structure the RCCL / callback baselines need, against the one
persistent launch of the mailbox path).  Usage: tools/time_exchange_modes.py [C3] [reps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import krotov_jl_amd as kr
from krotov_jl_amd import workloads as wl, _capi as capi

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
p = wl.config(cfg)
K, N = p["psi0"].shape
L, NT = p["eps"].shape
ref = None
for name, ex in (("mailbox (persistent kernel)", capi.KR_EXCHANGE_MAILBOX), ("rccl (ncclAllReduce per step, world-1 communicator)", capi.KR_EXCHANGE_RCCL),
                 ("callback (host function per step)", capi.KR_EXCHANGE_CALLBACK)):
    eng = kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K, exchange=ex)
    if ex == capi.KR_EXCHANGE_RCCL:
        eng.rccl_init(kr.HipEngine.rccl_unique_id())
    if ex == capi.KR_EXCHANGE_CALLBACK:
        eng.set_exchange_callback(lambda buf: None)
    eng.set_initial_states(p["psi0"])
    psiT = eng.forward(p["eps"])
    _, _, chiT = wl.chi_and_JT(p["functional"], psiT, p["tgt"])
    out = eng.iteration(p["eps"], chiT, p["S"], p["lam"])
    eng.enable_timing(True)
    for _ in range(reps):
        out = eng.iteration(p["eps"], chiT, p["S"], p["lam"])
    tm = eng.timings_ms()
    ms = tm["update"][0] / max(tm["update"][1], 1)
    if ref is None:
        ref = out
    print("%s %-52s update sweep %9.3f ms = %7.2f us/step   max|eps - eps_mailbox| = %.2e"
          % (cfg, name, ms, 1e3 * ms / NT, np.max(np.abs(out[0] - ref[0]))), flush=True)
    eng.close()
