"""Developer tool: time the three sweep kernels (HIP events inside the library) on a config."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import krotov_jl_amd as kr
from krotov_jl_amd import workloads as wl

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
p = wl.config(cfg)
K, N = p["psi0"].shape
L, NT = p["eps"].shape
kw = {}
if os.environ.get("TIME_SWEEPS_POLY"):   # non-linear controls: f_l(x) = x + x^2 / 2 (+ x^3 for the second control)
    kw["control_poly"] = np.array([[0.0, 1.0, 0.5, 0.0], [0.0, 1.0, 0.5, 1.0], [0.0, 1.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0]][:L])
eng = kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K, **kw)
eng.set_initial_states(p["psi0"])
psiT = eng.forward(p["eps"])
_, JT, chiT = wl.chi_and_JT(p["functional"], psiT, p["tgt"])
eng.iteration(p["eps"], chiT, p["S"], p["lam"])
eng.enable_timing(True)
t0 = time.perf_counter()
for _ in range(reps):
    eng.forward(p["eps"])
    eps_new, ga, psiT1 = eng.iteration(p["eps"], chiT, p["S"], p["lam"])
wall = (time.perf_counter() - t0) / reps
tm = eng.timings_ms()
print(cfg, eng.info())
for k, (ms, n) in tm.items():
    print("  %-8s %8.3f ms/launch  %6.2f us/step" % (k, ms / max(n, 1), 1e3 * ms / max(n, 1) / NT))
print("  wall per (forward + iteration) incl. host copies: %.3f ms" % (1e3 * wall))
print("  |d eps|max = %.3e  J_T = %.6f" % (np.max(np.abs(eps_new - p["eps"])), JT))
st = eng.debug_stamps()
if st[4]:
    cyc = sum(st[:4]); us = st[4] / 100.0
    print("  stamps (wg 0): matvec+dot %.0f  exchange %.0f  taylor %.0f  store %.0f cycles/step; clock %.2f GHz; %.2f us/step"
          % (st[0] / NT, st[1] / NT, st[2] / NT, st[3] / NT, cyc / us / 1e3, us / NT))
if st[5] and not eng.info().get("rowsplit_R", 0):
    print("  taylor phase split (wg 0): setup until first round %.0f, rounds %.0f, finalize %.0f cycles/step"
          % (st[5] / NT, (st[6] - st[5]) / NT, (st[2] - st[6]) / NT))
if eng.info().get("rowsplit_R", 0) and st[7]:
    print("  gather poll: %.0f cycles/round, %.2f passes/round" % (st[5] / st[7], st[6] / st[7]))
    print("  row-split stamps (wg 0, last sweep): stream %.0f cycles/step, row-dot %.0f and gather %.0f cycles/round, exchange %.0f cycles/step, %d rounds"
          % (st[0] / NT, st[1] / st[7], st[2] / st[7], st[3] / NT, st[7]))
