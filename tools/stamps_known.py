"""Developer tool: phase stamps (KROTOV_HIP_STAMPS=1, workgroup 0) of the KNOWN-pulse sweep of a config -- the last launch before the
stamps are read is a forward sweep.   KROTOV_HIP_STAMPS=1 python tools/stamps_known.py C5"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import krotov_jl_amd as kr
from krotov_jl_amd import workloads as wl

cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
p = wl.config(cfg)
K, N = p["psi0"].shape
L, NT = p["eps"].shape
eng = kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K)
eng.set_initial_states(p["psi0"])
eng.forward(p["eps"])
eng.enable_timing(True)
for _ in range(3):
    eng.forward(p["eps"])
tm = eng.timings_ms()
st = eng.debug_stamps()
info = eng.info()
print(cfg, "known-pulse sweep: %.3f ms per launch, %.2f us per step" % (tm["forward"][0] / tm["forward"][1], 1e3 * tm["forward"][0] / tm["forward"][1] / NT))
if info.get("rowsplit_R", 0) and st[7]:
    print("  row-split stamps (wg 0, last sweep): stream %.0f cycles/step, row-dot %.0f and gather-phase %.0f cycles per trajectory-round, %d rounds (%.2f per step and trajectory); "
          "poll %.0f cycles per round, %.2f passes per round" % (st[0] / NT, st[1] / st[7], st[2] / st[7], st[7], st[7] / NT / 2.0, st[5] / st[7], st[6] / st[7]))
    print("  per step: %.0f cycles = stream %.0f + rounds %.0f" % ((st[0] + st[1] + st[2] + st[3]) / NT, st[0] / NT, (st[1] + st[2]) / NT))
print(eng.plan())
