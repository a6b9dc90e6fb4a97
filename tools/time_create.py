"""Developer tool: wall time of kr_create (packs, norm bounds, calibration) per configuration:
   python tools/time_create.py [CONFIG ...]      (default C3 C4 C5 D:512,10,4,2 D:1024,10,2,2)"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import krotov_jl_amd as kr
from krotov_jl_amd import workloads as wl

for cfg in (sys.argv[1:] or ["C3", "C4", "C5", "D:512,10,4,2", "D:1024,10,2,2"]):
    p = wl.config(cfg)
    K = p["psi0"].shape[0]
    for cal in ["0", "1"]:
        os.environ["KROTOV_HIP_CALIBRATE"] = cal
        t0 = time.perf_counter()
        eng = kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K)
        t1 = time.perf_counter()
        print(cfg, "calibrate", cal, "create %.1f ms" % (1e3 * (t1 - t0)), eng.info()["exchange_poll_delay"], flush=True)
        del eng
