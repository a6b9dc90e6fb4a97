"""Diagnostic (needs a library whose kr_rs256.hip was compiled with -DKR_PAIR_PROBE): where a phase of the
row-split pair kernel spends its cycles -- deposit + barrier, wait for the early LDS-DMA pass, validity check, second barrier.
    KROTOV_HIP_LIB=.../libkrotov_hip_probe.so KROTOV_HIP_STAMPS=1 python tools/pair_probe.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import krotov_jl_amd as kr
from krotov_jl_amd import workloads as wl
p = wl.config("C5")
K, N = p["psi0"].shape
eng = kr.HipEngine(p["H0"], p["Hc"], p["tlist"], K)
eng.set_initial_states(p["psi0"])
psiT = eng.forward(p["eps"])
st = eng.debug_stamps()
n = st[7]
print("known sweep: rounds", n, "per round: rowdot %.0f phase %.0f | A(deposit+B1) %.0f  B(vmcnt wait) %.0f  C(check) %.0f  rest(B2..) %.0f" % (st[1]/n, st[2]/n, st[4]/n, st[5]/n, st[6]/n, (st[2]-st[4]-st[5]-st[6])/n))
