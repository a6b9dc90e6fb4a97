import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import krotov_jl_amd as kr, numpy as np
from krotov_jl_amd import workloads as wl
p = wl.dense_problem(12, 5, 3, 1, 1, 0.1, 1.0)
e = kr.HipEngine(p["H0"], p["Hc"], p["tlist"], 3)
e.set_initial_states(p["psi0"]); e.forward(p["eps"])
import torch
print("torch sees GPU after libkrotov_hip:", torch.cuda.is_available())
try:
    torch.cuda.init(); print(torch.zeros(2).cuda())
except Exception as ex:
    print("torch init error:", ex)
