"""Diagnostic: which combination of (our dlopen'ed RCCL, torch import, communicator destroy) upsets process exit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
import numpy as np
import krotov_jl_amd as kr
from krotov_jl_amd import workloads as wl, _capi as capi
if "torch_first" in mode:
    import torch
p = wl.dense_problem(8, 5, 4, 1, 1, 0.1, 1.0)
ex = capi.KR_EXCHANGE_RCCL if "rccl" in mode or "init" in mode else (capi.KR_EXCHANGE_CALLBACK if "callback" in mode else 0)
eng = kr.HipEngine(p["H0"], p["Hc"], p["tlist"], 4, exchange=ex)
print(eng.info(), flush=True)
if "init" in mode and "noinit" not in mode:
    eng.rccl_init(kr.HipEngine.rccl_unique_id())
eng.set_initial_states(p["psi0"])
psiT = eng.forward(p["eps"])
eng.iteration(p["eps"], p["tgt"], p["S"], p["lam"])
if "noclose" not in mode:
    eng.close()
if "torch_after" in mode:
    import torch
    if "cuda" in mode:
        torch.zeros(4, device="cuda")
print(mode, "done", flush=True)
