"""MI355X-native engine for the hot path of Krotov's method (drop-in for the propagation /
pulse-update loop of JuliaQuantumControl/Krotov.jl, /root/reference/src/optimize.jl:247-371).

The compute path is libkrotov_hip.so (hand-written HIP for gfx950) behind the C ABI in
include/krotov_hip.h.  This package is the host-side mirror of the reference's interface.
There is no CPU fallback.
"""
from . import _capi
from ._capi import KrotovHipError
from .engine import HipEngine
from . import workloads, distributed, optimize as _optimize_mod
from .optimize import (ControlProblem, Trajectory, hamiltonian, liouvillian, optimize, optimize_krotov, KrotovResult, KrotovWrk,
                       J_T_sm, J_T_ss, J_T_re, make_chi, discretize, discretize_on_midpoints, flattop)

__all__ = ["HipEngine", "KrotovHipError", "_capi"]
