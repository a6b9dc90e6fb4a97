# KrotovHIP.jl -- thin ccall shim that puts libkrotov_hip.so behind Krotov.jl's two hot functions.
#
# UNTESTED: there is no Julia toolchain in the build image.  This file shows the reference-side binding a
# maintainer would add (see INTEGRATION.md); it follows include/krotov_hip.h one to one.
#
# What it replaces in /root/reference/src/optimize.jl:
#   krotov_initial_fw_prop!(eps0, phi_k, k, wrk)   :247-265  -> one kr_forward for all k
#   krotov_iteration(wrk, eps_i, eps_ip1)          :279-371  -> chi(T) in Julia, then one kr_iteration
# Everything else (KrotovWrk, KrotovResult, callbacks, convergence, print_iters) stays Julia.
module KrotovHIP

using LinearAlgebra
import Krotov
import QuantumControl
using QuantumControl.QuantumPropagators.Controls: get_controls, discretize_on_midpoints
using QuantumControl.QuantumPropagators.Generators: Operator

const LIB = get(ENV, "KROTOV_HIP_LIB", joinpath(@__DIR__, "..", "krotov.jl_amd", "lib", "libkrotov_hip.so"))
const KR_ABI_VERSION = Int32(1)

# struct kr_config (include/krotov_hip.h)
struct KrConfig
    abi_version::Int32
    N::Int32
    NT::Int32
    K::Int32
    L::Int32
    G::Int32
    traj_generator::Ptr{Int32}
    tlist::Ptr{Float64}
    H0::Ptr{ComplexF64}
    Hc::Ptr{ComplexF64}
    hc_present::Ptr{UInt8}
    flags::UInt32
    prop_tol::Float64
    device::Int32
    rank::Int32
    world::Int32
    K_global::Int32
    reserved::NTuple{8,Int32}
end

mutable struct Engine
    handle::Ptr{Cvoid}
    N::Int
    NT::Int
    K::Int
    L::Int
end

function check(h::Ptr{Cvoid}, rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:kr_last_error, LIB), Cstring, (Ptr{Cvoid},), h))
    error("libkrotov_hip error $rc: $msg")   # an ErrorException: optimize.jl:218-226 captures it
end

"""Build the engine from what KrotovWrk already holds (workspace.jl:70-125): dense H0 / dH/deps_l,
tlist, initial states.  Requires a generator shared by all trajectories (G = 1)."""
function Engine(wrk)
    traj = wrk.trajectories[1]
    tlist = wrk.result.tlist
    N = length(traj.initial_state)
    K = length(wrk.trajectories)
    L = length(wrk.controls)
    H0 = Matrix{ComplexF64}(Array(traj.generator.ops[1]))           # drift term (Julia memory = column-major)
    Hc = Array{ComplexF64}(undef, N, N, L)
    present = ones(UInt8, L)
    for l = 1:L
        mu = wrk.control_derivs[1][l]
        if isnothing(mu)
            present[l] = 0
            Hc[:, :, l] .= 0
        else
            Hc[:, :, l] .= Array(mu)                                  # static derivative (optimize.jl:275-276)
        end
    end
    cfg = KrConfig(KR_ABI_VERSION, N, length(tlist) - 1, K, L, 1, C_NULL, pointer(tlist), pointer(H0),
                   pointer(Hc), pointer(present), UInt32(1), 0.0, -1, 0, 1, 0, ntuple(_ -> Int32(0), 8))
    href = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve tlist H0 Hc present begin
        rc = ccall((:kr_create, LIB), Cint, (Ref{KrConfig}, Ref{Ptr{Cvoid}}), cfg, href)
    end
    check(Ptr{Cvoid}(C_NULL), rc)
    eng = Engine(href[], N, length(tlist) - 1, K, L)
    psi0 = reduce(hcat, [Vector{ComplexF64}(t.initial_state) for t in wrk.trajectories])   # N x K = [K][N] in C
    GC.@preserve psi0 check(eng.handle, ccall((:kr_set_initial_states, LIB), Cint, (Ptr{Cvoid}, Ptr{ComplexF64}), eng.handle, psi0))
    finalizer(e -> ccall((:kr_destroy, LIB), Cint, (Ptr{Cvoid},), e.handle), eng)
    return eng
end

pulses_matrix(eps) = permutedims(reduce(hcat, eps))   # L vectors of N_T -> (N_T x L)' ; memory [L][N_T] needs N_T fastest
function pack_pulses(eps)
    NT = length(eps[1]); L = length(eps)
    out = Matrix{Float64}(undef, NT, L)               # column l contiguous = [L][N_T] in C order
    for l = 1:L; out[:, l] .= eps[l]; end
    return out
end

"""All-k replacement of `krotov_initial_fw_prop!` (optimize.jl:247-265)."""
function initial_fw_prop!(eng::Engine, eps0, wrk)
    e = pack_pulses(eps0)
    psiT = Matrix{ComplexF64}(undef, eng.N, eng.K)
    GC.@preserve e psiT check(eng.handle, ccall((:kr_forward, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{ComplexF64}), eng.handle, e, psiT))
    for k = 1:eng.K
        copyto!(wrk.fw_propagators[k].state, view(psiT, :, k))       # what update_result! reads (optimize.jl:379)
    end
end

"""Replacement of `krotov_iteration` (optimize.jl:279-371)."""
function iteration!(eng::Engine, wrk, eps_i, eps_ip1)
    chi = wrk.kwargs[:chi]
    Psi = [p.state for p in wrk.fw_propagators]
    X = wrk.chi_takes_tau ? chi(Psi, wrk.trajectories; tau = wrk.result.tau_vals) : chi(Psi, wrk.trajectories)
    chiT = reduce(hcat, [Vector{ComplexF64}(x) for x in X])
    e = pack_pulses(eps_i)
    S = pack_pulses(wrk.update_shapes)
    lam = Vector{Float64}(wrk.lambda_vals)
    e_new = similar(e)
    psiT = Matrix{ComplexF64}(undef, eng.N, eng.K)
    GC.@preserve e chiT S lam e_new psiT check(eng.handle,
        ccall((:kr_iteration, LIB), Cint,
              (Ptr{Cvoid}, Ptr{Float64}, Ptr{ComplexF64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{ComplexF64}),
              eng.handle, e, chiT, S, lam, e_new, wrk.g_a_int, psiT))
    for l = 1:eng.L
        eps_ip1[l] .= view(e_new, :, l)                               # in place: callbacks hold these vectors
    end
    for k = 1:eng.K
        copyto!(wrk.fw_propagators[k].state, view(psiT, :, k))
    end
end

"""Callback inspection of the device storages (workspace.jl:24): which = :fw or :bw, n = 1..N_T+1."""
function read_storage(eng::Engine, which::Symbol, k::Int, n::Int)
    out = Vector{ComplexF64}(undef, eng.N)
    GC.@preserve out check(eng.handle, ccall((:kr_read_storage, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{ComplexF64}),
                                             eng.handle, which == :bw ? 1 : 0, k - 1, n - 1, out))
    return out
end

end # module
