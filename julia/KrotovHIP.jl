# KrotovHIP.jl -- thin ccall shim that puts libkrotov_hip.so behind Krotov.jl's two hot functions.
#
# UNTESTED: there is no Julia toolchain in the build image.  This file shows the reference-side binding a
# maintainer would add (see INTEGRATION.md); it follows include/krotov_hip.h (ABI v2) one to one.
#
# What it replaces in /root/reference/src/optimize.jl:
#   krotov_initial_fw_prop!(eps0, phi_k, k, wrk)   :247-265  -> one kr_forward for all k
#   krotov_iteration(wrk, eps_i, eps_ip1)          :279-371  -> chi(T) in Julia, then one kr_iteration
# Everything else (KrotovWrk, KrotovResult, callbacks, convergence, print_iters) stays Julia.
#
# Generators: trajectories may share one generator or carry their own (control_derivs[k][l], workspace.jl:74): distinct
# generator OBJECTS become the G generators of the engine, traj_generator maps k -> g.  The drift is the generator evaluated
# with every control at zero, the control operators are the (static) control derivatives -- i.e. linear controls; a
# derivative that is not a constant operator (optimize.jl:268-272) is refused here (kr_config.control_poly covers polynomial
# amplitudes f_l(eps_l), which has no counterpart in QuantumControl's generator types to extract automatically).
module KrotovHIP

using LinearAlgebra
import Krotov
import QuantumControl
using QuantumControl.QuantumPropagators.Controls: get_controls, evaluate

const LIB = get(ENV, "KROTOV_HIP_LIB", joinpath(@__DIR__, "..", "krotov.jl_amd", "lib", "libkrotov_hip.so"))
const KR_ABI_VERSION = Int32(2)

# struct kr_config (include/krotov_hip.h, ABI v2) -- field for field
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
    exchange::Int32              # 0 auto, 1 mailbox, 2 RCCL per step, 3 host callback
    prop_method::Int32           # 0 auto, 1 Taylor, 2 Chebyshev, 3 Newton
    max_workgroups::Int32
    n_devices::Int32             # > 1: this ONE handle drives all listed GPUs (threads inside the library)
    devices::NTuple{8,Int32}
    control_poly_degree::Int32
    pad_::Int32
    control_poly::Ptr{Float64}
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

dense(op) = Matrix{ComplexF64}(Array(op))

"""Build the engine from what KrotovWrk already holds (workspace.jl:70-125): dense drift / dH/deps_l per distinct
generator, tlist, initial states.  `devices`: HIP ordinals of the GPUs this one handle should drive (default: the
current device); `prop_method`: :auto, :taylor, :cheby, :newton."""
function Engine(wrk; devices::Vector{Int}=Int[], prop_method::Symbol=:auto, prop_tol::Float64=0.0)
    trajs = wrk.trajectories
    tlist = Vector{Float64}(wrk.result.tlist)
    controls = wrk.controls
    N = length(trajs[1].initial_state)
    K = length(trajs)
    L = length(controls)
    # distinct generators (by identity) -> G, k -> g
    gens = Any[]
    tg = Vector{Int32}(undef, K)
    for (k, t) in enumerate(trajs)
        g = findfirst(x -> x === t.generator, gens)
        if g === nothing
            push!(gens, t.generator)
            g = length(gens)
        end
        tg[k] = g - 1
    end
    G = length(gens)
    zero_vals = IdDict(c => 0.0 for c in controls)
    H0 = Array{ComplexF64}(undef, N, N, G)
    Hc = Array{ComplexF64}(undef, N, N, L, G)
    present = ones(UInt8, L, G)
    for g = 1:G
        k = findfirst(==(Int32(g - 1)), tg)
        H0[:, :, g] .= dense(evaluate(gens[g], tlist, 1; vals_dict=zero_vals))      # drift: all controls at zero
        for l = 1:L
            mu = wrk.control_derivs[k][l]
            if isnothing(mu)
                present[l, g] = 0                                                      # optimize.jl:344
                Hc[:, :, l, g] .= 0
            elseif mu isa AbstractMatrix || !applicable(evaluate, mu, tlist, 1)
                Hc[:, :, l, g] .= dense(mu)                                            # static derivative, optimize.jl:275-276
            else
                error("KrotovHIP: control derivative $l of trajectory $k depends on the control value (non-linear control); " *
                      "not extracted automatically -- see kr_config.control_poly")
            end
        end
    end
    method = Dict(:auto => 0, :taylor => 1, :cheby => 2, :newton => 3)[prop_method]
    devs = ntuple(i -> Int32(i <= length(devices) ? devices[i] : 0), 8)
    cfg = KrConfig(KR_ABI_VERSION, N, length(tlist) - 1, K, L, G, G > 1 ? pointer(tg) : C_NULL, pointer(tlist), pointer(H0),
                   pointer(Hc), pointer(present), UInt32(1), prop_tol, -1, 0, 1, 0,
                   0, method, 0, Int32(length(devices)), devs, 0, 0, C_NULL, ntuple(_ -> Int32(0), 8))
    href = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve tg tlist H0 Hc present begin
        rc = ccall((:kr_create, LIB), Cint, (Ref{KrConfig}, Ref{Ptr{Cvoid}}), cfg, href)
    end
    check(Ptr{Cvoid}(C_NULL), rc)
    eng = Engine(href[], N, length(tlist) - 1, K, L)
    psi0 = reduce(hcat, [Vector{ComplexF64}(t.initial_state) for t in trajs])   # N x K = [K][N] in C
    GC.@preserve psi0 check(eng.handle, ccall((:kr_set_initial_states, LIB), Cint, (Ptr{Cvoid}, Ptr{ComplexF64}), eng.handle, psi0))
    finalizer(e -> ccall((:kr_destroy, LIB), Cint, (Ptr{Cvoid},), e.handle), eng)
    return eng
end

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

"""An MPI all-reduce as the per-step exchange (kr_config.exchange = 3): `fn(buf::Vector{Float64})` must sum in place
over all ranks.  The @cfunction must stay alive as long as the engine does."""
function set_exchange_callback!(eng::Engine, fn)
    tramp = (ctx::Ptr{Cvoid}, buf::Ptr{Float64}, count::Cint) -> begin
        try
            fn(unsafe_wrap(Array, buf, Int(count)))
            return Cint(0)
        catch
            return Cint(1)       # never throw across the C boundary
        end
    end
    cfn = @cfunction($tramp, Cint, (Ptr{Cvoid}, Ptr{Float64}, Cint))
    check(eng.handle, ccall((:kr_set_exchange_callback, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), eng.handle, cfn, C_NULL))
    return cfn
end

end # module
