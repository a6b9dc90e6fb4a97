# KrotovHIP.jl -- puts libkrotov_hip.so behind Krotov.jl's two hot functions WITHOUT patching Krotov.jl.
#
# UNTESTED: there is no Julia toolchain in the build image (the runnable stand-in for these `ccall`s is the C99 consumer
# tests/abi_consumer.c, which drives the same entry points in the same order).  This file follows include/krotov_hip.h
# (ABI v3) field for field; INTEGRATION.md explains every hook below against the reference's file:line.
#
#     using Krotov, QuantumControl, KrotovHIP
#     problem = ControlProblem(trajectories, tlist; prop_method = KrotovHIP, J_T = J_T_sm, iter_stop = 5, ...)
#     res = optimize(problem; method = Krotov)          # the public surface is unchanged
#
# How it hooks in (nothing in /root/reference/src is edited):
#
#  1. `prop_method = KrotovHIP` rides the reference's own `prop_`-prefixed kwargs (/root/reference/src/optimize.jl:137-153).
#     `KrotovWrk` builds its fw/bw propagators at /root/reference/src/workspace.jl:136-161 through `init_prop_trajectory` ->
#     `QuantumPropagators.init_prop(state, generator, tlist, Val(:KrotovHIP); ...)` [UPSTREAM-RECALLED: a module given as
#     `method` is turned into `Val(nameof(method))`].  The method defined below returns a `HIPPropagator`: a light placeholder
#     that only carries what the optimizer reads and writes on a propagator outside the hot functions -- `.state`
#     (optimize.jl:297,379) and `.parameters` (:249,305,322).  No device memory exists yet at that point.
#  2. The engine is born lazily, at the FIRST hot call on a finished `KrotovWrk` (it needs `control_derivs`, `update_shapes`,
#     `lambda_vals`, which only exist once the constructor at workspace.jl:65-200 has returned), and is kept in a
#     `WeakKeyDict{KrotovWrk,Engine}`: it dies with the workspace (finalizer -> kr_destroy).
#  3. Two METHODS are added to the reference's generic functions, more specific in the `wrk` argument than the untyped
#     originals, so dispatch picks them for every `KrotovWrk`; they look the workspace up and fall through to the original
#     (`invoke` with the untyped signature) unless its propagators are `HIPPropagator`s:
#         Krotov.krotov_initial_fw_prop!(ϵ⁽⁰⁾, ϕₖ, k, wrk::KrotovWrk)   optimize.jl:247-265   -> ONE kr_forward for all k
#         Krotov.krotov_iteration(wrk::KrotovWrk, ϵ⁽ⁱ⁾, ϵ⁽ⁱ⁺¹⁾)          optimize.jl:279-371   -> chi(T) in Julia, ONE kr_iteration
#     Everything else (KrotovWrk, KrotovResult, update_result!, callbacks, convergence, print_iters, finalize_result!) runs
#     as it is.
#
# States that do not support in-place updates (SVector: /root/reference/test/test_tls_optimization.jl:73-96) are RE-BOUND,
# as the reference does at optimize.jl:331-335; mutable ones are overwritten in place.
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
import QuantumControl.QuantumPropagators
const QP = QuantumControl.QuantumPropagators
using QuantumControl.QuantumPropagators.Controls: get_controls, evaluate
using QuantumControl.QuantumPropagators.Interfaces: supports_inplace

const LIB = get(ENV, "KROTOV_HIP_LIB", joinpath(@__DIR__, "..", "krotov.jl_amd", "lib", "libkrotov_hip.so"))
const KR_ABI_VERSION = Int32(3)

# struct kr_config (include/krotov_hip.h, ABI v3) -- field for field (tests/test_julia_shim_static.py compares it with the ctypes mirror)
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
    spin_timeout_ms::Int32       # bound of the in-kernel spins; 0 = default 4000 ms
    reserved0::Int32
    options::Cstring             # "NAME=integer;..." schedule / diagnostics options (INTEGRATION.md), or C_NULL
    reserved::NTuple{4,Int32}
end

# ---- 1. the placeholder propagator ------------------------------------------------------------------------------------
"""What `init_prop(...; method = KrotovHIP)` returns: the fields the optimizer touches on a propagator OUTSIDE the two hot
functions (`state`: optimize.jl:297,379; `parameters`: :249,305,322).  `prop_step!` / `reinit_prop!` on it are errors: whole
sweeps belong to the engine, there is no per-step entry point (see include/krotov_hip.h for why)."""
mutable struct HIPPropagator
    generator::Any
    state::Any
    tlist::Vector{Float64}
    parameters::Any
    backward::Bool
    kwargs::Dict{Symbol,Any}      # engine options given as prop_* kwargs: devices, hip_prop_method, prop_tol, spin_timeout_ms
    t::Float64                    # time of `state` while per-step callbacks are replayed (what a callback reads besides `state`)
    i::Int                        # ... and its index into tlist
end

function QP.init_prop(state, generator, tlist, ::Val{:KrotovHIP}; backward = false, parameters = nothing, kwargs...)
    tl = Vector{Float64}(tlist)
    return HIPPropagator(generator, state, tl, parameters, backward, Dict{Symbol,Any}(kwargs), backward ? tl[end] : tl[1], backward ? length(tl) : 1)
end
QP.prop_step!(::HIPPropagator) = error("KrotovHIP: the engine owns whole sweeps; there is no per-step propagation (kr_read_storage inspects stored states)")
QP.reinit_prop!(::HIPPropagator, state; kwargs...) = error("KrotovHIP: reinit_prop! is part of the sweeps the engine owns")

# ---- 2. the engine ----------------------------------------------------------------------------------------------------
mutable struct Engine
    handle::Ptr{Cvoid}
    N::Int
    NT::Int
    K::Int
    L::Int
    initial_done::Bool            # krotov_initial_fw_prop! is called once PER TRAJECTORY (optimize.jl:182-184): the first call runs all k
    lock::ReentrantLock           # ... possibly from several threads (use_threads = true)
end

function check(h::Ptr{Cvoid}, rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:kr_last_error, LIB), Cstring, (Ptr{Cvoid},), h))
    error("libkrotov_hip error $rc: $msg")   # an ErrorException: optimize.jl:218-226 captures it into result.message
end

dense(op) = Matrix{ComplexF64}(Array(op))

"""Build the engine from what a finished KrotovWrk holds (workspace.jl:70-125): dense drift / dH/deps_l per distinct
generator, tlist, initial states.  `devices`: HIP ordinals of the GPUs this one handle should drive (default: the current
device); `prop_method`: :auto, :taylor, :cheby, :newton."""
function Engine(wrk; devices::Vector{Int} = Int[], prop_method::Symbol = :auto, prop_tol::Float64 = 0.0, spin_timeout_ms::Int = 0, options::String = "")
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
        H0[:, :, g] .= dense(evaluate(gens[g], tlist, 1; vals_dict = zero_vals))      # drift: all controls at zero
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
                   0, method, 0, Int32(length(devices)), devs, 0, 0, C_NULL, Int32(spin_timeout_ms), Int32(0), Cstring(isempty(options) ? C_NULL : pointer(options)), ntuple(_ -> Int32(0), 4))
    href = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve tg tlist H0 Hc present options begin
        rc = ccall((:kr_create, LIB), Cint, (Ref{KrConfig}, Ref{Ptr{Cvoid}}), cfg, href)
    end
    check(Ptr{Cvoid}(C_NULL), rc)
    eng = Engine(href[], N, length(tlist) - 1, K, L, false, ReentrantLock())
    psi0 = reduce(hcat, [Vector{ComplexF64}(t.initial_state) for t in trajs])   # N x K = [K][N] in C
    GC.@preserve psi0 check(eng.handle, ccall((:kr_set_initial_states, LIB), Cint, (Ptr{Cvoid}, Ptr{ComplexF64}), eng.handle, psi0))
    finalizer(e -> ccall((:kr_destroy, LIB), Cint, (Ptr{Cvoid},), e.handle), eng)
    return eng
end

# One engine per workspace, created at the first hot call, gone with the workspace.
const ENGINES = WeakKeyDict{Krotov.KrotovWrk,Engine}()
const ENGINES_LOCK = ReentrantLock()

uses_hip(wrk) = !isempty(wrk.fw_propagators) && wrk.fw_propagators[1] isa HIPPropagator

function engine_for(wrk::Krotov.KrotovWrk)
    lock(ENGINES_LOCK) do
        get!(ENGINES, wrk) do
            opts = wrk.fw_propagators[1].kwargs          # prop_* kwargs that init_prop_trajectory passed through
            Engine(wrk; devices = Vector{Int}(get(opts, :devices, Int[])), prop_method = get(opts, :hip_prop_method, :auto),
                   prop_tol = Float64(get(opts, :tol, 0.0)), spin_timeout_ms = Int(get(opts, :spin_timeout_ms, 0)), options = String(get(opts, :hip_options, "")))
        end
    end
end

"""Hand a state vector (a column of what the engine returned) to a propagator placeholder: in place where the state type
allows it, re-bound otherwise (immutable states, e.g. SVector) -- the reference's own distinction at optimize.jl:331-335."""
function put_state!(p::HIPPropagator, v::AbstractVector{ComplexF64})
    if supports_inplace(p.state)
        copyto!(p.state, v)
    else
        p.state = convert(typeof(p.state), v)
    end
    return p.state
end

function pack_pulses(eps)
    NT = length(eps[1]); L = length(eps)
    out = Matrix{Float64}(undef, NT, L)               # column l contiguous = [L][N_T] in C order
    for l = 1:L; out[:, l] .= eps[l]; end
    return out
end

"""Per-step propagation callbacks (`wrk.fw_prop_kwargs[k][:callback]` / `wrk.bw_prop_kwargs[k][:callback]`, fired by the reference
after every `prop_step!`: optimize.jl:258-262 initial sweep, :310-314 backward sweep, :362-366 update sweep).  The engine runs whole
sweeps, so they are REPLAYED from the device storages right after the sweep, in the reference's order -- for k, for n -- with the
propagator placeholder carrying the state, time and grid index of that step (`kr_read_storage`; forward storage slot n + 1 after
step n, backward slot n: optimize.jl:263,315,367).  A callback therefore sees every state the reference would have shown it; what it
cannot do is influence the sweep that has already run (the reference's own callbacks only observe).  Needs the forward storage
(the default); the final states are put back afterwards."""
function replay_prop_callbacks!(eng::Engine, wrk, which::Symbol)
    props = which == :bw ? wrk.bw_propagators : wrk.fw_propagators
    kws = which == :bw ? wrk.bw_prop_kwargs : wrk.fw_prop_kwargs
    for k = 1:eng.K
        haskey(kws[k], :callback) || continue
        cb = kws[k][:callback]
        observables = get(kws[k], :observables, nothing)      # (the reference's default is its private _StoreState(); a callback that asks for it gets `nothing`)
        p = props[k]
        steps = which == :bw ? (eng.NT:-1:1) : (2:eng.NT+1)  # storage slots in the order the sweep filled them
        for slot in steps
            put_state!(p, read_storage(eng, which, k, slot))
            p.i = slot
            p.t = p.tlist[slot]
            cb(p, observables)
        end
    end
end
has_prop_callbacks(wrk, which::Symbol) = any(kw -> haskey(kw, :callback), which == :bw ? wrk.bw_prop_kwargs : wrk.fw_prop_kwargs)

"""All-k replacement of the body of `krotov_initial_fw_prop!` (optimize.jl:247-265)."""
function initial_fw_prop!(eng::Engine, eps0, wrk)
    e = pack_pulses(eps0)
    psiT = Matrix{ComplexF64}(undef, eng.N, eng.K)
    GC.@preserve e psiT check(eng.handle, ccall((:kr_forward, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{ComplexF64}), eng.handle, e, psiT))
    has_prop_callbacks(wrk, :fw) && replay_prop_callbacks!(eng, wrk, :fw)   # optimize.jl:258-262
    for k = 1:eng.K
        put_state!(wrk.fw_propagators[k], view(psiT, :, k))          # what update_result! reads (optimize.jl:379)
    end
end

"""Replacement of the body of `krotov_iteration` (optimize.jl:279-371)."""
function iteration!(eng::Engine, wrk, eps_i, eps_ip1)
    chi = wrk.kwargs[:chi]                                            # guaranteed by the KrotovWrk constructor (workspace.jl:171-176)
    Psi = [p.state for p in wrk.fw_propagators]
    X = wrk.chi_takes_tau ? chi(Psi, wrk.trajectories; tau = wrk.result.tau_vals) : chi(Psi, wrk.trajectories)   # optimize.jl:297-302
    chiT = reduce(hcat, [Vector{ComplexF64}(x) for x in X])
    e = pack_pulses(eps_i)
    S = pack_pulses(wrk.update_shapes)
    lam = Vector{Float64}(wrk.lambda_vals)
    e_new = similar(e)
    psiT = Matrix{ComplexF64}(undef, eng.N, eng.K)
    g_a_int = wrk.g_a_int                                            # Vector{Float64}(L): written in place, as optimize.jl:327,357 do
    GC.@preserve e chiT S lam e_new g_a_int psiT check(eng.handle,
        ccall((:kr_iteration, LIB), Cint,
              (Ptr{Cvoid}, Ptr{Float64}, Ptr{ComplexF64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{ComplexF64}),
              eng.handle, e, chiT, S, lam, e_new, g_a_int, psiT))
    for l = 1:eng.L
        eps_ip1[l] .= view(e_new, :, l)                               # in place: callbacks hold these vectors (test_iterations.jl:125-143)
    end
    has_prop_callbacks(wrk, :bw) && replay_prop_callbacks!(eng, wrk, :bw)   # optimize.jl:310-314
    has_prop_callbacks(wrk, :fw) && replay_prop_callbacks!(eng, wrk, :fw)   # optimize.jl:362-366
    for k = 1:eng.K
        put_state!(wrk.fw_propagators[k], view(psiT, :, k))
    end
end

# ---- 3. the two methods added to the reference's generic functions --------------------------------------------------
# More specific than the untyped originals in `wrk` only: no ambiguity, and every call the optimizer makes
# (optimize.jl:183,209) reaches them.  A workspace whose propagators are not ours goes to the original code unchanged.
function Krotov.krotov_initial_fw_prop!(ϵ⁽⁰⁾, ϕₖ, k, wrk::Krotov.KrotovWrk)
    uses_hip(wrk) || return invoke(Krotov.krotov_initial_fw_prop!, Tuple{Any,Any,Any,Any}, ϵ⁽⁰⁾, ϕₖ, k, wrk)
    eng = engine_for(wrk)
    lock(eng.lock) do
        if !eng.initial_done                       # the loop at optimize.jl:182-184 calls once per trajectory: the first call does all
            initial_fw_prop!(eng, ϵ⁽⁰⁾, wrk)
            eng.initial_done = true
        end
    end
    return nothing
end

function Krotov.krotov_iteration(wrk::Krotov.KrotovWrk, ϵ⁽ⁱ⁾, ϵ⁽ⁱ⁺¹⁾)
    uses_hip(wrk) || return invoke(Krotov.krotov_iteration, Tuple{Any,Any,Any}, wrk, ϵ⁽ⁱ⁾, ϵ⁽ⁱ⁺¹⁾)
    eng = engine_for(wrk)
    lock(eng.lock) do
        iteration!(eng, wrk, ϵ⁽ⁱ⁾, ϵ⁽ⁱ⁺¹⁾)
    end
    return nothing
end

"""Callback inspection of the device storages (workspace.jl:24): which = :fw or :bw, n = 1..N_T+1."""
function read_storage(eng::Engine, which::Symbol, k::Int, n::Int)
    out = Vector{ComplexF64}(undef, eng.N)
    GC.@preserve out check(eng.handle, ccall((:kr_read_storage, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{ComplexF64}),
                                             eng.handle, which == :bw ? 1 : 0, k - 1, n - 1, out))
    return out
end
read_storage(wrk::Krotov.KrotovWrk, which::Symbol, k::Int, n::Int) = read_storage(engine_for(wrk), which, k, n)

"""One line: which kernels carry this workspace's sweeps (kr_plan_description)."""
plan(eng::Engine) = unsafe_string(ccall((:kr_plan_description, LIB), Cstring, (Ptr{Cvoid},), eng.handle))

"""Upper bounds of ‖H₀‖₂, ‖H_l‖₂ per generator, `(1 + L) × G` (kr_get_norm_bounds): what sets the degree of every step polynomial."""
function norm_bounds(eng::Engine, L::Int, G::Int = 1)
    out = Matrix{Float64}(undef, 1 + L, G)
    GC.@preserve out check(eng.handle, ccall((:kr_get_norm_bounds, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), eng.handle, out))
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
