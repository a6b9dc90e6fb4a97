"""Host-side mirror of the reference's optimisation driver, so that parity tests read like the
reference's own tests and the HIP engine is exercised behind the reference's interface.

Restated from /root/reference (Julia is not available in the build image, so this is Python):

  optimize(problem; method=Krotov, kw...)  QuantumControl.optimize [UPSTREAM-RECALLED: merges kw into
                                           problem.kwargs, chains callbacks, adds the print_iters table]
  optimize_krotov(problem)                 src/optimize.jl:161-235
  KrotovWrk(problem)                       src/workspace.jl:65-200
  KrotovResult(problem)                    src/result.jl:34-104
  update_result! / finalize_result!        src/optimize.jl:374-406
  make_krotov_print_iters                  src/optimize.jl:413-496

The two hot functions (krotov_initial_fw_prop!, krotov_iteration: src/optimize.jl:247-371) are NOT
restated here: they are the C ABI calls kr_forward / kr_iteration of libkrotov_hip.so (engine.py).
No CPU implementation of them exists in this package.
"""
import datetime as _dt
import os
import sys
import time

import numpy as np

from .workloads import flattop  # noqa: F401  (re-exported for problem set-up)

__all__ = [
    "Trajectory", "ControlProblem", "Generator", "hamiltonian", "liouvillian", "liouvillian_generator", "optimize",
    "optimize_krotov",
    "KrotovResult", "KrotovWrk", "J_T_sm", "J_T_ss", "J_T_re", "make_chi", "discretize",
    "discretize_on_midpoints", "get_controls", "make_krotov_print_iters",
]


# ------------------------------------------------------------------ controls / discretisation
def discretize_on_midpoints(control, tlist):
    """QuantumPropagators.Controls.discretize_on_midpoints (workspace.jl:102,119,123)."""
    tlist = np.asarray(tlist, float)
    NT = len(tlist) - 1
    if callable(control):
        tm = np.empty(NT)
        tm[0], tm[-1] = tlist[0], tlist[-1]
        for i in range(1, NT - 1):
            tm[i] = tlist[i] + 0.5 * (tlist[i + 1] - tlist[i])
        return np.array([float(control(t)) for t in tm])
    c = np.asarray(control, float)
    if len(c) == NT:
        return c.copy()  # a copy: test/test_pulse_optimization.jl:42
    if len(c) == NT + 1:
        # exact inverse of discretize(): c[i] = (v[i-1] + v[i]) / 2  [UPSTREAM-RECALLED; the hand-over tests
        # test_tls_optimization.jl:126,160 (J_T continuity to 1e-14) only hold if the round trip is exact]
        # [UPSTREAM-RECALLED QuantumPropagators.Controls.discretize_on_midpoints(::Vector, tlist)]: first AND last interval value
        # pinned to the first / last point value, the interior by the recursion -- on a round trip through discretize() identical
        # to the pure recursion (the last interval comes back exactly), on an arbitrary array it differs in the last value only
        v = np.empty(NT)
        v[0] = c[0]
        v[-1] = c[-1]
        for i in range(1, NT - 1):
            v[i] = 2.0 * c[i] - v[i - 1]
        return v
    raise ValueError("control array must have length len(tlist) or len(tlist)-1")


def discretize(control, tlist):
    """QuantumPropagators.Controls.discretize (optimize.jl:403-405, result.jl:61)."""
    tlist = np.asarray(tlist, float)
    if callable(control):
        return np.array([float(control(t)) for t in tlist])
    p = np.asarray(control, float)
    if len(p) == len(tlist):
        return p.copy()
    if len(p) == len(tlist) - 1:
        v = np.zeros(len(p) + 1)
        v[0], v[-1] = p[0], p[-1]
        for i in range(1, len(v) - 1):
            v[i] = 0.5 * (p[i - 1] + p[i])
        return v
    raise ValueError("pulse array must have length len(tlist) or len(tlist)-1")


# ------------------------------------------------------------------ problem description
class Generator:
    """H(t) = H0 + sum_l control_l(t) * H_l   (QuantumPropagators.Generators.Generator for linear
    controls; `hamiltonian(H0, (H1, eps1), ...)`).  Controls are callables or arrays and are
    identified by object identity, as the reference's IdDict does (optimize.jl:249)."""

    def __init__(self, H0, terms):
        self.H0 = np.asarray(H0, np.complex128)
        self.ops = [np.asarray(op, np.complex128) for op, _ in terms]
        self.controls = [c for _, c in terms]


def hamiltonian(H0, *terms):
    return Generator(H0, list(terms))


liouvillian_generator = hamiltonian  # a Liouvillian in the d rho/dt = -i L rho convention is used the same way


def liouvillian(H, c_ops=(), convention="TDSE"):
    """QuantumPropagators.liouvillian(H, c_ops; convention=:TDSE) [UPSTREAM-RECALLED] for dense operators: the
    generator of d vec(rho)/dt = -i L vec(rho) (row-major vec: vec(A rho B) = (A kron B^T) vec(rho)), with
    L = [H, .] + i sum_j D[c_j], D[c] rho = c rho c^+ - {c^+ c, rho}/2.  `H` may be a matrix or a Generator with
    control terms; every control term H_l becomes [H_l, .] with the same control.  States are vec(rho)."""
    if convention != "TDSE":
        raise ValueError("only convention='TDSE' (d rho/dt = -i L rho) is supported")
    gen = H if isinstance(H, Generator) else Generator(H, [])
    n = gen.H0.shape[0]
    I = np.eye(n)

    def comm(A):
        return np.kron(A, I) - np.kron(I, A.T)
    L0 = comm(gen.H0).astype(np.complex128)
    for c in c_ops:
        c = np.asarray(c, np.complex128)
        cdc = c.conj().T @ c
        L0 = L0 + 1j * (np.kron(c, c.conj()) - 0.5 * np.kron(cdc, I) - 0.5 * np.kron(I, cdc.T))
    return Generator(L0, [(comm(op), ctrl) for op, ctrl in zip(gen.ops, gen.controls)])


class Trajectory:
    """QuantumControl.Trajectory(initial_state, generator; target_state)."""

    def __init__(self, initial_state, generator, target_state=None, **kwargs):
        self.initial_state = np.asarray(initial_state, np.complex128)
        self.generator = generator
        self.target_state = None if target_state is None else np.asarray(target_state, np.complex128)
        self.kwargs = kwargs


class ControlProblem:
    """QuantumControl.ControlProblem(trajectories, tlist; kwargs...)."""

    def __init__(self, trajectories, tlist, **kwargs):
        self.trajectories = list(trajectories)
        self.tlist = np.asarray(tlist, float)
        self.kwargs = dict(kwargs)


def get_controls(trajectories):
    """Unique controls over all trajectories, in order of first appearance (workspace.jl:70)."""
    seen, out = set(), []
    for traj in trajectories:
        for c in traj.generator.controls:
            if id(c) not in seen:
                seen.add(id(c))
                out.append(c)
    return out


# ------------------------------------------------------------------ functionals [UPSTREAM-RECALLED]
def _taus(states, trajectories):
    return np.array([np.vdot(t.target_state, s) for s, t in zip(states, trajectories)])


def J_T_sm(states, trajectories, tau=None):
    tau = _taus(states, trajectories) if tau is None else tau
    return 1.0 - abs(np.sum(tau) / len(tau)) ** 2


def J_T_ss(states, trajectories, tau=None):
    tau = _taus(states, trajectories) if tau is None else tau
    return 1.0 - np.sum(np.abs(tau) ** 2) / len(tau)


def J_T_re(states, trajectories, tau=None):
    tau = _taus(states, trajectories) if tau is None else tau
    return 1.0 - np.sum(tau).real / len(tau)


def make_chi(J_T, trajectories):
    """QuantumControl.Functionals.make_chi for the analytic cases (workspace.jl:171-173)."""
    K = len(trajectories)

    def chi_sm(states, trajs, tau=None):
        tau = _taus(states, trajs) if tau is None else tau
        return [(np.sum(tau) / K ** 2) * t.target_state for t in trajs]

    def chi_ss(states, trajs, tau=None):
        tau = _taus(states, trajs) if tau is None else tau
        return [(tau[k] / K) * t.target_state for k, t in enumerate(trajs)]

    def chi_re(states, trajs, tau=None):
        return [t.target_state / (2.0 * K) for t in trajs]

    table = {J_T_sm: chi_sm, J_T_ss: chi_ss, J_T_re: chi_re}
    if J_T in table:
        return table[J_T]
    raise ValueError("no analytic chi for J_T=%r; pass `chi` explicitly" % (J_T,))


_DEVICE_FUNCTIONAL = {J_T_sm: 0, J_T_ss: 1, J_T_re: 2}


# ------------------------------------------------------------------ result (result.jl:34-104)
class KrotovResult:
    def __init__(self, problem):
        tlist = problem.tlist
        controls = get_controls(problem.trajectories)
        self.tlist = np.array(tlist, float)
        self.iter_start = int(problem.kwargs.get("iter_start", 0))
        self.iter_stop = int(problem.kwargs.get("iter_stop", 5000))
        self.iter = self.iter_start
        self.secs = 0.0
        self.tau_vals = np.zeros(len(problem.trajectories), np.complex128)
        self.J_T = 0.0
        self.J_T_prev = 0.0
        self.guess_controls = [discretize(c, tlist) for c in controls]
        self.optimized_controls = [g.copy() for g in self.guess_controls]
        self.states = [np.empty_like(t.initial_state) for t in problem.trajectories]
        self.start_local_time = _dt.datetime.now()
        self.end_local_time = _dt.datetime.now()
        self.records = []
        self.converged = False
        self.message = "in progress"

    def __repr__(self):
        return "KrotovResult<%s>" % self.message

    def __str__(self):
        return ("Krotov Optimization Result\n--------------------------\n"
                "- Started at %s\n- Number of trajectories: %d\n- Number of iterations: %d\n"
                "- Value of functional: %.5e\n- Reason for termination: %s\n- Ended at %s (%s)\n"
                % (self.start_local_time, len(self.states), max(self.iter - self.iter_start, 0), self.J_T,
                   self.message, self.end_local_time, self.end_local_time - self.start_local_time))


# ------------------------------------------------------------------ workspace (workspace.jl:30-200)
def _default_engine_factory(H0, Hc, tlist, K, **kw):
    from .engine import HipEngine
    return HipEngine(H0, Hc, tlist, K, **kw)


class KrotovWrk:
    """Callback-visible workspace.  What the reference keeps per trajectory (propagators, storages)
    lives inside `self.engine` (device memory); fw/bw storages are reachable through
    engine.read_storage (workspace.jl:24)."""

    def __init__(self, problem, verbose=False):
        kwargs = dict(problem.kwargs)  # shallow copy (workspace.jl:76)
        self.use_threads = kwargs.get("use_threads", False)
        self.trajectories = list(problem.trajectories)
        self.controls = get_controls(self.trajectories)
        if len(self.controls) == 0:
            raise RuntimeError("no controls in trajectories: cannot optimize")  # workspace.jl:72
        tlist = problem.tlist
        default_shape = kwargs.get("update_shape", lambda t: 1.0)
        default_lambda = float(kwargs.get("lambda_a", 1.0))
        if "pulse_options" in kwargs:
            for key in ("update_shape", "lambda_a"):
                if key in kwargs:
                    print("Warning: `%s` is ignored due to given `pulse_options`" % key, file=sys.stderr)
            pulse_options = kwargs["pulse_options"]
        else:
            if "update_shape" not in kwargs and "lambda_a" not in kwargs and verbose:
                print("Warning: Using default pulse_options: (:lambda_a => 1.0, :update_shape => (t -> 1.0))",
                      file=sys.stderr)
            pulse_options = {id(c): {"lambda_a": default_lambda, "update_shape": default_shape} for c in self.controls}
        opts = []
        for c in self.controls:
            o = pulse_options.get(id(c), pulse_options.get(c) if _hashable(c) else None)
            if o is None:
                raise RuntimeError("pulse_options must be defined for all controls")  # workspace.jl:98
            opts.append(o)
        self.update_shapes = [discretize_on_midpoints(o["update_shape"], tlist) for o in opts]
        self.lambda_vals = [float(o["lambda_a"]) for o in opts]
        if "continue_from" in kwargs:
            print("Info: Continuing previous optimization", file=sys.stderr)
            result = kwargs["continue_from"]
            if not isinstance(result, KrotovResult):
                result = _convert_result(result, problem)
            result.iter_stop = int(kwargs.get("iter_stop", 5000))
            result.converged = False
            result.start_local_time = _dt.datetime.now()
            result.message = "in progress"
            self.pulses0 = [discretize_on_midpoints(c, tlist) for c in result.optimized_controls]
        else:
            result = KrotovResult(problem)
            self.pulses0 = [discretize_on_midpoints(c, tlist) for c in self.controls]
        self.result = result
        self.pulses1 = [p.copy() for p in self.pulses0]
        self.g_a_int = np.zeros(len(self.pulses0))
        if "J_T" not in kwargs:
            raise RuntimeError("`optimize` for `method=Krotov` must be passed the functional `J_T`.")  # workspace.jl:166
        J_T = kwargs["J_T"]
        if "chi" not in kwargs:
            kwargs["chi"] = make_chi(J_T, self.trajectories)
        self.kwargs = kwargs
        self.J_T_takes_tau = _takes_tau(J_T)
        self.chi_takes_tau = _takes_tau(kwargs["chi"])

        # ---- the device engine replaces fw/bw propagators and the three storages (workspace.jl:128-161)
        # distinct generators (by identity, or by value) -> G packs + the k -> g map
        gens, traj_gen = [], []
        for traj in self.trajectories:
            g = traj.generator
            idx = None
            for q, other in enumerate(gens):
                if g is other or (np.array_equal(g.H0, other.H0) and len(g.ops) == len(other.ops)
                                  and all(np.array_equal(a, b) for a, b in zip(g.ops, other.ops))
                                  and all(a is b for a, b in zip(g.controls, other.controls))):
                    idx = q
                    break
            if idx is None:
                gens.append(g)
                idx = len(gens) - 1
            traj_gen.append(idx)
        N = gens[0].H0.shape[0]
        G, L = len(gens), len(self.controls)
        H0 = np.zeros((G, N, N), np.complex128)
        Hc = np.zeros((G, L, N, N), np.complex128)
        present = np.zeros((G, L), np.uint8)
        for gi, g in enumerate(gens):
            H0[gi] = g.H0
            for l, c in enumerate(self.controls):  # control_derivs[k][l] (workspace.jl:74); `nothing` -> absent
                for op, gc in zip(g.ops, g.controls):
                    if gc is c:
                        Hc[gi, l] += op
                        present[gi, l] = 1
        self.control_derivs = [[Hc[traj_gen[k], l] if present[traj_gen[k], l] else None for l in range(L)]
                               for k in range(len(self.trajectories))]
        factory = kwargs.get("engine_factory", _default_engine_factory)
        ekw = {"prop_tol": float(kwargs["prop_tol"])} if "prop_tol" in kwargs else {}
        if G == 1:
            self.engine = factory(H0[0], Hc[0], tlist, len(self.trajectories), hc_present=present[0], **ekw)
        else:
            self.engine = factory(H0, Hc, tlist, len(self.trajectories), hc_present=present,
                                  traj_generator=np.array(traj_gen, np.int32), **ekw)
        self.engine.set_initial_states(np.array([t.initial_state for t in self.trajectories]))
        self.final_states = None


def _hashable(x):
    try:
        hash(x)
        return True
    except TypeError:
        return False


def _takes_tau(fn):
    import inspect
    try:
        return "tau" in inspect.signature(fn).parameters
    except (TypeError, ValueError):
        return False


def _convert_result(other, problem):
    """convert(KrotovResult, result) for results of other methods (workspace.jl:110-113): any
    object with the AbstractOptimizationResult fields."""
    r = KrotovResult(problem)
    for name in ("iter_start", "iter_stop", "iter", "secs", "tau_vals", "J_T", "J_T_prev", "guess_controls",
                 "optimized_controls", "states", "records", "converged", "message"):
        if hasattr(other, name):
            setattr(r, name, getattr(other, name))
    return r


class _PropView:
    """What a per-step propagation callback gets to see (optimize.jl:258-262, 310-314, 362-366): the propagator's
    `state`, the time-grid index and the trajectory index.  The sweeps run on the device as a whole, so the
    callbacks are REPLAYED from the device storages right after each sweep, in the reference's order (for k, for n)."""

    def __init__(self, state, tlist, i, k, backward):
        self.state, self.tlist, self.i, self.k, self.backward = state, tlist, i, k, backward
        self.t = tlist[i]


def _replay_prop_callbacks(wrk, which):
    key = {"fw": ("fw_prop_callback", "prop_callback"), "bw": ("bw_prop_callback", "prop_callback")}[which]
    cb = next((wrk.kwargs[k] for k in key if k in wrk.kwargs), None)
    if cb is None or not hasattr(wrk.engine, "read_storage"):
        return
    NT = len(wrk.result.tlist) - 1
    steps = range(1, NT + 1) if which == "fw" else range(NT - 1, -1, -1)
    for k in range(len(wrk.trajectories)):
        for n in steps:
            cb(_PropView(wrk.engine.read_storage(which, k, n), wrk.result.tlist, n, k, which == "bw"), None)


# ------------------------------------------------------------------ driver (optimize.jl:161-235)
def _update_result(wrk, i):
    """update_result! (optimize.jl:374-397)."""
    res = wrk.result
    J_T = wrk.kwargs["J_T"]
    res.J_T_prev = res.J_T
    for k in range(len(res.states)):
        res.states[k] = wrk.final_states[k]
    res.tau_vals = _taus_ignore_missing(res.states, wrk.trajectories)
    if wrk.J_T_takes_tau:
        res.J_T = float(J_T(res.states, wrk.trajectories, tau=res.tau_vals))
    else:
        res.J_T = float(J_T(res.states, wrk.trajectories))
    if i > 0:
        res.iter = i
    if i >= res.iter_stop:
        res.converged = True
        res.message = "Reached maximum number of iterations"
    prev = res.end_local_time
    res.end_local_time = _dt.datetime.now()
    res.secs = (res.end_local_time - prev).total_seconds()


def _taus_ignore_missing(states, trajectories):
    return np.array([np.vdot(t.target_state, s) if t.target_state is not None else 0.0
                     for s, t in zip(states, trajectories)], np.complex128)


def _krotov_iteration(wrk, eps_i, eps_ip1):
    """krotov_iteration(wrk, eps(i), eps(i+1)) (optimize.jl:279-371): chi(T) on the host (it may be a
    user function), then ONE call into the engine for backward sweep + sequential update sweep."""
    chi = wrk.kwargs["chi"]
    Psi = list(wrk.final_states)
    if wrk.chi_takes_tau:
        chiT = chi(Psi, wrk.trajectories, tau=wrk.result.tau_vals)
    else:
        chiT = chi(Psi, wrk.trajectories)
    S = np.array(wrk.update_shapes)
    lam = np.array(wrk.lambda_vals)
    eps_new, g_a_int, psiT = wrk.engine.iteration(np.array(eps_i), np.array(chiT), S, lam)
    _replay_prop_callbacks(wrk, "bw")
    _replay_prop_callbacks(wrk, "fw")
    for l in range(len(eps_ip1)):
        eps_ip1[l][:] = eps_new[l]  # in-place: callbacks hold references to these vectors
    wrk.g_a_int[:] = g_a_int
    wrk.final_states = [psiT[k] for k in range(psiT.shape[0])]


def optimize_krotov(problem):
    kw = problem.kwargs
    callback = kw.get("callback", lambda *a: None)
    if "update_hook" in kw or "info_hook" in kw:
        raise ValueError("The `update_hook` and `info_hook` arguments have been superseded by the `callback` argument")
    check_convergence = kw.get("check_convergence", lambda res: res)
    verbose = kw.get("verbose", False)
    skip_initial = kw.get("skip_initial_forward_propagation", False)

    wrk = KrotovWrk(problem, verbose=verbose)
    eps_i, eps_ip1 = wrk.pulses0, wrk.pulses1
    if skip_initial:
        print("Info: Skipping initial forward propagation", file=sys.stderr)
        wrk.final_states = [t.initial_state.copy() for t in wrk.trajectories]
    else:
        psiT = wrk.engine.forward(np.array(eps_i))  # krotov_initial_fw_prop! for all k (optimize.jl:182-184)
        _replay_prop_callbacks(wrk, "fw")
        wrk.final_states = [psiT[k] for k in range(psiT.shape[0])]
    _update_result(wrk, 0)
    info = callback(wrk, 0, eps_ip1, eps_i)
    if info is not None and len(info) > 0:
        wrk.result.records.append(tuple(info))
    i = wrk.result.iter
    # atexit_filename (optimize.jl:195-206; undocumented in the reference on purpose: the hook of @optimize_or_load): the result as
    # it stands is written there if the interpreter exits unexpectedly while the loop runs; the hook is removed afterwards (:229-231)
    atexit_filename = kw.get("atexit_filename", None)
    atexit_hook = None
    if atexit_filename is not None:
        import atexit
        import pickle

        def atexit_hook(result=wrk.result, filename=atexit_filename):
            with open(filename, "wb") as f:
                pickle.dump(result, f)
        atexit.register(atexit_hook)
        print("Info: Set callback to store result in %s on unexpected exit." % os.path.relpath(atexit_filename), file=sys.stderr)
    try:
        while not wrk.result.converged:
            i += 1
            _krotov_iteration(wrk, eps_i, eps_ip1)
            _update_result(wrk, i)
            info = callback(wrk, i, eps_ip1, eps_i)
            if info is not None and len(info) > 0:
                wrk.result.records.append(tuple(info))
            check_convergence(wrk.result)
            eps_i, eps_ip1 = eps_ip1, eps_i
    except BaseException as exc:  # Ctrl-C included (optimize.jl:218-226)
        if kw.get("rethrow_exceptions", False):
            raise
        wrk.result.message = "Exception: %s" % (exc,)
    # finalize_result! (optimize.jl:400-406)
    res = wrk.result
    res.end_local_time = _dt.datetime.now()
    for l in range(len(eps_i)):
        res.optimized_controls[l] = discretize(eps_i[l], res.tlist)
    if atexit_hook is not None:
        import atexit
        atexit.unregister(atexit_hook)
    if hasattr(wrk.engine, "close") and not kw.get("keep_engine", False):
        wrk.engine.close()
    return res


# ------------------------------------------------------------------ print_iters (optimize.jl:413-496)
_HEADER = ["iter.", "J_T", "∫gₐ(t)dt", "J", "ΔJ_T", "ΔJ", "secs"]


def make_krotov_print_iters(store_iter_info=(), out=None, **_):
    store = set(store_iter_info)
    for item in store:
        if item not in _HEADER:
            raise ValueError("Item %r in `store_iter_info` is not one of %r)" % (item, _HEADER))

    def print_table(wrk, iteration, *args):
        f = out or sys.stdout
        J_T = wrk.result.J_T
        g_a_int = float(np.sum(wrk.g_a_int))
        J = J_T + g_a_int
        dJ_T = J_T - wrk.result.J_T_prev
        dJ = dJ_T + g_a_int
        secs = wrk.result.secs
        vals = [iteration, J_T, g_a_int, J, dJ_T, dJ, secs]
        iter_stop = str(wrk.kwargs.get("iter_stop", 5000))
        widths = [max(len(iter_stop), 6), 11, 11, 11, 11, 11, 8]
        if iteration == 0:
            f.write("".join(h.rjust(w) for h, w in zip(_HEADER, widths)) + "\n")
        strs = [str(iteration), "%.2e" % J_T, "%.2e" % g_a_int, "%.2e" % J,
                ("%.2e" % dJ_T) if iteration > 0 else "n/a", ("%.2e" % dJ) if iteration > 0 else "n/a",
                "%.1f" % secs]
        f.write("".join(s.rjust(w) for s, w in zip(strs, widths)) + "\n")
        f.flush()
        return tuple(v for h, v in zip(_HEADER, vals) if h in store)

    return print_table


def _chain_callbacks(*callbacks):
    """QuantumControl.chain_callbacks [UPSTREAM-RECALLED]: call all, concatenate the returned tuples."""

    def chained(*args):
        res = ()
        for cb in callbacks:
            r = cb(*args)
            if r is not None:
                res = res + tuple(r)
        return res if len(res) > 0 else None

    return chained


def optimize(problem, method="Krotov", print_iters=True, store_iter_info=(), **kwargs):
    """QuantumControl.optimize(problem; method=Krotov, kwargs...) [UPSTREAM-RECALLED]: keyword arguments
    override problem.kwargs for this call; `callback` may be a tuple; with print_iters the iteration table
    is printed after the user callbacks and its `store_iter_info` values are appended to their records."""
    if str(method).lower() not in ("krotov",):
        raise ValueError("only method=Krotov is provided by this package")
    merged = dict(problem.kwargs)
    merged.update(kwargs)
    cb = merged.get("callback", None)
    cbs = list(cb) if isinstance(cb, (tuple, list)) else ([cb] if cb is not None else [])
    if print_iters:
        cbs.append(make_krotov_print_iters(store_iter_info=store_iter_info, out=merged.pop("print_iters_out", None)))
    if cbs:
        merged["callback"] = _chain_callbacks(*cbs)
    temp = ControlProblem(problem.trajectories, problem.tlist, **merged)
    return optimize_krotov(temp)
