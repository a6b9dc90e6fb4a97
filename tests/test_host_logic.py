"""CPU: the host-side driver (krotov.jl_amd/optimize.py, mirror of optimize_krotov / KrotovWrk /
KrotovResult) exercised with the oracle standing in for the device engine.  Mirrors the STRUCTURE of the
reference's tests (their random streams -- StableRNG, QuantumControlTestUtils -- cannot be reproduced)."""
import io

import numpy as np
import pytest

import krotov_jl_amd as kr
from oracle_engine import factory


def tls_problem(**kw):
    eps = lambda t: 0.2 * kr.flattop(t, T=5, t_rise=0.3)
    H0 = -0.5 * np.array([[1, 0], [0, -1]], complex)
    H1 = np.array([[0, 1], [1, 0]], complex)
    H = kr.hamiltonian(H0, (H1, eps))
    tlist = np.linspace(0, 5, 501)
    traj = kr.Trajectory(np.array([1, 0], complex), H, target_state=np.array([0, 1], complex))
    return kr.ControlProblem([traj], tlist, engine_factory=factory, **kw)


def dummy_problem(N=4, K=2, NT=30, seed=3, pulses_as_controls=False, **kw):
    rng = np.random.default_rng(seed)

    def herm():
        A = rng.normal(size=(N, N))
        return (A + A.T) / 2
    tlist = np.linspace(0, 3, NT + 1)
    ctrl = (0.1 * np.ones(NT)) if pulses_as_controls else (lambda t: 0.1 * kr.flattop(t, T=3, t_rise=0.5))
    H = kr.hamiltonian(herm(), (herm(), ctrl))
    trajs = []
    for _ in range(K):
        a = rng.normal(size=N) + 1j * rng.normal(size=N)
        b = rng.normal(size=N) + 1j * rng.normal(size=N)
        trajs.append(kr.Trajectory(a / np.linalg.norm(a), H, target_state=b / np.linalg.norm(b)))
    return kr.ControlProblem(trajs, tlist, engine_factory=factory, **kw), ctrl


def test_tls_optimization():
    """test_tls_optimization.jl:47-70."""
    problem = tls_problem(iter_stop=5, J_T=kr.J_T_sm,
                          check_convergence=lambda res: ((res.J_T < 1e-10) and (setattr(res, "converged", True)
                                                         or setattr(res, "message", "J_T < 1e-10"))))
    out = io.StringIO()
    res = kr.optimize(problem, method="Krotov", print_iters_out=out)
    assert res.J_T < 1e-3
    assert 1.0 < np.max(np.abs(res.optimized_controls[0])) < 1.2
    assert res.converged and res.message == "Reached maximum number of iterations"
    lines = out.getvalue().splitlines()
    assert lines[0].split() == ["iter.", "J_T", "∫gₐ(t)dt", "J", "ΔJ_T", "ΔJ", "secs"]
    assert len(lines) == 7 and lines[1].split()[4] == "n/a"


def test_iter_start_stop_records():
    """test_iterations.jl:13-35: iter_start=10, iter_stop=12 -> recorded iterations [0, 11, 12]."""
    problem, _ = dummy_problem(iter_start=10, iter_stop=12, J_T=kr.J_T_ss)
    res = kr.optimize(problem, store_iter_info=["iter.", "J_T"], print_iters_out=io.StringIO())
    assert [r[0] for r in res.records] == [0, 11, 12]
    assert res.iter == 12 and res.converged


def test_callbacks_and_records():
    """test_iterations.jl:38-121: tuple of callbacks, ordering, record layout."""
    log = []

    def cb1(wrk, it, *a):
        log.append("cb1 %d" % it)

    def cb2(wrk, it, *a):
        log.append("cb2 %d" % it)
        return ("cb2", it)
    problem, _ = dummy_problem(J_T=kr.J_T_ss)
    res = kr.optimize(problem, iter_stop=1, callback=(cb1, cb2), store_iter_info=["J_T"],
                      print_iters_out=io.StringIO())
    assert log == ["cb1 0", "cb2 0", "cb1 1", "cb2 1"]
    assert len(res.records) == 2 and len(res.records[0]) == 3
    assert res.records[0][0] == "cb2" and res.records[0][1] == 0 and isinstance(res.records[0][2], float)


def test_callback_can_modify_updated_pulse():
    """test_iterations.jl:125-143: a callback scaling eps(i+1) takes effect in the next iteration."""
    def reduce_pulse(wrk, it, eps_ip1, eps_i):
        g = np.linalg.norm(eps_i[0])
        u = np.linalg.norm(eps_ip1[0])
        eps_ip1[0] *= 0.8
        return (g, u, np.linalg.norm(eps_ip1[0]))
    problem, _ = dummy_problem(J_T=kr.J_T_ss)
    res = kr.optimize(problem, iter_stop=3, callback=reduce_pulse, store_iter_info=["iter.", "J_T"],
                      print_iters_out=io.StringIO())
    assert res.converged
    for i in range(1, len(res.records)):
        g, u, us, it, JT = res.records[i]
        assert np.isclose(us, 0.8 * u)
        if i >= 2:
            assert np.isclose(g, res.records[i - 1][2])


def test_pulse_as_control_untouched():
    """test_pulse_optimization.jl:10-44."""
    problem, ctrl = dummy_problem(pulses_as_controls=True, J_T=kr.J_T_re, iter_stop=2, lambda_a=0.05)
    guess = ctrl.copy()
    res = kr.optimize(problem, print_iters_out=io.StringIO())
    NT = len(problem.tlist) - 1
    assert len(res.optimized_controls[0]) == NT + 1          # :28
    assert np.array_equal(ctrl, guess)                       # :33-35 input pulse untouched
    assert kr.optimize_krotov is not None
    opt_on_mid = kr.discretize_on_midpoints(res.optimized_controls[0], problem.tlist)
    assert np.linalg.norm(opt_on_mid - guess) > 1e-3         # :38 the pulse changed


def test_empty_optimization_error():
    """test_empty_optimization.jl:8-34."""
    H = kr.hamiltonian(np.eye(2))
    traj = kr.Trajectory(np.array([1, 0], complex), H, target_state=np.array([0, 1], complex))
    problem = kr.ControlProblem([traj], np.linspace(0, 1, 11), J_T=kr.J_T_sm, engine_factory=factory)
    with pytest.raises(RuntimeError, match="no controls in trajectories: cannot optimize"):
        kr.optimize(problem, print_iters=False)


def test_missing_J_T_and_deprecated_hooks():
    problem = tls_problem(iter_stop=1)
    with pytest.raises(RuntimeError, match="must be passed the functional `J_T`"):
        kr.optimize(problem, print_iters=False)
    with pytest.raises(ValueError, match="superseded by the `callback`"):
        kr.optimize(tls_problem(J_T=kr.J_T_sm, info_hook=lambda *a: None), print_iters=False)


def test_continue_from():
    """workspace.jl:107-120: continuing keeps the iteration counter and starts from the optimized controls;
    the first record reproduces the previous J_T (test_tls_optimization.jl:126,160 to 1e-14)."""
    r1 = kr.optimize(tls_problem(iter_stop=2, J_T=kr.J_T_sm), print_iters=False)
    J1 = r1.J_T   # the continuation works in place on the same result object (workspace.jl:109)
    r2 = kr.optimize(tls_problem(J_T=kr.J_T_sm), continue_from=r1, iter_stop=4, store_iter_info=["iter.", "J_T"],
                     print_iters_out=io.StringIO())
    assert r2.iter == 4
    assert [r[0] for r in r2.records] == [0, 3, 4]
    assert abs(r2.records[0][1] - J1) < 1e-12      # tlist <-> midpoint round trip is exact
    assert r2.J_T < J1 and r2 is r1


def test_exception_is_captured():
    """optimize.jl:218-226."""
    def bad(wrk, it, *a):
        if it == 1:
            raise KeyboardInterrupt("stop")
    res = kr.optimize(tls_problem(iter_stop=3, J_T=kr.J_T_sm), callback=bad, print_iters=False)
    assert res.message.startswith("Exception:") and not res.converged
    with pytest.raises(KeyboardInterrupt):
        kr.optimize(tls_problem(iter_stop=3, J_T=kr.J_T_sm), callback=bad, print_iters=False, rethrow_exceptions=True)


def test_ensemble_of_generators_through_driver():
    """Robustness ensemble: the same control drives three TLS with different detunings (distinct generators per
    trajectory, control_derivs[k][l]: workspace.jl:74).  The driver must hand G = 3 generators to the engine
    and J_T_ss must decrease."""
    eps = lambda t: 0.2 * kr.flattop(t, T=5, t_rise=0.3)
    H1 = np.array([[0, 1], [1, 0]], complex)
    tlist = np.linspace(0, 5, 101)
    trajs = []
    for om in (0.9, 1.0, 1.1):
        H = kr.hamiltonian(-0.5 * om * np.array([[1, 0], [0, -1]], complex), (H1, eps))
        trajs.append(kr.Trajectory(np.array([1, 0], complex), H, target_state=np.array([0, 1], complex)))
    trajs.append(kr.Trajectory(np.array([0, 1], complex), trajs[1].generator, target_state=np.array([1, 0], complex)))
    problem = kr.ControlProblem(trajs, tlist, iter_stop=3, J_T=kr.J_T_ss, engine_factory=factory, keep_engine=True)
    out = io.StringIO()
    res = kr.optimize(problem, store_iter_info=["J_T"], print_iters_out=out)
    J = [r[0] for r in res.records]
    assert J[3] < J[2] < J[1] < J[0]


def test_liouvillian_helper_matches_master_equation():
    """liouvillian(H, c_ops): d vec(rho)/dt = -i L vec(rho) must reproduce -i[H,rho] + D[c]rho."""
    rng = np.random.default_rng(5)
    n = 3
    A = rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))
    H0 = (A + A.conj().T) / 2
    c = rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))
    B = rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))
    rho = B @ B.conj().T
    rho /= np.trace(rho)
    L = kr.liouvillian(kr.hamiltonian(H0), [c])
    lhs = (-1j * L.H0 @ rho.reshape(-1)).reshape(n, n)
    cdc = c.conj().T @ c
    rhs = -1j * (H0 @ rho - rho @ H0) + c @ rho @ c.conj().T - 0.5 * (cdc @ rho + rho @ cdc)
    assert np.max(np.abs(lhs - rhs)) < 1e-13
    ctrl = lambda t: 0.1
    L2 = kr.liouvillian(kr.hamiltonian(H0, (A + A.conj().T, ctrl)), [c])
    assert len(L2.ops) == 1 and L2.controls[0] is ctrl and L2.H0.shape == (9, 9)


def test_prop_callbacks_are_replayed():
    """fw/bw per-step callbacks (optimize.jl:258-262, 310-314): replayed from the storages after each sweep."""
    seen = {"fw": 0, "bw": 0}

    def fw_cb(prop, observables):
        assert not prop.backward and prop.state.shape == (2,)
        seen["fw"] += 1

    def bw_cb(prop, observables):
        assert prop.backward
        seen["bw"] += 1
    problem = tls_problem(iter_stop=1, J_T=kr.J_T_sm, fw_prop_callback=fw_cb, bw_prop_callback=bw_cb, keep_engine=True)
    kr.optimize(problem, print_iters=False)
    assert seen["fw"] == 2 * 500 and seen["bw"] == 500   # initial sweep + iteration 1; one backward sweep


def test_atexit_filename_saves_the_result_on_unexpected_exit(tmp_path):
    """optimize.jl:195-206, 229-231: with `atexit_filename` the result as it stands is stored if the interpreter exits while the
    optimization runs (here: a callback that ends the process in iteration 2), and the hook is gone once optimize returns."""
    import os
    import pickle
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "result.pkl"
    script = textwrap.dedent("""
        import sys
        sys.path[:0] = [%r, %r, %r]
        import test_host_logic as t
        import krotov_jl_amd as kr
        def cb(wrk, it, *a):
            if it == 2:
                sys.exit(3)          # an exit the optimization loop does not see to its end
        try:
            kr.optimize(t.tls_problem(iter_stop=5, J_T=kr.J_T_sm, atexit_filename=%r), callback=cb, print_iters=False,
                        rethrow_exceptions=True)
        except SystemExit:
            raise
        """) % (root, os.path.join(root, "tests"), os.path.join(root, "oracle"), str(out))
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300)
    assert r.returncode == 3, r.stderr[-2000:]
    assert "Set callback to store result" in r.stderr
    with open(out, "rb") as f:
        res = pickle.load(f)
    assert res.iter == 2 and res.J_T < res.J_T_prev and len(res.optimized_controls) == 1
    # a run that ends normally removes the hook: nothing is written at interpreter exit
    out2 = tmp_path / "unused.pkl"
    script2 = textwrap.dedent("""
        import sys
        sys.path[:0] = [%r, %r, %r]
        import test_host_logic as t
        import krotov_jl_amd as kr
        res = kr.optimize(t.tls_problem(iter_stop=1, J_T=kr.J_T_sm, atexit_filename=%r), print_iters=False)
        assert res.iter == 1
        """) % (root, os.path.join(root, "tests"), os.path.join(root, "oracle"), str(out2))
    r = subprocess.run([sys.executable, "-c", script2], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert not out2.exists()
