"""Synthetic workloads of BASELINE.json's configs.  All fp64, seeds fixed.

C1  TLS of /root/reference/test/test_tls_optimization.jl:12-28,50-63 (nt=100 variant too)
C2  N=16, NT=500,  K=32,   L=1
C3  N=64, NT=1000, K=256,  L=2   (the headline configuration)
C4  N=64, NT=1000, K=1024, L=2
C5  N=256 Liouvillian (16-level system, vectorised rho), NT=2000, K=64, L=1
"""
import numpy as np


def blackman(t, t0, T, a=0.16):
    dT = T - t0
    box = 1.0 if (t0 <= t <= T) else 0.0
    return 0.5 * box * (1.0 - a - np.cos(2 * np.pi * (t - t0) / dT) + a * np.cos(4 * np.pi * (t - t0) / dT))


def flattop(t, T, t_rise, t0=0.0, t_fall=None):
    """QuantumPropagators.Shapes.flattop(...; func=:blackman) [UPSTREAM-RECALLED]."""
    if t_fall is None:
        t_fall = t_rise
    f = 0.0
    if t0 <= t <= T:
        f = 1.0
        if t <= t0 + t_rise:
            f = blackman(t, t0, t0 + 2 * t_rise)
        elif t >= T - t_fall:
            f = blackman(t, T - 2 * t_fall, T)
    return f


def midpoints(tlist):
    tl = np.asarray(tlist, float)
    tm = 0.5 * (tl[:-1] + tl[1:])
    tm[0], tm[-1] = tl[0], tl[-1]
    return tm


def random_hermitian(rng, N, radius=1.0):
    A = rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N))
    H = 0.5 * (A + A.conj().T)
    return H * (radius / np.max(np.abs(np.linalg.eigvalsh(H))))


def random_states(rng, K, N):
    if K <= N:
        Q, _ = np.linalg.qr(rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N)))
        return np.ascontiguousarray(Q[:, :K].T)
    v = rng.normal(size=(K, N)) + 1j * rng.normal(size=(K, N))
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def dense_problem(N, NT, K, L, seed, dt, lam, amp=0.2):
    """Shared dense Hermitian generator, seeded."""
    rng = np.random.default_rng(seed)
    H0 = random_hermitian(rng, N)
    Hc = np.array([random_hermitian(rng, N) for _ in range(L)])
    tlist = np.arange(NT + 1) * dt
    T = tlist[-1]
    tm = midpoints(tlist)
    shape = np.array([flattop(t, T=T, t_rise=0.1 * T) for t in tm])
    eps = np.array([amp * (1.0 - 0.3 * l) * shape for l in range(L)])
    S = np.tile(shape, (L, 1))
    psi0 = random_states(rng, K, N)
    tgt = random_states(np.random.default_rng(seed + 7), K, N)
    return dict(H0=H0, Hc=Hc, tlist=tlist, eps=eps, S=S, lam=np.full(L, float(lam)), psi0=psi0, tgt=tgt,
                functional="ss", name="N%d_NT%d_K%d_L%d" % (N, NT, K, L))


def tls_problem(npoints=501):
    H0 = -0.5 * np.array([[1, 0], [0, -1]], np.complex128)
    Hc = np.array([[[0, 1], [1, 0]]], np.complex128)
    tlist = np.linspace(0.0, 5.0, npoints)
    tm = midpoints(tlist)
    eps = np.array([[0.2 * flattop(t, T=5.0, t_rise=0.3) for t in tm]])
    NT = npoints - 1
    return dict(H0=H0, Hc=Hc, tlist=tlist, eps=eps, S=np.ones((1, NT)), lam=np.array([1.0]),
                psi0=np.array([[1, 0]], np.complex128), tgt=np.array([[0, 1]], np.complex128),
                functional="sm", name="TLS_%d" % npoints)


def liouvillian_problem(n_levels=16, NT=2000, K=64, seed=256001, dt=0.05, lam=5.0, n_jump=2, rate=0.01):
    """C5: drho/dt = -i Lrho, L = [H, .] + i sum_j D[c_j] in that convention; dense, non-Hermitian.
    Row-major vectorisation: vec(A rho B) = (A kron B^T) vec(rho)."""
    rng = np.random.default_rng(seed)
    n = n_levels
    I = np.eye(n)

    def comm(H):
        return np.kron(H, I) - np.kron(I, H.T)

    H0 = random_hermitian(rng, n)
    H1 = random_hermitian(rng, n)
    L0 = comm(H0).astype(np.complex128)
    for _ in range(n_jump):
        c = (rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))) / np.sqrt(n)
        cdc = c.conj().T @ c
        D = np.kron(c, c.conj()) - 0.5 * np.kron(cdc, I) - 0.5 * np.kron(I, cdc.T)
        L0 = L0 + 1j * rate * D
    Lc = comm(H1).astype(np.complex128)[None]
    tlist = np.arange(NT + 1) * dt
    T = tlist[-1]
    tm = midpoints(tlist)
    shape = np.array([flattop(t, T=T, t_rise=0.1 * T) for t in tm])

    def rand_rho(r):
        A = r.normal(size=(n, n)) + 1j * r.normal(size=(n, n))
        rho = A @ A.conj().T
        return (rho / np.trace(rho)).reshape(-1)

    psi0 = np.array([rand_rho(rng) for _ in range(K)])
    r2 = np.random.default_rng(seed + 7)
    tgt = np.array([rand_rho(r2) for _ in range(K)])
    return dict(H0=L0, Hc=Lc, tlist=tlist, eps=0.2 * shape[None], S=shape[None], lam=np.array([float(lam)]),
                psi0=psi0, tgt=tgt, functional="re", name="Liouville%d_NT%d_K%d" % (n * n, NT, K))


def config(name):
    if name == "C1":
        return tls_problem(101)
    if name == "C1_501":
        return tls_problem(501)
    if name == "C2":
        return dense_problem(16, 500, 32, 1, 16001, 0.1, 1.0)
    if name == "C3":
        return dense_problem(64, 1000, 256, 2, 64001, 0.05, 5.0)
    if name == "C4":
        return dense_problem(64, 1000, 1024, 2, 64001, 0.05, 5.0)
    if name == "C5":
        return liouvillian_problem()
    if name.startswith("D:"):   # ad-hoc dense problem for the developer tools: "D:N,NT,K,L"
        N, NT, K, L = (int(x) for x in name[2:].split(","))
        return dense_problem(N, NT, K, L, 7000 + N, 0.05, 5.0)
    if name.startswith("L:"):   # ad-hoc open system for the developer tools: "L:levels,NT,K" (N = levels^2, one control)
        n, NT, K = (int(x) for x in name[2:].split(","))
        return liouvillian_problem(n_levels=n, NT=NT, K=K, seed=9000 + n)
    raise KeyError(name)


def chi_and_JT(functional, psiT, tgt, K_global=None):
    """tau, J_T and chi(T) of the standard functionals [QuantumControl.Functionals, weights 1];
    host-side counterpart of the device kernels (optimize.jl:297-302,381-386)."""
    K = len(psiT) if K_global is None else K_global
    tau = np.einsum("kn,kn->k", np.conj(tgt), psiT)
    if functional == "sm":
        return tau, 1.0 - abs(np.sum(tau) / K) ** 2, (np.sum(tau) / K ** 2) * tgt
    if functional == "ss":
        return tau, 1.0 - np.sum(np.abs(tau) ** 2) / K, (tau / K)[:, None] * tgt
    if functional == "re":
        return tau, 1.0 - np.sum(tau).real / K, tgt / (2.0 * K)
    raise KeyError(functional)
