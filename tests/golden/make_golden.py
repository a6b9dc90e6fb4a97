"""Generates the golden fixtures in this directory with the NumPy/SciPy oracle twin
(oracle/krotov_oracle.py; scipy.linalg.expm).  The reference itself (Julia) cannot run in the build
image, so these vectors are the restatement's, pinned by the reference's TLS thresholds
(test/test_tls_optimization.jl:66-67) -- "parity unpinned" at the bit level, see DESIGN.md.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import krotov_oracle as o  # noqa: E402


def herm(rng, n):
    A = rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))
    H = 0.5 * (A + A.conj().T)
    return H / np.max(np.abs(np.linalg.eigvalsh(H)))


def main():
    # KAT-TLS: pulses after iterations 1 and 5, J_T history, optimized control on tlist
    hist, pulses, ctrl = o.optimize_tls_kat(iters=5, npoints=501)
    np.savez(os.path.join(HERE, "tls_kat.npz"), J_T=np.array(hist), pulse_iter0=pulses[0][0],
             pulse_iter1=pulses[1][0], pulse_iter5=pulses[5][0], optimized_control=ctrl)

    # seeded N=16 / K=4 / NT=50 / L=1 Hermitian case: one full iteration
    rng = np.random.default_rng(160450)
    N, K, NT, L = 16, 4, 50, 1
    H0, Hc = herm(rng, N), np.array([herm(rng, N) for _ in range(L)])
    tlist = np.linspace(0, 5.0, NT + 1)
    eps = 0.2 * np.array([[o.flattop(t, T=5.0, t_rise=0.5) for t in 0.5 * (tlist[:-1] + tlist[1:])]])
    S = np.ones((L, NT))
    lam = np.array([0.5])
    Q, _ = np.linalg.qr(rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N)))
    psi0 = np.ascontiguousarray(Q[:, :K].T)
    Q2, _ = np.linalg.qr(rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N)))
    tgt = np.ascontiguousarray(Q2[:, :K].T)
    psiT, _ = o.forward(H0, Hc, tlist, eps, psi0)
    tau = o.taus(psiT, tgt)
    chiT = o.chi_ss(tau, tgt)
    eps_new, ga, psiT1, X, Phi = o.iteration(H0, Hc, tlist, eps, S, lam, psi0, chiT)
    np.savez(os.path.join(HERE, "hermitian_n16.npz"), H0=H0, Hc=Hc, tlist=tlist, eps=eps, S=S, lam=lam, psi0=psi0,
             tgt=tgt, psiT0=psiT, chiT=chiT, eps_new=eps_new, g_a_int=ga, psiT1=psiT1, X0=X[0], X_mid=X[NT // 2])

    # seeded N=8 non-Hermitian (dissipative) case, 2 controls, non-uniform grid
    rng = np.random.default_rng(80808)
    N, K, NT, L = 8, 3, 40, 2
    H0 = herm(rng, N) - 0.05j * np.diag(rng.uniform(0, 1, N)) + 0.03 * rng.normal(size=(N, N))
    Hc = np.array([herm(rng, N) for _ in range(L)])
    tlist = np.cumsum(np.r_[0.0, rng.uniform(0.05, 0.15, NT)])
    eps = rng.normal(size=(L, NT)) * 0.2
    S = rng.uniform(0.5, 1.0, (L, NT))
    lam = np.array([2.0, 3.0])
    psi0 = rng.normal(size=(K, N)) + 1j * rng.normal(size=(K, N))
    psi0 /= np.linalg.norm(psi0, axis=1, keepdims=True)
    tgt = rng.normal(size=(K, N)) + 1j * rng.normal(size=(K, N))
    tgt /= np.linalg.norm(tgt, axis=1, keepdims=True)
    psiT, _ = o.forward(H0, Hc, tlist, eps, psi0)
    chiT = o.chi_re(o.taus(psiT, tgt), tgt)
    eps_new, ga, psiT1, X, Phi = o.iteration(H0, Hc, tlist, eps, S, lam, psi0, chiT)
    np.savez(os.path.join(HERE, "nonhermitian_n8.npz"), H0=H0, Hc=Hc, tlist=tlist, eps=eps, S=S, lam=lam, psi0=psi0,
             tgt=tgt, psiT0=psiT, chiT=chiT, eps_new=eps_new, g_a_int=ga, psiT1=psiT1, X0=X[0], X_mid=X[NT // 2])
    dump_flat("hermitian_n16")
    dump_flat("nonhermitian_n8")
    print("golden fixtures written to", HERE)


def dump_flat(name):
    """The same fixture as a flat little-endian binary in the C ABI's memory layout (include/krotov_hip.h), for consumers that
    cannot read .npz -- tests/abi_consumer.c, the plain-C99 stand-in for the ccall binding:
        int32  magic 0x4B52474C ("KRGL"), N, NT, K, L, 0, 0, 0
        double tlist[NT+1], H0[N*N] complex COLUMN-major, Hc[L][N*N] complex column-major, eps[L][NT], S[L][NT], lam[L],
               psi0[K][N] complex, chiT[K][N], psiT0[K][N], eps_new[L][NT], g_a_int[L], psiT1[K][N], X0[K][N], X_mid[K][N]"""
    g = np.load(os.path.join(HERE, name + ".npz"))
    N, L = g["H0"].shape[0], g["Hc"].shape[0]
    NT, K = g["eps"].shape[1], g["psi0"].shape[0]
    with open(os.path.join(HERE, name + ".bin"), "wb") as f:
        np.array([0x4B52474C, N, NT, K, L, 0, 0, 0], "<i4").tofile(f)
        g["tlist"].astype("<f8").tofile(f)
        np.ascontiguousarray(g["H0"].T).astype("<c16").tofile(f)                      # column-major
        np.ascontiguousarray(np.swapaxes(g["Hc"], -1, -2)).astype("<c16").tofile(f)
        for key, dt in (("eps", "<f8"), ("S", "<f8"), ("lam", "<f8"), ("psi0", "<c16"), ("chiT", "<c16"), ("psiT0", "<c16"),
                        ("eps_new", "<f8"), ("g_a_int", "<f8"), ("psiT1", "<c16"), ("X0", "<c16"), ("X_mid", "<c16")):
            np.ascontiguousarray(g[key]).astype(dt).tofile(f)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "flat":      # only re-dump the flat binaries from the committed .npz files
        dump_flat("hermitian_n16")
        dump_flat("nonhermitian_n8")
    else:
        main()
