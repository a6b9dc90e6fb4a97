"""ctypes loader for oracle/libkrotov_oracle.so (the plain-C restatement).

TEST INFRASTRUCTURE ONLY -- see the header of krotov_oracle.c.  Same array
conventions as krotov_oracle.py (NumPy C-order H (N,N)); the transposition to
the C ABI's column-major operator layout happens here.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_here = os.path.dirname(os.path.abspath(__file__))
_so = os.path.join(_here, "libkrotov_oracle.so")


def build(force=False):
    if force or not os.path.exists(_so) or os.path.getmtime(_so) < os.path.getmtime(
            os.path.join(_here, "krotov_oracle.c")):
        subprocess.check_call(["make", "-C", _here, "-s"])
    return _so


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _colmajor(H):
    H = np.asarray(H, np.complex128)
    return np.ascontiguousarray(np.swapaxes(H, -1, -2))


def max_threads():
    return lib().orc_max_threads()


def expm(A):
    A = np.asarray(A, np.complex128)
    n = A.shape[0]
    Ac = _colmajor(A)
    E = np.empty_like(Ac)
    rc = lib().orc_expm(C.c_int(n), _p(Ac), _p(E))
    if rc:
        raise RuntimeError("orc_expm failed rc=%d" % rc)
    return np.ascontiguousarray(E.T)


def forward(H0, Hc, tlist, eps, psi0, store=True, mode=1, nthreads=1):
    K, N = psi0.shape
    L, NT = eps.shape
    tl = np.ascontiguousarray(tlist, np.float64)
    H0c, Hcc = _colmajor(H0), _colmajor(Hc)
    e = np.ascontiguousarray(eps, np.float64)
    p0 = np.ascontiguousarray(psi0, np.complex128)
    Phi = np.empty((NT + 1, K, N), np.complex128) if store else None
    psiT = np.empty((K, N), np.complex128)
    rc = lib().orc_forward(N, NT, K, L, _p(tl), _p(H0c), _p(Hcc), _p(e), _p(p0), mode, nthreads,
                           _p(Phi) if store else None, _p(psiT))
    if rc:
        raise RuntimeError("orc_forward failed rc=%d" % rc)
    return psiT, Phi


def backward(H0, Hc, tlist, eps, chiT, mode=1, nthreads=1):
    K, N = chiT.shape
    L, NT = eps.shape
    tl = np.ascontiguousarray(tlist, np.float64)
    H0c, Hcc = _colmajor(H0), _colmajor(Hc)
    e = np.ascontiguousarray(eps, np.float64)
    c = np.ascontiguousarray(chiT, np.complex128)
    X = np.empty((NT + 1, K, N), np.complex128)
    rc = lib().orc_backward(N, NT, K, L, _p(tl), _p(H0c), _p(Hcc), _p(e), _p(c), mode, nthreads, _p(X))
    if rc:
        raise RuntimeError("orc_backward failed rc=%d" % rc)
    return X


def update_forward(H0, Hc, tlist, eps_guess, S, lam, psi0, X, mask=None, store=True, mode=1,
                   nthreads=1):
    K, N = psi0.shape
    L, NT = eps_guess.shape
    tl = np.ascontiguousarray(tlist, np.float64)
    H0c, Hcc = _colmajor(H0), _colmajor(Hc)
    e = np.ascontiguousarray(eps_guess, np.float64)
    Sc = np.ascontiguousarray(S, np.float64)
    lc = np.ascontiguousarray(lam, np.float64)
    p0 = np.ascontiguousarray(psi0, np.complex128)
    Xc = np.ascontiguousarray(X, np.complex128)
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    eps_new = np.empty((L, NT))
    ga = np.empty(L)
    Phi = np.empty((NT + 1, K, N), np.complex128) if store else None
    psiT = np.empty((K, N), np.complex128)
    rc = lib().orc_update_forward(N, NT, K, L, _p(tl), _p(H0c), _p(Hcc), None if m is None else _p(m),
                                  _p(e), _p(Sc), _p(lc), _p(p0), _p(Xc), mode, nthreads, _p(eps_new),
                                  _p(ga), _p(Phi) if store else None, _p(psiT))
    if rc:
        raise RuntimeError("orc_update_forward failed rc=%d" % rc)
    return eps_new, ga, psiT, Phi


def iteration(H0, Hc, tlist, eps_guess, S, lam, psi0, chiT, mask=None, mode=1, nthreads=1):
    X = backward(H0, Hc, tlist, eps_guess, chiT, mode, nthreads)
    eps_new, ga, psiT, Phi = update_forward(H0, Hc, tlist, eps_guess, S, lam, psi0, X, mask, True, mode,
                                            nthreads)
    return eps_new, ga, psiT, X, Phi
