/* CPU ORACLE (plain C99) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker / the timed CPU baseline ("port").
 * libkrotov_hip.so never links or calls anything in here.
 *
 * PARITY UNPINNED: the reference is Julia; no `julia` in the build container, the
 * arithmetic lives in un-vendored, un-pinned QuantumPropagators.jl /
 * QuantumControl.jl (/root/reference/Project.toml:19), and the reference's tests
 * hold no golden vectors.  Pins: the TLS threshold assertions
 * (/root/reference/test/test_tls_optimization.jl:66-67) and agreement with the
 * independent NumPy/SciPy twin (oracle/krotov_oracle.py; scipy.linalg.expm).
 *
 * What is restated, in the reference's order of arithmetic:
 *   orc_forward         krotov_initial_fw_prop!  /root/reference/src/optimize.jl:247-265
 *   orc_backward        backward sweep           /root/reference/src/optimize.jl:303-317
 *                       (adjoint generator, backward=true: workspace.jl:69,149-161)
 *   orc_update_forward  sequential update sweep  /root/reference/src/optimize.jl:321-370
 *   orc_expm            what ExpProp's prop_step! does per step [UPSTREAM
 *                       QuantumPropagators: U = exp(-i H dt), Julia's dense exp =
 *                       Higham-2005 Pade scaling-and-squaring]
 *
 * Memory conventions = the C ABI's (include/krotov_hip.h): operators are N*N
 * complex128, COLUMN-major interleaved (re,im) = Julia Matrix{ComplexF64};
 * states [K][N]; storages [(NT+1)][K][N]; pulses [L][NT].
 *
 * mode bit 0: 0 = as the reference executes it (one expm per trajectory per step,
 *                 K-fold redundant when trajectories share H) -- "CPU-a/b"
 *             1 = one shared propagator per step                -- "CPU-c"
 * nthreads  : OpenMP threads over trajectories in the sweeps the reference marks
 *             @threadsif (optimize.jl:182,303,360); the du sum stays serial (:340-349).
 */
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef double complex zc;

#define A_(M, i, j) (M)[(size_t)(i) + (size_t)(j) * n]

static void zgemm_nn(int n, const zc *A, const zc *B, zc *C) {
    /* C = A*B, column-major */
    for (int j = 0; j < n; ++j) {
        for (int i = 0; i < n; ++i) A_(C, i, j) = 0;
        for (int p = 0; p < n; ++p) {
            zc b = A_(B, p, j);
            for (int i = 0; i < n; ++i) A_(C, i, j) += A_(A, i, p) * b;
        }
    }
}

static double norm1(int n, const zc *A) {
    double m = 0;
    for (int j = 0; j < n; ++j) {
        double s = 0;
        for (int i = 0; i < n; ++i) s += cabs(A_(A, i, j));
        if (s > m) m = s;
    }
    return m;
}

/* Solve P X = Q in place (X overwrites Q); P destroyed. Partial pivoting. */
static int zsolve(int n, zc *P, zc *Q) {
    for (int c = 0; c < n; ++c) {
        int piv = c;
        double best = cabs(A_(P, c, c));
        for (int r = c + 1; r < n; ++r) {
            double v = cabs(A_(P, r, c));
            if (v > best) { best = v; piv = r; }
        }
        if (best == 0.0) return -1;
        if (piv != c) {
            for (int j = 0; j < n; ++j) {
                zc t = A_(P, c, j); A_(P, c, j) = A_(P, piv, j); A_(P, piv, j) = t;
                t = A_(Q, c, j); A_(Q, c, j) = A_(Q, piv, j); A_(Q, piv, j) = t;
            }
        }
        zc inv = 1.0 / A_(P, c, c);
        for (int r = c + 1; r < n; ++r) {
            zc f = A_(P, r, c) * inv;
            if (f == 0) continue;
            for (int j = c; j < n; ++j) A_(P, r, j) -= f * A_(P, c, j);
            for (int j = 0; j < n; ++j) A_(Q, r, j) -= f * A_(Q, c, j);
        }
    }
    for (int j = 0; j < n; ++j) {
        for (int r = n - 1; r >= 0; --r) {
            zc s = A_(Q, r, j);
            for (int c = r + 1; c < n; ++c) s -= A_(P, r, c) * A_(Q, c, j);
            A_(Q, r, j) = s / A_(P, r, r);
        }
    }
    return 0;
}

/* E = exp(A).  Higham (2005) scaling and squaring, Pade orders 3,5,7,9,13.  W: caller's workspace of 7 n*n complex. */
static int expm_ws(int n, const zc *A, zc *E, zc *W) {
    static const double b3[] = {120., 60., 12., 1.};
    static const double b5[] = {30240., 15120., 3360., 420., 30., 1.};
    static const double b7[] = {17297280., 8648640., 1995840., 277200., 25200., 1512., 56., 1.};
    static const double b9[] = {17643225600., 8821612800., 2075673600., 302702400., 30270240.,
                                2162160., 110880., 3960., 90., 1.};
    static const double b13[] = {64764752532480000., 32382376266240000., 7771770303897600.,
                                 1187353796428800., 129060195264000., 10559470521600.,
                                 670442572800., 33522128640., 1323241920., 40840800., 960960.,
                                 16380., 182., 1.};
    static const double theta[] = {1.495585217958292e-2, 2.539398330063230e-1,
                                   9.504178996162932e-1, 2.097847961257068e0,
                                   5.371920351148152e0};
    size_t nn = (size_t)n * n;
    zc *A2 = W, *A4 = W + nn, *A6 = W + 2 * nn, *U = W + 3 * nn, *V = W + 4 * nn,
       *T = W + 5 * nn, *As = W + 6 * nn;
    double nA = norm1(n, A);
    int s = 0, rc = 0;
    memcpy(As, A, sizeof(zc) * nn);
    if (nA <= theta[3]) {
        const double *b;
        int m;
        if (nA <= theta[0]) { b = b3; m = 3; }
        else if (nA <= theta[1]) { b = b5; m = 5; }
        else if (nA <= theta[2]) { b = b7; m = 7; }
        else { b = b9; m = 9; }
        /* U = A * sum_{k odd} b_k A^{k-1},  V = sum_{k even} b_k A^k */
        zgemm_nn(n, As, As, A2);
        zc *P = A4, *Pn = A6; /* running even power, ping-pong */
        for (size_t e = 0; e < nn; ++e) { T[e] = 0; V[e] = 0; }
        for (int i = 0; i < n; ++i) { A_(T, i, i) = b[1]; A_(V, i, i) = b[0]; }
        /* P = A^2 */
        memcpy(P, A2, sizeof(zc) * nn);
        for (int k = 1; 2 * k <= m; ++k) {
            for (size_t e = 0; e < nn; ++e) { T[e] += b[2 * k + 1] * P[e]; V[e] += b[2 * k] * P[e]; }
            if (2 * (k + 1) <= m) { zgemm_nn(n, P, A2, Pn); zc *t = P; P = Pn; Pn = t; }
        }
        zgemm_nn(n, As, T, U);
    } else {
        double r = nA / theta[4];
        if (r > 1.0) { s = (int)ceil(log2(r)); if (s < 0) s = 0; }
        double sc = ldexp(1.0, -s);
        for (size_t e = 0; e < nn; ++e) As[e] *= sc;
        zgemm_nn(n, As, As, A2);
        zgemm_nn(n, A2, A2, A4);
        zgemm_nn(n, A4, A2, A6);
        const double *b = b13;
        /* U = A*(A6*(b13 A6 + b11 A4 + b9 A2) + b7 A6 + b5 A4 + b3 A2 + b1 I) */
        for (size_t e = 0; e < nn; ++e) T[e] = b[13] * A6[e] + b[11] * A4[e] + b[9] * A2[e];
        zgemm_nn(n, A6, T, V);
        for (size_t e = 0; e < nn; ++e) V[e] += b[7] * A6[e] + b[5] * A4[e] + b[3] * A2[e];
        for (int i = 0; i < n; ++i) A_(V, i, i) += b[1];
        zgemm_nn(n, As, V, U);
        /* V = A6*(b12 A6 + b10 A4 + b8 A2) + b6 A6 + b4 A4 + b2 A2 + b0 I */
        for (size_t e = 0; e < nn; ++e) T[e] = b[12] * A6[e] + b[10] * A4[e] + b[8] * A2[e];
        zgemm_nn(n, A6, T, V);
        for (size_t e = 0; e < nn; ++e) V[e] += b[6] * A6[e] + b[4] * A4[e] + b[2] * A2[e];
        for (int i = 0; i < n; ++i) A_(V, i, i) += b[0];
    }
    /* E = (V-U)^{-1} (V+U) */
    for (size_t e = 0; e < nn; ++e) { T[e] = V[e] - U[e]; E[e] = V[e] + U[e]; }
    if (zsolve(n, T, E) != 0) rc = -1;
    for (int q = 0; q < s && rc == 0; ++q) {
        zgemm_nn(n, E, E, T);
        memcpy(E, T, sizeof(zc) * nn);
    }
    return rc;
}

int orc_expm(int n, const double *A_in, double *E_out) {
    zc *W = (zc *)malloc(sizeof(zc) * (size_t)n * n * 7);
    if (!W) return -2;
    int rc = expm_ws(n, (const zc *)A_in, (zc *)E_out, W);
    free(W);
    return rc;
}

/* H(eps_n) = H0 + sum_l eps_l[n] H_l ; A = (-i*sdt) * H  or, for adj, (-i*sdt) * H^dagger */
static void build_arg(int n, int L, const zc *H0, const zc *Hc, const double *eps, int NT, int step,
                      double sdt, int adj, zc *A) {
    size_t nn = (size_t)n * n;
    for (size_t e = 0; e < nn; ++e) {
        zc h = H0[e];
        for (int l = 0; l < L; ++l) h += eps[(size_t)l * NT + step] * Hc[(size_t)l * nn + e];
        A[e] = h;
    }
    if (adj) {
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < j; ++i) {
                zc a = conj(A_(A, i, j)), b = conj(A_(A, j, i));
                A_(A, i, j) = b; A_(A, j, i) = a;
            }
        for (int i = 0; i < n; ++i) A_(A, i, i) = conj(A_(A, i, i));
    }
    zc f = -I * sdt;
    for (size_t e = 0; e < nn; ++e) A[e] *= f;
}

static void matvec(int n, const zc *U, const zc *x, zc *y) {
    for (int i = 0; i < n; ++i) y[i] = 0;
    for (int j = 0; j < n; ++j) {
        zc b = x[j];
        for (int i = 0; i < n; ++i) y[i] += A_(U, i, j) * b;
    }
}

/* One prop_step! for every trajectory (optimize.jl:257,309,361). */
static int step_all(int n, int NT, int K, int L, const zc *H0, const zc *Hc, const double *eps,
                    int step, double sdt, int adj, int mode, int nthreads, zc *state /*[K][n]*/) {
    size_t nn = (size_t)n * n;
    int rc = 0;
    if (mode & 1) {
        zc *A = (zc *)malloc(sizeof(zc) * nn * 2);
        zc *U = A + nn;
        build_arg(n, L, H0, Hc, eps, NT, step, sdt, adj, A);
        rc = orc_expm(n, (double *)A, (double *)U);
#pragma omp parallel for num_threads(nthreads) schedule(static)
        for (int k = 0; k < K; ++k) {
            zc y[n];
            matvec(n, U, state + (size_t)k * n, y);
            memcpy(state + (size_t)k * n, y, sizeof(zc) * n);
        }
        free(A);
    } else {
        /* one workspace per THREAD (argument, propagator, 7 Pade matrices, result vector), not one malloc per trajectory per
         * step: the allocator's lock was what limited the threaded variant (round-2 review) */
#pragma omp parallel num_threads(nthreads)
        {
            zc *A = (zc *)malloc(sizeof(zc) * (nn * 9 + (size_t)n));
            zc *U = A + nn, *W = A + 2 * nn, *y = A + 9 * nn;
#pragma omp for schedule(static)
            for (int k = 0; k < K; ++k) {
                if (!A) {
#pragma omp atomic write
                    rc = -2;
                    continue;
                }
                build_arg(n, L, H0, Hc, eps, NT, step, sdt, adj, A);
                if (expm_ws(n, A, U, W) != 0) {
#pragma omp atomic write
                    rc = -1;
                }
                matvec(n, U, state + (size_t)k * n, y);
                memcpy(state + (size_t)k * n, y, sizeof(zc) * n);
            }
            free(A);
        }
    }
    return rc;
}

int orc_forward(int N, int NT, int K, int L, const double *tlist, const double *H0,
                const double *Hc, const double *eps, const double *psi0, int mode, int nthreads,
                double *Phi_out /* nullable [(NT+1)][K][N] */, double *psiT_out /* [K][N] */) {
    size_t sl = (size_t)K * N;
    zc *psi = (zc *)malloc(sizeof(zc) * sl);
    if (nthreads < 1) nthreads = 1;
    memcpy(psi, psi0, sizeof(zc) * sl);
    if (Phi_out) memcpy(Phi_out, psi, sizeof(zc) * sl);
    int rc = 0;
    for (int n = 0; n < NT && rc == 0; ++n) {
        rc = step_all(N, NT, K, L, (const zc *)H0, (const zc *)Hc, eps, n, tlist[n + 1] - tlist[n], 0,
                      mode, nthreads, psi);
        if (Phi_out) memcpy((zc *)Phi_out + (size_t)(n + 1) * sl, psi, sizeof(zc) * sl);
    }
    memcpy(psiT_out, psi, sizeof(zc) * sl);
    free(psi);
    return rc;
}

int orc_backward(int N, int NT, int K, int L, const double *tlist, const double *H0,
                 const double *Hc, const double *eps, const double *chiT, int mode, int nthreads,
                 double *X_out /* [(NT+1)][K][N] */) {
    size_t sl = (size_t)K * N;
    zc *chi = (zc *)malloc(sizeof(zc) * sl);
    if (nthreads < 1) nthreads = 1;
    memcpy(chi, chiT, sizeof(zc) * sl);
    memcpy((zc *)X_out + (size_t)NT * sl, chi, sizeof(zc) * sl);
    int rc = 0;
    for (int n = NT - 1; n >= 0 && rc == 0; --n) {
        rc = step_all(N, NT, K, L, (const zc *)H0, (const zc *)Hc, eps, n, -(tlist[n + 1] - tlist[n]),
                      1, mode, nthreads, chi);
        memcpy((zc *)X_out + (size_t)n * sl, chi, sizeof(zc) * sl);
    }
    free(chi);
    return rc;
}

int orc_update_forward(int N, int NT, int K, int L, const double *tlist, const double *H0,
                       const double *Hc, const unsigned char *mask /* nullable [L] */,
                       const double *eps_guess, const double *S, const double *lam,
                       const double *psi0, const double *X, int mode, int nthreads,
                       double *eps_new /* [L][NT] */, double *g_a_int /* [L] */,
                       double *Phi_out /* nullable */, double *psiT_out) {
    size_t sl = (size_t)K * N, nn = (size_t)N * N;
    const zc *Hcz = (const zc *)Hc;
    const zc *Xz = (const zc *)X;
    zc *psi = (zc *)malloc(sizeof(zc) * sl);
    double *terms = (double *)malloc(sizeof(double) * (size_t)K);
    if (nthreads < 1) nthreads = 1;
    memcpy(psi, psi0, sizeof(zc) * sl);
    memcpy(eps_new, eps_guess, sizeof(double) * (size_t)L * NT);
    for (int l = 0; l < L; ++l) g_a_int[l] = 0.0;
    if (Phi_out) memcpy(Phi_out, psi, sizeof(zc) * sl);
    int rc = 0;
    for (int n = 0; n < NT && rc == 0; ++n) {
        double dt = tlist[n + 1] - tlist[n];
        for (int l = 0; l < L; ++l) { /* optimize.jl:340-349: l outer, k inner, serial */
            double du = 0.0;
            if (!mask || mask[l]) {
                /* the K terms Im<chi_k|mu_l|psi_k> are independent: they may be EVALUATED by several threads (each one
                 * exactly as the serial loop would), but they are ADDED in the reference's order k = 0..K-1 */
#pragma omp parallel for num_threads(nthreads) schedule(static)
                for (int k = 0; k < K; ++k) {
                    zc t[N];
                    matvec(N, Hcz + (size_t)l * nn, psi + (size_t)k * N, t);
                    zc d = 0;
                    const zc *chi = Xz + (size_t)n * sl + (size_t)k * N;
                    for (int i = 0; i < N; ++i) d += conj(chi[i]) * t[i];
                    terms[k] = cimag(d);
                }
                for (int k = 0; k < K; ++k) du += terms[k];
            }
            double alpha = S[(size_t)l * NT + n] / lam[l]; /* optimize.jl:351-358 */
            eps_new[(size_t)l * NT + n] = eps_guess[(size_t)l * NT + n] + alpha * du;
            g_a_int[l] += alpha * fabs(du) * fabs(du) * dt;
        }
        rc = step_all(N, NT, K, L, (const zc *)H0, Hcz, eps_new, n, dt, 0, mode, nthreads, psi);
        if (Phi_out) memcpy((zc *)Phi_out + (size_t)(n + 1) * sl, psi, sizeof(zc) * sl);
    }
    memcpy(psiT_out, psi, sizeof(zc) * sl);
    free(psi);
    free(terms);
    return rc;
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
