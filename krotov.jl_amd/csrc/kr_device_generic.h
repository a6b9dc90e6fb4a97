// Generic sweep kernels: any state dimension N, any number of controls L, any number of trajectories K, and NO
// co-residency requirement -- ordinary grids, one launch per sweep (known pulses) or a few launches per time step (update
// sweep).  They are
//   * the multi-pass path behind the capacity limits of the persistent kernels (N > 256, L > 4, more trajectories than
//     co-resident workgroups can carry, or a device on which the persistent grid cannot be resident at once), and
//   * the carrier of the per-step exchanges that cannot run inside a kernel: the RCCL all-reduce of the L partial sums
//     that BASELINE.json's north_star names as the multi-GPU baseline, and a host callback (MPI, gloo, ...).
// Update sweep, per step n (/root/reference/src/optimize.jl:328-370):
//     kr_gen_partial   partial[tile][l] = sum_{k in tile} Im<chi_k(t_n)|H_l|psi_k(t_n)>            (:339-349)
//     kr_gen_reduce    du_local[l] = sum_tiles partial (fixed order)
//     [exchange]       du_total = sum over ranks (ncclAllReduce on the stream / host callback / none)
//     kr_gen_form      eps_new[l][n] = eps[l][n] + alpha du, g_a_int += ...; Heps = H0 + sum eps_new_l H_l (:351-358)
//     kr_gen_apply     psi_k <- exp(-i Heps dt) psi_k (Taylor/Horner, same plan as the persistent kernels)  (:360-368)
// Operators are read column-major exactly as the caller (Julia) hands them over: thread i owns row i, so the loads of a
// column are coalesced; the adjoint pack (backward sweep, workspace.jl:69) is a second column-major copy of H^dagger.
// A workgroup of 256 threads carries KTG trajectories (each operator element is loaded once for all of them); with N < 256
// the JQ = 256 / RP thread groups split the columns and their partial row sums meet in LDS in a fixed order.
#pragma once
#include "kr_common.h"

// non-linear controls: amplitude f_l(x) and derivative f_l'(x) of control l (Horner)
__device__ __forceinline__ double kr_gen_f(const KrGenArgs &a, const int l, const double x) {
    if (!a.poly) return x;
    const double *cf = a.poly + (size_t)l * (a.poly_deg + 1);
    double v = cf[a.poly_deg];
    for (int p = a.poly_deg - 1; p >= 0; --p) v = fma(v, x, cf[p]);
    return v;
}
__device__ __forceinline__ double kr_gen_fprime(const KrGenArgs &a, const int l, const double x) {
    if (!a.poly) return 1.0;
    const double *cf = a.poly + (size_t)l * (a.poly_deg + 1);
    double v = a.poly_deg * cf[a.poly_deg];
    for (int p = a.poly_deg - 1; p >= 1; --p) v = fma(v, x, p * cf[p]);
    return v;
}

__device__ __forceinline__ void kr_gen_plan(const KrGenArgs &a, const double theta, int &s, int &m) {
    if (theta <= KR_THETA_MAX) {
        s = 1;
        m = 1;
        while (m < KR_MAX_ORDER - 1 && theta > a.thr[m]) ++m;   // thr is increasing in m: same plan as kr_plan_fast
    } else {
        kr_taylor_plan(theta, a.tol, s, m);
    }
}

// out[q][i] = sum_j (sum_m c_m M_m[i][j]) x[q][j] for the rows i = this thread's row (+ RP, ...), all KTG states q.
// NMAT matrices M_m (consecutive N*N blocks at Hb) with real coefficients c_m; x and out are LDS vectors [KTG][N];
// red: LDS scratch [JQ][KTG][RP].  All threads call it; ends with a barrier.
template <int KTG>
__device__ __forceinline__ void kr_gen_matvec(const KrGenArgs &a, const double2 *__restrict__ Hb, const int nmat, const double *c,
                                              const double2 *x, double2 *out, double2 *red) {
    const int N = a.N, RP = a.RP, JQ = a.JQ, tid = threadIdx.x;
    const int i0 = tid % RP, jq = tid / RP;
    const size_t nn = (size_t)N * N;
    for (int ib = 0; ib < N; ib += RP) {
        const int i = ib + i0;
        double sr[KTG], si[KTG];
#pragma unroll
        for (int q = 0; q < KTG; ++q) { sr[q] = 0.0; si[q] = 0.0; }
        if (i < N && jq < JQ) {
            for (int j = jq; j < N; j += JQ) {
                double hr = 0.0, hi = 0.0;
                for (int mm = 0; mm < nmat; ++mm) {
                    const double2 h = Hb[(size_t)mm * nn + i + (size_t)j * N];
                    hr = fma(c[mm], h.x, hr);
                    hi = fma(c[mm], h.y, hi);
                }
#pragma unroll
                for (int q = 0; q < KTG; ++q) {
                    const double2 xv = x[q * N + j];
                    sr[q] = fma(hr, xv.x, fma(-hi, xv.y, sr[q]));
                    si[q] = fma(hr, xv.y, fma(hi, xv.x, si[q]));
                }
            }
        }
        if (JQ > 1) {
            if (i0 < RP && jq < JQ) {
#pragma unroll
                for (int q = 0; q < KTG; ++q) red[(jq * KTG + q) * RP + i0] = make_double2(sr[q], si[q]);
            }
            __syncthreads();
            if (jq == 0 && i < N) {
#pragma unroll
                for (int q = 0; q < KTG; ++q) {
                    double tr = 0.0, ti = 0.0;
                    for (int g = 0; g < JQ; ++g) { const double2 v = red[(g * KTG + q) * RP + i0]; tr += v.x; ti += v.y; }
                    out[q * N + i] = make_double2(tr, ti);
                }
            }
            __syncthreads();
        } else if (i < N) {
#pragma unroll
            for (int q = 0; q < KTG; ++q) out[q * N + i] = make_double2(sr[q], si[q]);
        }
    }
    __syncthreads();
}

// psi (LDS, [KTG][N]) <- exp(-i sdt sum_m c_m M_m) psi; y0, y1, y2: LDS scratch vectors.  Three step propagators
// (kr_config.prop_method, the reference's prop_method: /root/reference/src/optimize.jl:108,137-153):
//   Taylor     Horner form, order and sub-steps from the norm bound theta of THIS step (kr_taylor_plan)
//   Chebyshev  (Hermitian generators) exp(-i A) = sum_k (2 - d_k0) (-i)^k J_k(th) T_k(A / th) by Clenshaw's recurrence;
//              th = |dt| * Lambda with Lambda a bound of the spectral radius over the WHOLE control range the sweep was
//              initialised for (update sweep: the guess range widened 5x, as transform_control_ranges does, optimize.jl:238-244)
//   Newton     interpolation polynomial of exp(-i z) at Leja points of [-th, th] in Newton form,
//              p(A) psi = sum_k d_k prod_{j<k} (A - z_j) psi (divided differences d_k from the host, long double)
// Per-step tables (host, a.ptab + step * a.ptab_stride): [0] degree d, [1] sub-steps s, [2] th of a sub-step, then
// Chebyshev: J_0..J_d; Newton: z_0..z_{d-1}, then (re, im) of d_0..d_d.
template <int KTG>
__device__ __forceinline__ void kr_gen_propagate(const KrGenArgs &a, const double2 *Hb, const int nmat, const double *c,
                                                 const double theta, const double sdt, const int step, double2 *psi, double2 *y0,
                                                 double2 *y1, double2 *y2, double2 *red, const int gen = 0) {
    const int tot = KTG * a.N;
    if (a.method == KR_METHOD_CHEBY || a.method == KR_METHOD_NEWTON) {
        const double *tab = a.ptab + (size_t)step * a.ptab_stride;
        const int d = (int)tab[0], s = (int)tab[1];
        const double th = tab[2];
        const double fs = sdt / (double)s;          // A = fs * H per sub-step, spectrum within [-th, th]
        for (int sub = 0; sub < s; ++sub) {
            if (a.method == KR_METHOD_CHEBY) {
                // Clenshaw: b_k = c_k psi + 2 At b_{k+1} - b_{k+2}, At = A / th; result = c_0 psi + At b_1 - b_2
                // vectors: y0 = b_{k+1}, y2 = b_{k+2}, y1 = H b_{k+1}
                const double sc = fs / th;
                for (int e = threadIdx.x; e < tot; e += blockDim.x) { y0[e] = make_double2(0.0, 0.0); y2[e] = make_double2(0.0, 0.0); }
                __syncthreads();
                for (int k = d; k >= 0; --k) {
                    const double jk = tab[3 + k] * (k == 0 ? 1.0 : 2.0);
                    // c_k = jk * (-i)^k
                    const int q = k & 3;
                    const double cr = (q == 0) ? jk : ((q == 2) ? -jk : 0.0), ci = (q == 1) ? -jk : ((q == 3) ? jk : 0.0);
                    if (k < d) kr_gen_matvec<KTG>(a, Hb, nmat, c, y0, y1, red);   // (k = d: b_{d+1} = 0)
                    const double two = (k == 0) ? 1.0 : 2.0;
                    for (int e = threadIdx.x; e < tot; e += blockDim.x) {
                        const double2 p = psi[e], b2 = y2[e];
                        double2 hv = make_double2(0.0, 0.0);
                        if (k < d) hv = y1[e];
                        const double nr = cr * p.x - ci * p.y + two * sc * hv.x - b2.x;
                        const double ni = cr * p.y + ci * p.x + two * sc * hv.y - b2.y;
                        y2[e] = y0[e];                       // b_{k+2} <- b_{k+1}
                        y0[e] = make_double2(nr, ni);        // b_k  (k = 0: the result)
                    }
                    __syncthreads();
                }
                for (int e = threadIdx.x; e < tot; e += blockDim.x) psi[e] = y0[e];
                __syncthreads();
            } else {
                // Newton: w_0 = psi, acc = d_0 w_0; w_{k+1} = (A - z_k) w_k = fs H w_k - z_k w_k; acc += d_{k+1} w_{k+1}
                const double *z = tab + 3, *dd = tab + 3 + d;
                for (int e = threadIdx.x; e < tot; e += blockDim.x) {
                    const double2 p = psi[e];
                    y0[e] = p;
                    y2[e] = make_double2(dd[0] * p.x - dd[1] * p.y, dd[0] * p.y + dd[1] * p.x);
                }
                __syncthreads();
                for (int k = 0; k < d; ++k) {
                    kr_gen_matvec<KTG>(a, Hb, nmat, c, y0, y1, red);
                    const double zr = z[k], dr = dd[2 * (k + 1)], di = dd[2 * (k + 1) + 1];
                    for (int e = threadIdx.x; e < tot; e += blockDim.x) {
                        const double2 hv = y1[e], w = y0[e], acc = y2[e];
                        const double wr = fs * hv.x - zr * w.x, wi = fs * hv.y - zr * w.y;
                        y0[e] = make_double2(wr, wi);
                        y2[e] = make_double2(acc.x + dr * wr - di * wi, acc.y + dr * wi + di * wr);
                    }
                    __syncthreads();
                }
                for (int e = threadIdx.x; e < tot; e += blockDim.x) psi[e] = y2[e];
                __syncthreads();
            }
        }
        return;
    }
    int s, m;
    kr_gen_plan(a, theta, s, m);
    // economized step polynomial (kr_device.h: KrEcon; Hermitian generators): the step's row where it covers theta and saves a degree
    const double *row = a.econ ? a.econ + ((size_t)gen * a.NT + step) * KR_ECON_W : nullptr;
    bool eco = false;
    if (row) {
        const int me = (int)row[31];
        if (me > 0 && theta <= row[30] && (s > 1 || me < m)) { eco = true; s = 1; m = me; }
    }
    const double fs = (s == 1) ? sdt : sdt / (double)s;
    for (int sub = 0; sub < s; ++sub) {
        const double2 *src = psi;
        for (int j = m; j >= 1; --j) {   // y <- psi + (-i fs g_j) H y  (g_j = 1 / j; economized: a_j / a_j-1, last round a_0 psi + a_1 z y_1)
            kr_gen_matvec<KTG>(a, Hb, nmat, c, src, y1, red);
            const double f = fs * (eco ? row[j] : kr_rinv[j]);
            const double c0 = (eco && j == 1) ? row[0] : 1.0;
            for (int e = threadIdx.x; e < tot; e += blockDim.x) {
                const double2 hv = y1[e], p = psi[e];
                y0[e] = make_double2(fma(f, hv.y, c0 * p.x), fma(-f, hv.x, c0 * p.y));
            }
            __syncthreads();
            src = y0;
        }
        for (int e = threadIdx.x; e < tot; e += blockDim.x) psi[e] = y0[e];
        __syncthreads();
    }
}

#define KR_GEN_LDS(KTG) \
    extern __shared__ __attribute__((aligned(16))) double2 kr_gen_lds[]; \
    double2 *psi = kr_gen_lds, *y0 = psi + (KTG) * a.N, *y1 = y0 + (KTG) * a.N, *y2 = y1 + (KTG) * a.N, *red = y2 + (KTG) * a.N;

// known-pulse sweep, one launch: every workgroup runs its tile through all N_T steps (no inter-workgroup dependency)
template <int KTG>
__global__ __launch_bounds__(256) void kr_gen_known(const KrGenArgs a) {
    KR_GEN_LDS(KTG)
    const int tile = blockIdx.x, N = a.N, NT = a.NT, L = a.L, tot = KTG * N;
    const int gen = a.tile_gen ? a.tile_gen[tile] : 0;
    const double2 *Hb = a.H + (size_t)gen * a.gen_stride;
    const size_t tb = (size_t)tile * KTG * N, slab = (size_t)a.Kp * N;
    for (int e = threadIdx.x; e < tot; e += blockDim.x) {
        const double2 v = a.init[tb + e];
        psi[e] = v;
        if (a.store) a.store[(size_t)((a.dir > 0) ? 0 : NT) * slab + tb + e] = v;
    }
    __syncthreads();
    double c[1 + 32];
    for (int step = 0; step < NT; ++step) {
        const int n = (a.dir > 0) ? step : (NT - 1 - step);
        const double sdt = (a.dir > 0) ? a.dt[n] : -a.dt[n];
        double theta = a.nrm[(size_t)gen * (1 + L)];
        c[0] = 1.0;
        for (int l = 0; l < L; ++l) { c[1 + l] = kr_gen_f(a, l, a.eps[(size_t)l * NT + n]); theta += fabs(c[1 + l]) * a.nrm[(size_t)gen * (1 + L) + 1 + l]; }
        kr_gen_propagate<KTG>(a, Hb, 1 + L, c, theta * fabs(sdt), sdt, n, psi, y0, y1, y2, red, gen);
        if (a.store) {
            const int slot = (a.dir > 0) ? (n + 1) : n;
            for (int e = threadIdx.x; e < tot; e += blockDim.x) a.store[(size_t)slot * slab + tb + e] = psi[e];
        }
    }
    if (a.fin) for (int e = threadIdx.x; e < tot; e += blockDim.x) a.fin[tb + e] = psi[e];
}

// partial sums of the update integrand of step n over this tile's trajectories: psi(t_n) is in LDS (`psi`), chi(t_n) is fetched here
template <int KTG>
__device__ __forceinline__ void kr_gen_partial_body(const KrGenArgs &a, const int n, const int tile, const int gen, double2 *psi, double2 *y0,
                                                    double2 *y1, double2 *red, double *sred) {
    const int N = a.N, L = a.L, tot = KTG * N;
    const size_t tb = (size_t)tile * KTG * N, slab = (size_t)a.Kp * N;
    for (int e = threadIdx.x; e < tot; e += blockDim.x) y0[e] = a.X[(size_t)n * slab + tb + e];   // chi(t_n)
    __syncthreads();
    const double one = 1.0;
    for (int l = 0; l < L; ++l) {
        double acc = 0.0;
        if ((a.present[gen] >> l) & 1u) {
            kr_gen_matvec<KTG>(a, a.H + (size_t)gen * a.gen_stride + (size_t)(1 + l) * N * N, 1, &one, psi, y1, red);
            for (int e = threadIdx.x; e < tot; e += blockDim.x) {
                const double2 ch = y0[e], av = y1[e];
                acc += ch.x * av.y - ch.y * av.x;   // Im(conj(chi) a)
            }
        }
        sred[threadIdx.x] = acc;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {   // fixed tree: bitwise reproducible
            if ((int)threadIdx.x < o) sred[threadIdx.x] += sred[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) a.partial[(size_t)tile * L + l] = sred[0];
        __syncthreads();
    }
}
// update sweep, step a.n: the partial sums as a kernel of their own (first step; every step when the sum crosses ranks)
template <int KTG>
__global__ __launch_bounds__(256) void kr_gen_partial(const KrGenArgs a) {
    KR_GEN_LDS(KTG)
    __shared__ double sred[256];
    const int tile = blockIdx.x, N = a.N, tot = KTG * N;
    const int gen = a.tile_gen ? a.tile_gen[tile] : 0;
    const size_t tb = (size_t)tile * KTG * N;
    for (int e = threadIdx.x; e < tot; e += blockDim.x) psi[e] = a.state[tb + e];
    kr_gen_partial_body<KTG>(a, a.n, tile, gen, psi, y0, y1, red, sred);
}

// du_local[l] = sum over tiles, fixed order (one workgroup)
__global__ __launch_bounds__(256) void kr_gen_reduce(const KrGenArgs a) {
    __shared__ double sred[256];
    for (int l = 0; l < a.L; ++l) {
        double acc = 0.0;
        for (int t = threadIdx.x; t < a.tiles; t += 256) acc += a.partial[(size_t)t * a.L + l];
        sred[threadIdx.x] = acc;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) sred[threadIdx.x] += sred[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) a.du_local[l] = sred[0];
        __syncthreads();
    }
}

// pulse update of step a.n (identical on every rank: du_total is) and H(eps_new) for every generator
__global__ __launch_bounds__(256) void kr_gen_form(const KrGenArgs a) {
    const int N = a.N, NT = a.NT, L = a.L, n = a.n;
    const size_t nn = (size_t)N * N, idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    // a.fused (one rank, device-side sum; round 3): every block sums the tiles' partials ITSELF, in the order and tree of
    // kr_gen_reduce (bitwise the same totals in every block), instead of waiting for a kernel of one workgroup to do it --
    // the stepped sweep is a chain of dependent launches, and this takes one of four out of every time step
    __shared__ double sdu[32];
    if (a.fused) {
        __shared__ double sred[256];
        for (int l = 0; l < L; ++l) {
            double acc = 0.0;
            for (int t = threadIdx.x; t < a.tiles; t += 256) acc += a.partial[(size_t)t * L + l];
            sred[threadIdx.x] = acc;
            __syncthreads();
            for (int o = 128; o > 0; o >>= 1) {
                if ((int)threadIdx.x < o) sred[threadIdx.x] += sred[threadIdx.x + o];
                __syncthreads();
            }
            if (threadIdx.x == 0) sdu[l] = sred[0];
            __syncthreads();
        }
        if (blockIdx.x == 0 && threadIdx.x < L) a.du_local[threadIdx.x] = sdu[threadIdx.x];
    }
    if (idx >= nn * a.G) return;
    const int gen = (int)(idx / nn);
    const size_t e = idx % nn;
    const double2 *Hb = a.H + (size_t)gen * a.gen_stride;
    double2 h = Hb[e];
    for (int l = 0; l < L; ++l) {
        const double al = a.alpha[(size_t)l * NT + n], eg = a.eps[(size_t)l * NT + n];
        // mu_l = f_l'(eps_guess) H_l (the derivative is evaluated at the guess value, optimize.jl:337,345)
        const double dut = a.fused ? sdu[l] : a.du_total[l];
        const double du = a.poly ? __dmul_rn(kr_gen_fprime(a, l, eg), dut) : dut;
        const double en = __dadd_rn(eg, __dmul_rn(al, du));   // optimize.jl:355, no FMA contraction
        const double2 hl = Hb[(size_t)(1 + l) * nn + e];
        const double amp = kr_gen_f(a, l, en);
        h.x = fma(amp, hl.x, h.x);
        h.y = fma(amp, hl.y, h.y);
        if (idx == 0) {
            // Chebyshev / Newton were initialised for |eps_l| <= erange[l] (guess range x 5): an update that leaves it would be
            // propagated with a polynomial that no longer covers the spectrum -- report instead of returning wrong states
            if (a.erange && fabs(en) > a.erange[l] && a.status) { a.status[1] = 5u; a.status[2] = (unsigned)n; a.status[0] = 1u; }
            a.eps_new[(size_t)l * NT + n] = en;
            const double ad = fabs(du);
            a.g_a_int[l] = __dadd_rn(a.g_a_int[l], __dmul_rn(__dmul_rn(al, __dmul_rn(ad, ad)), a.dt[n]));   // optimize.jl:356
        }
    }
    a.Heps[(size_t)gen * nn + e] = h;
}

// forward step of the update sweep under the new pulse
template <int KTG>
__global__ __launch_bounds__(256) void kr_gen_apply(const KrGenArgs a) {
    KR_GEN_LDS(KTG)
    const int tile = blockIdx.x, N = a.N, NT = a.NT, L = a.L, n = a.n, tot = KTG * N;
    const int gen = a.tile_gen ? a.tile_gen[tile] : 0;
    const size_t tb = (size_t)tile * KTG * N, slab = (size_t)a.Kp * N;
    for (int e = threadIdx.x; e < tot; e += blockDim.x) psi[e] = a.state[tb + e];
    __syncthreads();
    double theta = a.nrm[(size_t)gen * (1 + L)];
    for (int l = 0; l < L; ++l) theta += fabs(kr_gen_f(a, l, a.eps_new[(size_t)l * NT + n])) * a.nrm[(size_t)gen * (1 + L) + 1 + l];
    const double one = 1.0, dtn = a.dt[n];
    kr_gen_propagate<KTG>(a, a.Heps + (size_t)gen * N * N, 1, &one, theta * fabs(dtn), dtn, n, psi, y0, y1, y2, red, gen);
    for (int e = threadIdx.x; e < tot; e += blockDim.x) {
        a.state[tb + e] = psi[e];
        if (a.store) a.store[(size_t)(n + 1) * slab + tb + e] = psi[e];
        if (a.fin && n == NT - 1) a.fin[tb + e] = psi[e];
    }
    // a.fused: the partial sums of step n + 1 from the state this workgroup has just produced (still in LDS): the next
    // launch is the pulse update, not a kernel that reloads psi to do this
    if (a.fused && n + 1 < NT) {
        __shared__ double sred[256];
        __syncthreads();
        kr_gen_partial_body<KTG>(a, n + 1, tile, gen, psi, y0, y1, red, sred);
    }
}
#undef KR_GEN_LDS
