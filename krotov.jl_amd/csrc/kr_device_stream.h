// Streamed-generator variant of the sweep kernels for 64 < N <= 256 (NP in {128, 256}).
//
// The generator does not fit the register file (N = 256: 1 MB per matrix), so every matvec round streams
// the thread-major packs of H0, H_1..H_L from L2/HBM (coalesced 16-B loads) and combines
// H(eps) = H0 + sum_l eps_l H_l on the fly.  One workgroup of 4*NP threads (512 or 1024) carries KT
// trajectories; each loaded generator element is reused for all KT states.  Same algorithm, same
// exchange, same numerics as kr_device.h -- only the operand source differs.  This is the functional
// large-N path; an MFMA / row-split version is the next step.
#pragma once
#include "kr_device.h"

typedef unsigned int kr_su4 __attribute__((ext_vector_type(4)));
template <int NP, int LMAX, int KT>
struct KrGeomS {
    static constexpr int JQ = 4;
    static constexpr int THREADS = NP * JQ;  // 512 or 1024
    static constexpr int NW = THREADS / 64;
    static constexpr int JS = NP / JQ;       // 32 or 64 columns per thread
    static constexpr int RT = THREADS;
    static constexpr int KTILE = KT;
    static constexpr int NPP = NP + JQ;
    static constexpr int ROWS = NP;
    static_assert(NP == 128 || NP == 256, "NP in {128,256}");
};

// One streamed pass over the generator packs.  For every trajectory q of the tile:
//   o[q][0]      = row sum of H(eps) x_q,   H(eps) = H0 + sum_l e_l H_l
//   o[q][1 + l]  = row sum of H_l x_q       (only when NOUT == 1 + LMAX: phase A of the update sweep)
template <int JS, int JQ, int RT, int LMAX, int KT, int NOUT>
__device__ __forceinline__ void kr_stream_rows(const __amdgpu_buffer_rsrc_t rsrc, const int tt, const int L,
                                               const double (&e)[LMAX], const double2 *const (&xp)[KT],
                                               double (&outr)[KT][NOUT], double (&outi)[KT][NOUT]) {
    double ar[KT][NOUT], br[KT][NOUT], ai[KT][NOUT], bi[KT][NOUT];
#pragma unroll
    for (int q = 0; q < KT; ++q)
#pragma unroll
        for (int o = 0; o < NOUT; ++o) { ar[q][o] = 0; br[q][o] = 0; ai[q][o] = 0; bi[q][o] = 0; }
    constexpr int UNR = 4;
    // (not unrolled: hipcc otherwise issues the loads of ALL chunks of the pass up front -- 96 x 16 bytes per thread at NP = 128 --
    //  and spills 1-2.5 KB per thread: 600 us per step at K = 256 against 21 us with one chunk in flight; round 3)
#pragma unroll 1
    for (int j0 = 0; j0 < JS; j0 += UNR) {
        double2 h[LMAX + 1][UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
#pragma unroll
            for (int m = 0; m <= LMAX; ++m) {
                h[m][u] = make_double2(0.0, 0.0);
                if (m <= L) {
                    // buffer load: resource (SGPRs) + one per-lane byte offset + a scalar offset per element.  With global
                    // addressing hipcc hoists the 64-bit address of every load of the unrolled pass out of the step loop
                    // (NP = 128: 96 addresses, 1.8-2.5 KB of scratch per thread: 600 us per step at K = 256; round 3)
                    const unsigned soff = (unsigned)((((unsigned)m * JS + (unsigned)(j0 + u)) * RT) * 16u);
                    const kr_su4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, tt * 16, soff, 0);
                    double2 d;
                    __builtin_memcpy(&d, &raw, 16);
                    h[m][u] = d;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            double cr = h[0][u].x, ci = h[0][u].y;
#pragma unroll
            for (int l = 0; l < LMAX; ++l) { cr = fma(e[l], h[1 + l][u].x, cr); ci = fma(e[l], h[1 + l][u].y, ci); }
#pragma unroll
            for (int q = 0; q < KT; ++q) {
                const double2 x = xp[q][j0 + u];
                ar[q][0] = fma(cr, x.x, ar[q][0]); br[q][0] = fma(ci, x.y, br[q][0]);
                ai[q][0] = fma(cr, x.y, ai[q][0]); bi[q][0] = fma(ci, x.x, bi[q][0]);
                if (NOUT > 1) {
#pragma unroll
                    for (int l = 0; l < LMAX; ++l) {
                        if (l < L) {
                            const double2 hl = h[1 + l][u];
                            ar[q][1 + l] = fma(hl.x, x.x, ar[q][1 + l]); br[q][1 + l] = fma(hl.y, x.y, br[q][1 + l]);
                            ai[q][1 + l] = fma(hl.x, x.y, ai[q][1 + l]); bi[q][1 + l] = fma(hl.y, x.x, bi[q][1 + l]);
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < KT; ++q)
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            outr[q][o] = kr_row_sum<JQ>(ar[q][o] - br[q][o]);
            outi[q][o] = kr_row_sum<JQ>(ai[q][o] + bi[q][o]);
        }
}

// ---- known-pulse sweep (forward, or backward through the adjoint pack) --------------------------
template <int NP, int LMAX, int KT>
__global__ __launch_bounds__((KrGeomS<NP, LMAX, KT>::THREADS)) void kr_stream_known(const KrSweepArgs a) {
    using G = KrGeomS<NP, LMAX, KT>;
    constexpr int JS = G::JS, JQ = G::JQ, RT = G::RT, NPP = G::NPP;
    __shared__ __attribute__((aligned(16))) double2 sx[2][KT][NPP];
    __shared__ __attribute__((aligned(16))) double2 sy[2][KT][NPP];
    const int tid = threadIdx.x, wg = blockIdx.x, lane = tid & 63;
    const int tt = tid, i = tt / JQ, jq = tt % JQ;
    const bool writer = (jq == 0), row_ok = (i < a.N);
    const double my_rinv = kr_rinv[lane <= KR_MAX_ORDER ? lane : 0];
    const double my_thr = kr_load_my_thr(a);
    const KrGen<LMAX> gx = kr_gen_ctx<LMAX>(a, a.wg_gen ? a.wg_gen[blockIdx.x] : 0);
    const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)gx.Hpack, 0, (int)(a.pack_stride * 16), 0x00020000);
    const int pi_ = i + i / JS, xoff = jq * (JS + 1);
    const int N = a.N, NT = a.NT, L = a.L;
    const size_t slab = (size_t)a.Kp * N;
    // economized step polynomial (kr_device.h: KrEcon; Hermitian or nearly Hermitian generators): lane j holds entry j of the step's row
    const double *econ_rows = a.econ ? a.econ + ((size_t)gx.gen * NT) * KR_ECON_W + (lane & (KR_ECON_W - 1)) : nullptr;

    double pr[KT], pim[KT];
    size_t kbase[KT];
#pragma unroll
    for (int q = 0; q < KT; ++q) {
        kbase[q] = (size_t)(wg * KT + q) * N;
        double2 v = make_double2(0.0, 0.0);
        if (row_ok) v = a.init[kbase[q] + i];
        pr[q] = v.x; pim[q] = v.y;
        if (writer) sx[0][q][pi_] = v;
        if (a.store && writer && row_ok) a.store[(size_t)((a.dir > 0) ? 0 : NT) * slab + kbase[q] + i] = v;
    }
    __syncthreads();
    int cur = 0;
    for (int step = 0; step < NT; ++step) {
        const int n = (a.dir > 0) ? step : (NT - 1 - step);
        const double sdt = (a.dir > 0) ? a.dt[n] : -a.dt[n];
        double e[LMAX];
        double theta = gx.nrm[0];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            e[l] = (l < L) ? a.eps[(size_t)l * NT + n] : 0.0;
            theta += fabs(e[l]) * gx.nrm[1 + l];
        }
        theta *= fabs(sdt);
        int s, m;
        kr_plan_fast(a, my_thr, theta, s, m);
        const double ec = econ_rows ? econ_rows[(size_t)n * KR_ECON_W] : 0.0;
        int se = s, me = m;
        KrEcon eco = kr_econ_select(ec, theta, se, me);
        if (eco.on && (s > 1 || me < m)) { s = se; m = me; }          // (a row whose degree saves nothing is not used)
        else { eco.on = false; eco.a0 = 1.0; eco.a1 = 1.0; }
        const double my_g = eco.on ? ec : my_rinv;   // Horner ratios of this step, lane j holds g_j
        const double fs = (s == 1) ? sdt : sdt / (double)s;
        for (int sub = 0; sub < s; ++sub) {
            int w = 0;
            for (int j = m; j >= 1; --j) {
                // (last round: a_0 psi + a_1 z y_1 -- entry 1 of the row is a_1 itself, not a ratio; Taylor: 1, 1)
                const double f = (j == 1) ? fs * eco.a1 : fs * kr_rinv_lane(my_g, j);
                const double c0 = (j == 1) ? eco.a0 : 1.0;
                const double2 *xp[KT];
#pragma unroll
                for (int q = 0; q < KT; ++q) xp[q] = (j == m) ? &sx[cur][q][xoff] : &sy[w ^ 1][q][xoff];
                double sr[KT][1], si[KT][1];
                kr_stream_rows<JS, JQ, RT, LMAX, KT, 1>(hrsrc, tt, L, e, xp, sr, si);
#pragma unroll
                for (int q = 0; q < KT; ++q) {
                    const double yr = fma(f, si[q][0], c0 * pr[q]);
                    const double yi = fma(-f, sr[q][0], c0 * pim[q]);
                    if (j > 1) {
                        if (writer) sy[w][q][pi_] = make_double2(yr, yi);
                    } else {
                        pr[q] = yr; pim[q] = yi;
                        if (writer) sx[cur ^ 1][q][pi_] = make_double2(yr, yi);
                    }
                }
                __syncthreads();
                w ^= 1;
            }
            cur ^= 1;
        }
        if (a.store && writer && row_ok) {
            const int slot = (a.dir > 0) ? (n + 1) : n;
#pragma unroll
            for (int q = 0; q < KT; ++q) a.store[(size_t)slot * slab + kbase[q] + i] = make_double2(pr[q], pim[q]);
        }
    }
    if (a.fin && writer && row_ok) {
#pragma unroll
        for (int q = 0; q < KT; ++q) a.fin[kbase[q] + i] = make_double2(pr[q], pim[q]);
    }
}

// ---- update sweep ---------------------------------------------------------------------------------
template <int NP, int LMAX, int KT>
__global__ __launch_bounds__((KrGeomS<NP, LMAX, KT>::THREADS)) void kr_stream_update(const KrSweepArgs a) {
    using G = KrGeomS<NP, LMAX, KT>;
    constexpr int JS = G::JS, JQ = G::JQ, RT = G::RT, NPP = G::NPP, ROWS = G::ROWS;
    __shared__ __attribute__((aligned(16))) double2 sx[2][KT][NPP];
    __shared__ __attribute__((aligned(16))) double2 sy[2][KT][NPP];
    __shared__ double spart[LMAX * ROWS];
    __shared__ double stot[LMAX + 1];
    const int tid = threadIdx.x, wg = blockIdx.x, lane = tid & 63;
    const int tt = tid, i = tt / JQ, jq = tt % JQ;
    const bool writer = (jq == 0), row_ok = (i < a.N);
    const double my_rinv = kr_rinv[lane <= KR_MAX_ORDER ? lane : 0];
    const double my_thr = kr_load_my_thr(a);
    const KrGen<LMAX> gx = kr_gen_ctx<LMAX>(a, a.wg_gen ? a.wg_gen[blockIdx.x] : 0);
    const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)gx.Hpack, 0, (int)(a.pack_stride * 16), 0x00020000);
    const int pi_ = i + i / JS, xoff = jq * (JS + 1);
    const int N = a.N, NT = a.NT, L = a.L;
    const size_t slab = (size_t)a.Kp * N;
    // economized step polynomial (kr_device.h: KrEcon; Hermitian or nearly Hermitian generators): lane j holds entry j of the step's row
    const double *econ_rows = a.econ ? a.econ + ((size_t)gx.gen * NT) * KR_ECON_W + (lane & (KR_ECON_W - 1)) : nullptr;

    double pr[KT], pim[KT], cr[KT], ci[KT];
    size_t kbase[KT];
#pragma unroll
    for (int q = 0; q < KT; ++q) {
        kbase[q] = (size_t)(wg * KT + q) * N;
        double2 v = make_double2(0.0, 0.0), c = make_double2(0.0, 0.0);
        if (row_ok) { v = a.init[kbase[q] + i]; c = a.X[kbase[q] + i]; }
        pr[q] = v.x; pim[q] = v.y; cr[q] = c.x; ci[q] = c.y;
        if (writer) sx[0][q][pi_] = v;
        if (a.store && writer && row_ok) a.store[kbase[q] + i] = v;
    }
    double gint[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) gint[l] = 0.0;
    __syncthreads();
    int cur = 0;
    double *const dulog = a.dulog + (size_t)wg * NT * LMAX;   // (kr_sweep_update, kr_device.h)
    for (int n = 0; n < NT; ++n) {
        const double dt = a.dt[n];
        const double ec = econ_rows ? econ_rows[(size_t)n * KR_ECON_W] : 0.0;
        double eg[LMAX], al[LMAX];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            eg[l] = (l < L) ? a.eps[(size_t)l * NT + n] : 0.0;
            al[l] = (l < L) ? a.alpha[(size_t)l * NT + n] : 0.0;
        }
        // phase A: b = H(eps_guess) psi, a_l = H_l psi in ONE pass over the generator
        const double2 *xp[KT];
#pragma unroll
        for (int q = 0; q < KT; ++q) xp[q] = &sx[cur][q][xoff];
        double orr[KT][1 + LMAX], oi[KT][1 + LMAX];
        kr_stream_rows<JS, JQ, RT, LMAX, KT, 1 + LMAX>(hrsrc, tt, L, eg, xp, orr, oi);
        double part[LMAX];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            part[l] = 0.0;
            if (l < L && ((gx.present >> l) & 1u)) {
#pragma unroll
                for (int q = 0; q < KT; ++q) part[l] += cr[q] * oi[q][1 + l] - ci[q] * orr[q][1 + l];
            }
        }
        if (n + 1 < NT && row_ok) {  // chi(t_{n+1}); these lanes are not the polling wave's critical loads
#pragma unroll
            for (int q = 0; q < KT; ++q) {
                const double2 c = a.X[(size_t)(n + 1) * slab + kbase[q] + i];
                cr[q] = c.x; ci[q] = c.y;
            }
        }
        double du[LMAX];
        if (!kr_exchange<LMAX, ROWS>(a, n, wg, writer, i, part, du, spart, stot)) return;
        if (tid == 128) {
#pragma unroll
            for (int l = 0; l < LMAX; ++l) dulog[(size_t)n * LMAX + l] = (l == 0 && a.rs_fault == 2 && wg == 1 && n == 3) ? -du[l] : du[l];
        }

        double e[LMAX], de[LMAX];
        double theta = gx.nrm[0];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            e[l] = 0.0; de[l] = 0.0;
            if (l < L) {
                const double deps = __dmul_rn(al[l], du[l]);
                e[l] = __dadd_rn(eg[l], deps);
                de[l] = e[l] - eg[l];
                const double ad = fabs(du[l]);
                gint[l] = __dadd_rn(gint[l], __dmul_rn(__dmul_rn(al[l], __dmul_rn(ad, ad)), dt));
                theta += fabs(e[l]) * gx.nrm[1 + l];
            }
        }
        theta *= fabs(dt);
        int s, m;
        kr_plan_fast(a, my_thr, theta, s, m);
        int se = s, me = m;
        KrEcon eco = kr_econ_select(ec, theta, se, me);
        if (eco.on && (s > 1 || me < m)) { s = se; m = me; }          // (a row whose degree saves nothing is not used)
        else { eco.on = false; eco.a0 = 1.0; eco.a1 = 1.0; }
        const double my_g = eco.on ? ec : my_rinv;   // Horner ratios of this step, lane j holds g_j
        const double fs = (s == 1) ? dt : dt / (double)s;
        {
            double vr[KT], vi[KT], yr[KT], yi[KT];
#pragma unroll
            for (int q = 0; q < KT; ++q) {
                double wr = orr[q][0], wi = oi[q][0];
#pragma unroll
                for (int l = 0; l < LMAX; ++l) { wr = fma(de[l], orr[q][1 + l], wr); wi = fma(de[l], oi[q][1 + l], wi); }
                vr[q] = fs * wi; vi[q] = -fs * wr;
                yr[q] = vr[q]; yi[q] = vi[q];
                if (m > 1 && writer) sy[0][q][pi_] = make_double2(vr[q], vi[q]);
            }
            if (m > 1) __syncthreads();
            int w = 1;
            for (int j = m; j >= 2; --j) {
                const double f = fs * kr_rinv_lane(my_g, j);
#pragma unroll
                for (int q = 0; q < KT; ++q) xp[q] = &sy[w ^ 1][q][xoff];
                double sr[KT][1], si[KT][1];
                kr_stream_rows<JS, JQ, RT, LMAX, KT, 1>(hrsrc, tt, L, e, xp, sr, si);
#pragma unroll
                for (int q = 0; q < KT; ++q) {
                    yr[q] = fma(f, si[q][0], vr[q]);
                    yi[q] = fma(-f, sr[q][0], vi[q]);
                    if (j > 2 && writer) sy[w][q][pi_] = make_double2(yr[q], yi[q]);
                }
                if (j > 2) __syncthreads();
                w ^= 1;
            }
#pragma unroll
            for (int q = 0; q < KT; ++q) {
                pr[q] = fma(eco.a1, yr[q], eco.a0 * pr[q]); pim[q] = fma(eco.a1, yi[q], eco.a0 * pim[q]);   // psi' = a_0 psi + a_1 y_1  (Taylor: 1, 1)
                if (writer) sx[cur ^ 1][q][pi_] = make_double2(pr[q], pim[q]);
            }
            __syncthreads();
            cur ^= 1;
        }
        for (int sub = 1; sub < s; ++sub) {
            int w = 0;
            for (int j = m; j >= 1; --j) {
                const double f = fs * kr_rinv_lane(my_rinv, j);
#pragma unroll
                for (int q = 0; q < KT; ++q) xp[q] = (j == m) ? &sx[cur][q][xoff] : &sy[w ^ 1][q][xoff];
                double sr[KT][1], si[KT][1];
                kr_stream_rows<JS, JQ, RT, LMAX, KT, 1>(hrsrc, tt, L, e, xp, sr, si);
#pragma unroll
                for (int q = 0; q < KT; ++q) {
                    const double y_r = fma(f, si[q][0], pr[q]);
                    const double y_i = fma(-f, sr[q][0], pim[q]);
                    if (j > 1) {
                        if (writer) sy[w][q][pi_] = make_double2(y_r, y_i);
                    } else {
                        pr[q] = y_r; pim[q] = y_i;
                        if (writer) sx[cur ^ 1][q][pi_] = make_double2(y_r, y_i);
                    }
                }
                __syncthreads();
                w ^= 1;
            }
            cur ^= 1;
        }
        if (a.store && writer && row_ok) {
#pragma unroll
            for (int q = 0; q < KT; ++q) a.store[(size_t)(n + 1) * slab + kbase[q] + i] = make_double2(pr[q], pim[q]);
        }
    }
    if (a.fin && writer && row_ok) {
#pragma unroll
        for (int q = 0; q < KT; ++q) a.fin[kbase[q] + i] = make_double2(pr[q], pim[q]);
    }
    if (wg == 0 && tid == 0) {
#pragma unroll
        for (int l = 0; l < LMAX; ++l) if (l < L) a.g_a_int[l] = gint[l];
    }
}

template <int NP>
static hipError_t kr_launch_stream_np(int lmax, int kt, int kind, int grid, hipStream_t s, const KrSweepArgs *a) {
#define KR_CASE_S(LM, KTv)                                                                          \
    if (lmax == LM && kt == KTv) {                                                                  \
        if (kind == 0) kr_launch_kernel((kr_stream_known<NP, LM, KTv>), grid, KrGeomS<NP, LM, KTv>::THREADS, 0, s, *a); \
        else kr_launch_kernel((kr_stream_update<NP, LM, KTv>), grid, KrGeomS<NP, LM, KTv>::THREADS, 0, s, *a);          \
        return hipGetLastError();                                                                   \
    }
    KR_CASE_S(2, 1) KR_CASE_S(2, 2) KR_CASE_S(2, 4)
    KR_CASE_S(4, 1) KR_CASE_S(4, 2) KR_CASE_S(4, 4)
#undef KR_CASE_S
    return hipErrorInvalidValue;
}
