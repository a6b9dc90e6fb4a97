// Update sweep on the matrix cores for NP = 64: v_mfma_f64_4x4x4_4b_f64 (four 4x4x4 blocks per instruction).
//
// Why this instruction: it issues every 16 cycles per SIMD from one wave with >= 2 independent accumulators
// (tools/ubench/mfma4x4.hip: 16.1-17.0; 22 on one dependent chain) = 512 flop / 16 cycles = the fp64 peak, the same rate
// as v_mfma_f64_16x16x4_f64 -- but its tile is 16 rows x 4 columns, so ONE workgroup needs only 2 trajectories to fill
// it.  The register-resident VALU sweep (kr_device.h) reaches that rate only on paper: next to the 32 FMAs of a row
// slice it spends 8 LDS reads, 12 DPP moves and 6 adds per trajectory and round, and with two or four trajectories per
// team (K = 512 / 1024 per GPU at N = 64) it spills.
//
// Packed complex layout.  A 4-column tile carries 2 trajectories: column j = 2 t + c holds component c (0 = re, 1 = im) of
// trajectory t.  For a complex matrix H = Hr + i Hi and a vector w,   packed(H w) = Hr * packed(w) + Hi * packed(i w),
// so one complex product is two real MFMA chains into the same kind of accumulator and the result comes out packed again.
// Every vector lives in LDS twice, as packed(w) ("Y") and packed(-i w) ("Z"); the products are taken of z = -i w, i.e.
//      D = Hr * Z + Hi * Y = packed(-i H w),
// which is exactly what the Taylor recurrence needs:   y_{j-1} = psi + (fs / j) (-i H y_j)   is then LANE-LOCAL (each lane
// holds one real component of one element: no cross-lane traffic, no complex arithmetic), and so are the update integrand
// Im<chi|H_l|psi> = sum chi_c (-i H_l psi)_c and the first-application shortcut.
//
// Mapping: 256 threads = 4 waves, one per SIMD (up to 512 VGPRs per lane: the A operands of H0, H_1..H_L and H(eps), 64
// VGPRs each, stay in registers for the whole sweep, also for L = 3, 4 -- the VALU variants for LMAX = 4 spill).  Wave w
// owns rows 16 w .. 16 w + 15 (its four 4x4 blocks), lane l = (f0 = l & 3, blk = (l >> 2) & 3, f2 = l >> 4):
//      A operand, k-block kb : H[16 w + 4 blk + f0][4 kb + f2]          (lane map measured: tools/ubench/mfma4x4.hip)
//      B operand, k-block kb : X[4 kb + f2][column f0]                  (the same for all four blocks: LDS broadcast)
//      D                     : Y[16 w + 4 blk + f2][column f0]
// A round = 16 k-blocks x 2 chains = 32 MFMAs per tile and wave (512 cycles), one 8-byte LDS read per MFMA, two 8-byte LDS
// writes per tile, one s_barrier.  TILES tiles per workgroup (KT = 2 TILES trajectories).
//
// Reference semantics: kr_sweep_update (kr_device.h), i.e. /root/reference/src/optimize.jl:321-370.
#pragma once
#include "kr_device.h"

__device__ __forceinline__ double kr_mfma4(const double a, const double b, const double c) {
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

#define KR_MF_TW 256  // doubles per packed tile array: 64 rows x 4 columns

// d[t] = packed(-i H w)[this lane] for the TILES tiles whose packed(-i w) / packed(w) arrays start at Z / Y
template <int TILES>
__device__ __forceinline__ void kr_mf_apply(const double (&Ar)[16], const double (&Ai)[16], const double *__restrict__ Z,
                                            const double *__restrict__ Y, const int boff, double (&d)[TILES]) {
    double pr[TILES], pi[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t) { pr[t] = 0.0; pi[t] = 0.0; }
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            const double bz = Z[t * KR_MF_TW + kb * 16 + boff], by = Y[t * KR_MF_TW + kb * 16 + boff];
            pr[t] = kr_mfma4(Ar[kb], bz, pr[t]);
            pi[t] = kr_mfma4(Ai[kb], by, pi[t]);
        }
    }
#pragma unroll
    for (int t = 0; t < TILES; ++t) d[t] = pr[t] + pi[t];
}

template <int LMAX, int TILES, int SCHED, int CAL, int HERM>
__global__ __launch_bounds__(256, 1) void kr_mf_update(const KrSweepArgs a) {
    constexpr int KT = 2 * TILES, TW = KR_MF_TW, NW = 4;
    constexpr int CH = KT;  // 16-B elements per lane when the memory wave moves a whole tile (KT * 64 / 64)
    __shared__ __attribute__((aligned(16))) double sxY[2][TILES][TW];  // psi(t_n), packed, by parity `cur`
    __shared__ __attribute__((aligned(16))) double sxZ[2][TILES][TW];  // -i psi(t_n)
    __shared__ __attribute__((aligned(16))) double syY[2][TILES][TW];  // Horner vectors, ping-pong
    __shared__ __attribute__((aligned(16))) double syZ[2][TILES][TW];
    __shared__ __attribute__((aligned(16))) double scY[2][TILES][TW];  // chi(t_k) in slot k & 1, packed
    __shared__ __attribute__((aligned(16))) double scZ[HERM ? 2 : 1][HERM ? TILES : 1][HERM ? TW : 4];  // -i chi (matvec operand when HERM)
    __shared__ double spart[LMAX * NW];
    __shared__ double stot[LMAX + 1];

    const int tid = threadIdx.x, wg = blockIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f0 = lane & 3, blk = (lane >> 2) & 3, f2 = lane >> 4;
    const int arow = 16 * wave + 4 * blk + f0;   // row of this lane's A elements
    const int drow = 16 * wave + 4 * blk + f2;   // row of this lane's D element
    const int dt = f0 >> 1, dc = f0 & 1;         // trajectory within the tile / component of this lane's D element
    const int boff = f2 * 4 + f0;                // B operand: X[4 kb + f2][f0]
    const int doff = drow * 4 + f0;              // where this lane's component lives in a packed array
    const int zoff = drow * 4 + (f0 ^ 1);        // -i w: (w_i, -w_r) -> this lane's component goes to the OTHER column
    const double zs = dc ? 1.0 : -1.0;           //        with this sign
    const bool memwave = (wave == NW - 1);
    const double my_rinv = kr_rinv[lane <= KR_MAX_ORDER ? lane : 0];
    const double my_thr = kr_load_my_thr(a);
    const KrGen<LMAX> gx = kr_gen_ctx<LMAX>(a, a.wg_gen ? a.wg_gen[blockIdx.x] : 0);
    const int N = a.N, NT = a.NT, L = a.L;
    const size_t slab = (size_t)a.Kp * N;
    const size_t tile_base = (size_t)wg * KT * N;  // the tile's trajectories are contiguous in every [Kp][N] slab
    const int tile_elems = KT * N;
    const bool row_ok = drow < N;

    // A operands of H0, H_1..H_L (dense row-major [m][64][64], zero padded): 16 k-blocks x (re, im) per matrix
    double Hr[LMAX + 1][16], Hi[LMAX + 1][16];
    {
        const double2 *Hd = a.Hdense + (size_t)gx.gen * a.dense_stride;
#pragma unroll
        for (int m = 0; m <= LMAX; ++m) {
#pragma unroll
            for (int kb = 0; kb < 16; ++kb) {
                double2 h = make_double2(0.0, 0.0);
                if (m <= L) h = Hd[((size_t)m * 64 + arow) * 64 + 4 * kb + f2];
                Hr[m][kb] = h.x;
                Hi[m][kb] = h.y;
            }
        }
    }

    // memory wave: element e = lane + 64 c of the tile <-> (trajectory tq, row i); packed position of its double2
    int m_pos[CH];
    bool m_ok[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int e = lane + 64 * c;
        m_ok[c] = e < tile_elems;
        const int tq = m_ok[c] ? e / N : 0, i = m_ok[c] ? e % N : 0;
        m_pos[c] = (tq >> 1) * TW + i * 4 + 2 * (tq & 1);
    }
    double2 cpf[CH];  // chi prefetch registers (memory wave)

    // initial state into registers (D layout) and LDS (packed psi and -i psi); rows >= N stay zero
    double psi[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        double v = 0.0;
        if (row_ok) v = ((const double *)a.init)[(tile_base + (size_t)(2 * t + dt) * N + drow) * 2 + dc];
        psi[t] = v;
        sxY[0][t][doff] = v;
        sxZ[0][t][zoff] = zs * v;
        double c0 = 0.0, c1 = 0.0;
        if (row_ok) {
            c0 = ((const double *)a.X)[(tile_base + (size_t)(2 * t + dt) * N + drow) * 2 + dc];              // chi(t_0)
            if (HERM && NT > 1) c1 = ((const double *)a.X)[(slab + tile_base + (size_t)(2 * t + dt) * N + drow) * 2 + dc];  // chi(t_1)
        }
        scY[0][t][doff] = c0;
        scY[1][t][doff] = c1;
        if (HERM) { scZ[0][t][zoff] = zs * c0; scZ[1][t][zoff] = zs * c1; }
    }
    if (memwave && a.store) {
#pragma unroll
        for (int c = 0; c < CH; ++c)
            if (m_ok[c]) a.store[tile_base + lane + 64 * c] = a.init[tile_base + lane + 64 * c];
    }
    double gint[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) gint[l] = 0.0;
    __syncthreads();
    // HERM: cl[l][t] = packed(-i H_l chi(t_n))[this lane], carried from the window of step n-1; n = 0 here
    double cl[LMAX][TILES];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
#pragma unroll
        for (int t = 0; t < TILES; ++t) cl[l][t] = 0.0;
        if (HERM && l < L) kr_mf_apply<TILES>(Hr[1 + l], Hi[1 + l], &scZ[0][0][0], &scY[0][0][0], boff, cl[l]);
    }

    const bool stamp = (a.debug != nullptr) && (wg == 0);
    unsigned long long tA = 0, tX = 0, tB = 0, tS = 0, t_prev = 0, rt0 = 0;
    if (stamp) { t_prev = __builtin_amdgcn_s_memtime(); rt0 = kr_realtime(); }
    int cur = 0;
    double eg_nx[LMAX], al_nx[LMAX], dt_nx = a.dt[0];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
        eg_nx[l] = (l < L) ? a.eps[(size_t)l * NT] : 0.0;
        al_nx[l] = (l < L) ? a.alpha[(size_t)l * NT] : 0.0;
    }
    for (int n = 0; n < NT; ++n) {
        const double dtn = dt_nx;
        double eg[LMAX], al[LMAX];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) { eg[l] = eg_nx[l]; al[l] = al_nx[l]; }
        if (n + 1 < NT) {
            dt_nx = a.dt[n + 1];
#pragma unroll
            for (int l = 0; l < LMAX; ++l)
                if (l < L) { eg_nx[l] = a.eps[(size_t)l * NT + n + 1]; al_nx[l] = a.alpha[(size_t)l * NT + n + 1]; }
        }
        // D[m][t] = packed(-i H_m psi)[this lane], m = 0 (drift), 1..L (controls)
        double D[LMAX + 1][TILES];
        double part[LMAX];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) part[l] = 0.0;
        auto products = [&]() {
#pragma unroll
            for (int m = 0; m <= LMAX; ++m) {
#pragma unroll
                for (int t = 0; t < TILES; ++t) D[m][t] = 0.0;
                if (m <= L) kr_mf_apply<TILES>(Hr[m], Hi[m], &sxZ[cur][0][0], &sxY[cur][0][0], boff, D[m]);
            }
        };
        if (!HERM) {
            // ---- phase A: Im<chi|H_l|psi> = Im(conj(chi) a), a = H_l psi = i D  ->  chi_r D_r + chi_i D_i, lane by lane ----
            products();
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                const double chc = scY[n & 1][t][doff];
#pragma unroll
                for (int l = 0; l < LMAX; ++l)
                    if (l < L && ((gx.present >> l) & 1u)) part[l] = fma(chc, D[1 + l][t], part[l]);
            }
        } else {
            // ---- Im<H_l chi|psi> = Im(conj(c) psi), c = H_l chi = i cl  ->  -(cl_r psi_r + cl_i psi_i) ----
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
#pragma unroll
                for (int l = 0; l < LMAX; ++l)
                    if (l < L && ((gx.present >> l) & 1u)) part[l] = fma(-cl[l][t], psi[t], part[l]);
            }
        }
#pragma unroll
        for (int l = 0; l < LMAX; ++l) part[l] = kr_wave_sum(part[l]);   // fixed order: bitwise reproducible
        auto window = [&]() {
            if (HERM) {
                products();
#pragma unroll
                for (int l = 0; l < LMAX; ++l)
                    if (l < L && n + 1 < NT)
                        kr_mf_apply<TILES>(Hr[1 + l], Hi[1 + l], &scZ[(n + 1) & 1][0][0], &scY[(n + 1) & 1][0][0], boff, cl[l]);
            }
        };
        if (stamp) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tA += t - t_prev; t_prev = t; }

        // ---- global sum over all trajectories of all workgroups (and GPUs): one value per wave goes in ----
        double du[LMAX];
        if (!kr_exchange<LMAX, NW, SCHED>(a, n, wg, lane == 0, wave, part, du, spart, stot, window)) return;
        if (stamp) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tX += t - t_prev; t_prev = t; }

        // ---- pulse update (optimize.jl:351-358), identical on every workgroup ----
        double e[LMAX];
        double theta = gx.nrm[0];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            e[l] = 0.0;
            if (l < L) {
                const double alpha = al[l];
                e[l] = __dadd_rn(eg[l], __dmul_rn(alpha, du[l]));
                const double ad = fabs(du[l]);
                gint[l] = __dadd_rn(gint[l], __dmul_rn(__dmul_rn(alpha, __dmul_rn(ad, ad)), dtn));
                theta += fabs(e[l]) * gx.nrm[1 + l];
                if (wg == 0 && memwave && lane == 0) a.eps_new[(size_t)l * NT + n] = e[l];
            }
        }
        theta *= fabs(dtn);
        int s, m;
        kr_plan_fast(a, my_thr, theta, s, m);
        const double fs = (s == 1) ? dtn : dtn / (double)s;

        // ---- psi <- exp(-i H(eps_new) dt) psi.  -i H(eps_new) psi = D_0 + sum e_l D_l is known already, so the first
        //      sub-step costs m - 1 products:  psi' = psi + y_1,  y_m = v,  y_{j-1} = v + (fs / j)(-i H y_j),  v = fs (-i H psi)
        double v[TILES], y[TILES];
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            double w = D[0][t];
#pragma unroll
            for (int l = 0; l < LMAX; ++l) w = fma(e[l], D[1 + l][t], w);
            v[t] = fs * w;
            y[t] = v[t];
            if (m > 1) { syY[0][t][doff] = v[t]; syZ[0][t][zoff] = zs * v[t]; }
        }
        // A operands of H(eps_new)
        double Er[16], Ei[16];
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            double r = Hr[0][kb], im = Hi[0][kb];
#pragma unroll
            for (int l = 0; l < LMAX; ++l) { r = fma(e[l], Hr[1 + l][kb], r); im = fma(e[l], Hi[1 + l][kb], im); }
            Er[kb] = r;
            Ei[kb] = im;
        }
        if (m > 1) __syncthreads();
        // memory wave: start fetching chi(t_{n+1}) (HERM: chi(t_{n+2})); it lands while the Taylor rounds run
        if (memwave && n + 1 + HERM < NT) {
#pragma unroll
            for (int c = 0; c < CH; ++c)
                cpf[c] = a.X[(size_t)(n + 1 + HERM) * slab + tile_base + min(lane + 64 * c, tile_elems - 1)];
        }
        int wb = 1;
        for (int j = m; j >= 2; --j) {
            const double f = fs * kr_rinv_lane(my_rinv, j);
            double d[TILES];
            kr_mf_apply<TILES>(Er, Ei, &syZ[wb ^ 1][0][0], &syY[wb ^ 1][0][0], boff, d);
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                y[t] = fma(f, d[t], v[t]);
                if (j > 2) { syY[wb][t][doff] = y[t]; syZ[wb][t][zoff] = zs * y[t]; }
            }
            if (j > 2) __syncthreads();
            wb ^= 1;
        }
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            psi[t] += y[t];
            sxY[cur ^ 1][t][doff] = psi[t];
            sxZ[cur ^ 1][t][zoff] = zs * psi[t];
        }
        auto hand_over_chi = [&]() {   // chi(t_{n+1}) (HERM: chi(t_{n+2})) from the prefetch registers into LDS
            if (memwave && n + 1 + HERM < NT) {
                const int par = (n + 1 + HERM) & 1;
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    if (m_ok[c]) {
                        *(double2 *)(&scY[par][0][0] + m_pos[c]) = cpf[c];
                        if (HERM) *(double2 *)(&scZ[par][0][0] + m_pos[c]) = make_double2(cpf[c].y, -cpf[c].x);
                    }
                }
            }
        };
        if (s == 1) hand_over_chi();
        __syncthreads();
        cur ^= 1;
        for (int sub = 1; sub < s; ++sub) {  // further sub-steps (only when ||H dt|| > KR_THETA_MAX): plain Horner from psi
            int w2 = 0;
            for (int j = m; j >= 1; --j) {
                const double f = fs * kr_rinv_lane(my_rinv, j);
                double d[TILES];
                if (j == m) kr_mf_apply<TILES>(Er, Ei, &sxZ[cur][0][0], &sxY[cur][0][0], boff, d);
                else kr_mf_apply<TILES>(Er, Ei, &syZ[w2 ^ 1][0][0], &syY[w2 ^ 1][0][0], boff, d);
#pragma unroll
                for (int t = 0; t < TILES; ++t) {
                    const double yn = fma(f, d[t], psi[t]);
                    if (j > 1) { syY[w2][t][doff] = yn; syZ[w2][t][zoff] = zs * yn; }
                    else { psi[t] = yn; sxY[cur ^ 1][t][doff] = yn; sxZ[cur ^ 1][t][zoff] = zs * yn; }
                }
                if (j == 1 && sub == s - 1) hand_over_chi();
                __syncthreads();
                w2 ^= 1;
            }
            cur ^= 1;
        }
        if (stamp) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tB += t - t_prev; t_prev = t; }

        // memory wave: forward storage slot n+1 <- psi(t_{n+1}) (coalesced 16-B stores from LDS)
        if (memwave && a.store) {
#pragma unroll
            for (int c = 0; c < CH; ++c)
                if (m_ok[c]) a.store[(size_t)(n + 1) * slab + tile_base + lane + 64 * c] = *(const double2 *)(&sxY[cur][0][0] + m_pos[c]);
        }
        if (stamp) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tS += t - t_prev; t_prev = t; }
    }
    if (stamp && tid == 0) {
        a.debug[0] = tA; a.debug[1] = tX; a.debug[2] = tB; a.debug[3] = tS;
        a.debug[4] = kr_realtime() - rt0; a.debug[5] = 0; a.debug[6] = 0;
    }
    if (a.fin && row_ok) {
#pragma unroll
        for (int t = 0; t < TILES; ++t)
            ((double *)a.fin)[(tile_base + (size_t)(2 * t + dt) * N + drow) * 2 + dc] = psi[t];
    }
    if (wg == 0 && tid == 0) {
#pragma unroll
        for (int l = 0; l < LMAX; ++l) if (l < L) a.g_a_int[l] = gint[l];
    }
}

// kind = 1 | SCHED << 4 | CAL << 5 | HERM << 6 (as kr_launch_np); tiles = trajectories per workgroup / 2
static hipError_t kr_launch_mfma64(int lmax, int tiles, int kind, int grid, hipStream_t s, const KrSweepArgs *a) {
    if ((kind & 15) != 1) return hipErrorInvalidValue;
#define KR_CASE_M(LM, TL)                                                                                          \
    if (lmax == LM && tiles == TL) {                                                                               \
        switch (kind >> 4) {                                                                                       \
        case 0: hipLaunchKernelGGL((kr_mf_update<LM, TL, 0, 0, 0>), dim3(grid), dim3(256), 0, s, *a); break;      \
        case 1: hipLaunchKernelGGL((kr_mf_update<LM, TL, 1, 0, 0>), dim3(grid), dim3(256), 0, s, *a); break;      \
        case 2: hipLaunchKernelGGL((kr_mf_update<LM, TL, 0, 1, 0>), dim3(grid), dim3(256), 0, s, *a); break;      \
        case 3: hipLaunchKernelGGL((kr_mf_update<LM, TL, 1, 1, 0>), dim3(grid), dim3(256), 0, s, *a); break;      \
        case 4: hipLaunchKernelGGL((kr_mf_update<LM, TL, 0, 0, 1>), dim3(grid), dim3(256), 0, s, *a); break;      \
        case 5: hipLaunchKernelGGL((kr_mf_update<LM, TL, 1, 0, 1>), dim3(grid), dim3(256), 0, s, *a); break;      \
        case 6: hipLaunchKernelGGL((kr_mf_update<LM, TL, 0, 1, 1>), dim3(grid), dim3(256), 0, s, *a); break;      \
        default: hipLaunchKernelGGL((kr_mf_update<LM, TL, 1, 1, 1>), dim3(grid), dim3(256), 0, s, *a); break;     \
        }                                                                                                          \
        return hipGetLastError();                                                                                  \
    }
    KR_CASE_M(2, 1) KR_CASE_M(2, 2) KR_CASE_M(2, 4)
    KR_CASE_M(4, 1) KR_CASE_M(4, 2) KR_CASE_M(4, 4)
#undef KR_CASE_M
    return hipErrorInvalidValue;
}
