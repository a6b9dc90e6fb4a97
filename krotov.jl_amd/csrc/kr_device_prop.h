// Precomputed step propagators for the two KNOWN-pulse sweeps (initial forward sweep, backward sweep).
//
// When the pulses of a sweep are known in advance (optimize.jl:247-265 and :303-317) all N_T step
// propagators are independent of each other: kr_expm_batch builds them in parallel (one workgroup per time
// step; the columns of U_n = exp(-i H(eps_n) dt) are the unit vectors pushed through the same Horner/Taylor
// rounds the sweeps use, so accuracy and sub-stepping rules are identical), and kr_sweep_apply then needs
// ONE matvec round per step instead of m ~ 9.  The update sweep cannot use this: eps_n is only known after
// the global sum of step n (optimize.jl:339-361).
//
// U_n is stored thread-major for the apply kernel's 4*NP-thread teams:
//   Upack[n][jj * 4NP + i*4 + jq] = U_n[i][jq*(NP/4) + jj]            (16 B per element)
#pragma once
// (included at the end of kr_device.h)

// Geometry of the consumer kr_sweep_apply: lanes per row JQA (8 with 512-thread workgroups for NP = 64, else 4).
// U_n is stored thread-major for it:  Upack[n][jj * (JQA*NP) + i*JQA + jq] = U_n[i][jq*(NP/JQA) + jj].
template <int NP> struct KrApplyGeom {
    static constexpr int JQA = (NP >= 64) ? KR_APPLY_JQ64 : 4;
    static constexpr int JS = NP / JQA;
    static constexpr int RT = NP * JQA;
    static constexpr int THREADS = (RT > 256) ? RT : 256;
    static constexpr int SG = THREADS / RT;
};

// ---- batched propagator construction -------------------------------------------------------------
template <int NP, int LMAX, int CB>
__global__ __launch_bounds__((KrGeom<NP, LMAX, 1>::THREADS), (KrGeom<NP, LMAX, 1>::THREADS / 256)) void kr_expm_batch(const KrSweepArgs a) {
    double2 *__restrict__ Upack = a.U;
    using G = KrGeom<NP, LMAX, 1>;
    constexpr int JS = G::JS, JQ = G::JQ, RT = G::RT, SG = G::SG, NPP = G::NPP;
    constexpr int JS_A = KrApplyGeom<NP>::JS, RT_A = KrApplyGeom<NP>::RT, JQ_A = KrApplyGeom<NP>::JQA;  // consumer geometry
    __shared__ __attribute__((aligned(16))) double2 sy[2][SG * CB][NPP];
    __shared__ __attribute__((aligned(16))) double2 sH0[G::H0LDS ? JS * RT : 1];

    const int tid = threadIdx.x, lane = tid & 63;
    const int team = tid / RT, tt = tid % RT, i = tt / JQ, jq = tt % JQ;
    const bool writer = (jq == 0);
    const double my_rinv = kr_rinv[lane <= KR_MAX_ORDER ? lane : 0];
    const double my_thr = kr_load_my_thr(a);
    const KrGen<LMAX> gx = kr_gen_ctx<LMAX>(a, blockIdx.x / a.NT);
    const int pi_ = i + i / JS, xoff = jq * (JS + 1);
    const int NT = a.NT, L = a.L;
    const int n = blockIdx.x % a.NT;
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
    // economized polynomial of the step (KrEcon, kr_device.h) where the sweep has rows for it: Hermitian generators, ||H dt|| <= 1
    const KrEcon eco = kr_econ_select(a.econ ? a.econ[((size_t)gx.gen * NT + n) * KR_ECON_W + (lane & (KR_ECON_W - 1))] : 0.0, theta, s, m);
    const double my_g = eco.on ? a.econ[((size_t)gx.gen * NT + n) * KR_ECON_W + (lane & (KR_ECON_W - 1))] : my_rinv;
    const double fs = (s == 1) ? sdt : sdt / (double)s;

    // H(eps_n) slice of this thread
    double Er[JS], Ei[JS];
#pragma unroll
    for (int jj = 0; jj < JS; ++jj) {
        const double2 h0 = gx.Hpack[((size_t)0 * JS + jj) * RT + tt];
        double r = h0.x, im = h0.y;
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            if (l < L) {
                const double2 hl = gx.Hpack[((size_t)(1 + l) * JS + jj) * RT + tt];
                r = fma(e[l], hl.x, r);
                im = fma(e[l], hl.y, im);
            }
        }
        Er[jj] = r;
        Ei[jj] = im;
    }
    (void)sH0;

    double2 *Un = Upack + ((size_t)gx.gen * NT + n) * NP * NP;
    // columns c = (chunk * SG + team) * CB + b ; every team runs the same number of rounds (uniform barriers)
    for (int c0 = 0; c0 < NP; c0 += SG * CB) {
        const int cbase = c0 + team * CB;
        double pr[CB], pim[CB];
#pragma unroll
        for (int b = 0; b < CB; ++b) {
            pr[b] = (i == cbase + b) ? 1.0 : 0.0;  // unit vector e_c
            pim[b] = 0.0;
            if (writer) sy[0][team * CB + b][pi_] = make_double2(pr[b], 0.0);
        }
        __syncthreads();
        int w = 1;  // sy[0] holds the current state
        for (int sub = 0; sub < s; ++sub) {
            for (int j = m; j >= 1; --j) {
                const double f = fs * kr_rinv_lane(my_g, j);     // (economized: g_j, and a_1 at j = 1)
                const double idc = (j == 1) ? eco.a0 : 1.0;     // coefficient of the vector itself: a_0 in the last round
#pragma unroll
                for (int b = 0; b < CB; ++b) {
                    const int tq = team * CB + b;
                    double sr, si;
                    kr_row_dot<JS, JQ>(Er, Ei, &sy[w ^ 1][tq][xoff], sr, si);
                    const double yr = fma(f, si, idc * pr[b]);
                    const double yi = fma(-f, sr, idc * pim[b]);
                    if (j == 1) { pr[b] = yr; pim[b] = yi; }
                    if (writer) sy[w][tq][pi_] = make_double2(yr, yi);
                }
                __syncthreads();
                w ^= 1;
            }
        }
        if (writer) {
#pragma unroll
            for (int b = 0; b < CB; ++b) {
                const int c = cbase + b;
                if (c < NP) Un[(size_t)(c % JS_A) * RT_A + i * JQ_A + (c / JS_A)] = make_double2(pr[b], pim[b]);
            }
        }
        __syncthreads();
    }
}

// ---- known-pulse sweep with precomputed propagators: one round per step ---------------------------
// TB = 1 (round 5, N <= 32: the known-pulse sweeps in blocks of time): the grid is blocks x tiles; workgroup (tb, tile)
// starts from the boundary state a.tb_bound[tb] (written by the TB = 0 sweep over the block propagators), applies the steps of sweep
// positions [tb S, tb S + len - 1) and stores every state into the slot the sequential sweep fills with it; the block's last position
// is the next block's boundary (the last block adds the final boundary: storage's last slot == the final states, bit for bit).
template <int NP, int KT, int TB = 0>
__global__ __launch_bounds__((KrApplyGeom<NP>::THREADS), (KrApplyGeom<NP>::THREADS / 256)) void kr_sweep_apply(const KrSweepArgs a) {
    const int ntiles_tb = TB ? a.tb_nitems : 1;
    const int wgx = TB ? (int)(blockIdx.x % ntiles_tb) : (int)blockIdx.x, tb = TB ? (int)(blockIdx.x / ntiles_tb) : 0;
    const int NTg = TB ? a.tb_NT : a.NT;   // length of the grid the propagators are indexed by
    const double2 *__restrict__ Upack = a.U + (size_t)(a.wg_gen ? a.wg_gen[wgx / (a.gen_div > 0 ? a.gen_div : 1)] : 0) * NTg * NP * NP;
    using GA = KrApplyGeom<NP>;
    constexpr int JQ = GA::JQA, JS = GA::JS, RT = GA::RT, SG = GA::SG, KTILE = SG * KT, NPP = NP + JQ, NW = GA::THREADS / 64;
    constexpr int CH = (KTILE * NP + 63) / 64;
    __shared__ __attribute__((aligned(16))) double2 sx[2][KTILE][NPP];

    const int tid = threadIdx.x, wg = wgx, lane = tid & 63, wave = tid >> 6;
    const int team = tid / RT, tt = tid % RT, i = tt / JQ, jq = tt % JQ;
    const bool writer = (jq == 0), row_ok = (i < a.N);
    const bool memwave = (wave == NW - 1);
    const int pi_ = i + i / JS, xoff = jq * (JS + 1);
    const int N = a.N;
    const size_t slab = (size_t)a.Kp * N;
    const size_t tile_base = (size_t)wg * KTILE * N;
    const int tile_elems = KTILE * N;
    // sweep positions: step s of this workgroup is position j0 + s of the sweep; NT = the steps it applies
    const int j0 = TB ? tb * a.tb_S : 0;
    const int NT = TB ? (((a.tb_S < NTg - j0) ? a.tb_S : (NTg - j0)) - 1) : a.NT;
    const double2 *__restrict__ init = TB ? a.tb_bound + (size_t)tb * slab : a.init;

    int m_tq[CH], m_i[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int e = lane + 64 * c;
        m_tq[c] = (e < tile_elems) ? e / N : -1;
        m_i[c] = (e < tile_elems) ? e % N : 0;
    }
#pragma unroll
    for (int q = 0; q < KT; ++q) {
        const int tq = q * SG + team;
        double2 v = make_double2(0.0, 0.0);
        if (row_ok) v = init[tile_base + (size_t)tq * N + i];
        if (writer) sx[0][tq][pi_] = v;
    }
    if (memwave && a.store) {
        const int slot0 = (a.dir > 0) ? j0 : NTg - j0;
#pragma unroll
        for (int c = 0; c < CH; ++c)
            if (m_tq[c] >= 0) a.store[(size_t)slot0 * slab + tile_base + lane + 64 * c] = init[tile_base + lane + 64 * c];
        if (TB && j0 + NT + 1 == NTg) {   // the last block: the final boundary into the last slot of the sweep
            const int slotl = (a.dir > 0) ? NTg : 0;
            const double2 *lastb = a.tb_bound + (size_t)(tb + 1) * slab;
#pragma unroll
            for (int c = 0; c < CH; ++c)
                if (m_tq[c] >= 0) a.store[(size_t)slotl * slab + tile_base + lane + 64 * c] = lastb[tile_base + lane + 64 * c];
        }
    }
    // Propagator slices: THREE register sets in rotation, the loop unrolled by three -- step k computes from set k % 3 while the
    // slice of step k + 2 lands in set (k + 2) % 3, whose contents step k - 1 has used up: two steps in flight, and no set is both
    // read and reloaded in the same step.  The loads are asm statements and the waits are written out (vmcnt(2 JS): everything but
    // the two youngest slices has landed; the memory wave's stores only make that stricter): with compiler-visible loads the
    // set in use is COPIED out of the way of the next load, or the loop head waits for vmcnt(0) -- either way the loads of step
    // n + 1 were only issued once those of step n had arrived, and a step took the L2 latency plus the transfer of the slice
    // (C3: 0.85 us per step against ~0.35 us of arithmetic).  Loads past the end re-read the last slice.
    static_assert(2 * JS <= 63, "vmcnt field");
    typedef double kr_v2d __attribute__((ext_vector_type(2)));   // (a vector type: asm register operands cannot be structs)
    kr_v2d u0[JS], u1[JS], u2[JS];
    auto slice_of = [&](const int step) {
        const int jp = j0 + step;
        const int sc = (jp < NTg) ? jp : (NTg - 1);
        const int n = (a.dir > 0) ? sc : (NTg - 1 - sc);
        return Upack + (size_t)n * NP * NP + tt;
    };
#define KR_APPLY_FILL(set_, step_)                                                                                     \
    {                                                                                                                  \
        const double2 *U_ = slice_of(step_);                                                                           \
        _Pragma("unroll") for (int jj = 0; jj < JS; ++jj) {                                                            \
            const double2 *p_ = U_ + (size_t)jj * RT;                                                                  \
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(set_[jj]) : "v"(p_) : "memory");                    \
        }                                                                                                              \
    }
    // (one asm statement holds the wait and names every register of the set: nothing that reads the set can move above it)
#define KR_APPLY_WAIT(set_)                                                                                            \
    {                                                                                                                  \
        if constexpr (JS == 8)                                                                                         \
            asm volatile("s_waitcnt vmcnt(16)" : "+v"(set_[0]), "+v"(set_[1]), "+v"(set_[2]), "+v"(set_[3]), "+v"(set_[4]), "+v"(set_[5]), "+v"(set_[6]), "+v"(set_[7]) :: "memory"); \
        else if constexpr (JS == 4)                                                                                    \
            asm volatile("s_waitcnt vmcnt(8)" : "+v"(set_[0]), "+v"(set_[1]), "+v"(set_[2]), "+v"(set_[3]) :: "memory"); \
        else                                                                                                           \
            asm volatile("s_waitcnt vmcnt(4)" : "+v"(set_[0]), "+v"(set_[1]) :: "memory");                             \
    }
    static_assert(JS == 8 || JS == 4 || JS == 2, "slice sizes of the apply geometry");
    KR_APPLY_FILL(u0, 0)
    KR_APPLY_FILL(u1, 1)
    __syncthreads();

    int cur = 0;
#define KR_APPLY_STEP(step_, use_, fill_)                                                                              \
    {                                                                                                                  \
        const int n = (a.dir > 0) ? (j0 + (step_)) : (NTg - 1 - j0 - (step_));                                         \
        KR_APPLY_FILL(fill_, (step_) + 2)                                                                              \
        KR_APPLY_WAIT(use_)                                                                                            \
        double Ur[JS], Ui[JS];                                                                                         \
        _Pragma("unroll") for (int jj = 0; jj < JS; ++jj) { Ur[jj] = use_[jj].x; Ui[jj] = use_[jj].y; }                \
        _Pragma("unroll") for (int q = 0; q < KT; ++q) {                                                               \
            const int tq = q * SG + team;                                                                              \
            double sr, si;                                                                                             \
            kr_row_dot<JS, JQ>(Ur, Ui, &sx[cur][tq][xoff], sr, si);                                                    \
            if (writer) sx[cur ^ 1][tq][pi_] = make_double2(sr, si);                                                   \
        }                                                                                                              \
        __syncthreads();                                                                                               \
        cur ^= 1;                                                                                                      \
        if (memwave && a.store) {                                                                                      \
            const int slot = (a.dir > 0) ? (n + 1) : n;                                                                \
            _Pragma("unroll") for (int c = 0; c < CH; ++c) {                                                           \
                if (m_tq[c] >= 0) {                                                                                    \
                    const int ii = m_i[c];                                                                             \
                    a.store[(size_t)slot * slab + tile_base + lane + 64 * c] = sx[cur][m_tq[c]][ii + ii / JS];         \
                }                                                                                                      \
            }                                                                                                          \
        }                                                                                                              \
    }
    for (int step = 0; step < NT; step += 3) {
        KR_APPLY_STEP(step, u0, u2)
        if (step + 1 < NT) KR_APPLY_STEP(step + 1, u1, u0)
        if (step + 2 < NT) KR_APPLY_STEP(step + 2, u2, u1)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the last fills: nothing reads them)
#undef KR_APPLY_STEP
#undef KR_APPLY_FILL
#undef KR_APPLY_WAIT
    if (a.fin && memwave) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (m_tq[c] >= 0) {
                const int ii = m_i[c];
                a.fin[tile_base + lane + 64 * c] = sx[cur][m_tq[c]][ii + ii / JS];
            }
        }
    }
}

// ---- block propagators of the time-blocked known-pulse sweeps, N <= 32 (round 5) ------------------------------
// P_b = W_{j1 - 1} ... W_{j0}: the step propagators of sweep positions [j0, j1) = [b S, min((b + 1) S, N_T)), position j being
// interval n = j (forward) or N_T - 1 - j (backward: U_d then holds the adjoint sweep's propagators), read and written in the
// thread-major order of kr_sweep_apply (KrApplyGeom).  One workgroup per (generator, block); the running product and the next factor
// live in LDS as plain matrices, the factor after that is already on its way (registers).  Grid = G x NB.
template <int NP>
__global__ __launch_bounds__(256, 1) void kr_tb_products_small(const KrSweepArgs a) {
    static_assert(NP <= 32, "small state dimensions only (NP = 64: kr_device_tblock.h)");
    using GA = KrApplyGeom<NP>;
    constexpr int JQ = GA::JQA, JS = GA::JS, RT = GA::RT, EL = (NP * NP + 255) / 256, LD = NP + 1;
    __shared__ double2 sR[2][NP * LD], sW[NP * LD];
    const int tid = threadIdx.x;
    const int NB = a.tb_NB, NTg = a.tb_NT, S = a.tb_S;
    const int g = blockIdx.x / NB, b = blockIdx.x % NB;
    const int j0 = b * S, j1 = (j0 + S < NTg) ? j0 + S : NTg;
    const double2 *__restrict__ Ug = a.U + (size_t)g * NTg * NP * NP;
    auto interval = [&](const int j) { return (a.dir > 0) ? j : (NTg - 1 - j); };
    // element e of the thread-major pack is U[i][c] with:
    auto row_of = [&](const int e) { return (e % RT) / JQ; };
    auto col_of = [&](const int e) { return ((e % RT) % JQ) * JS + e / RT; };
    double2 w[EL];
#pragma unroll
    for (int q = 0; q < EL; ++q) {
        const int e = tid + 256 * q;
        if (e < NP * NP) sR[0][row_of(e) * LD + col_of(e)] = Ug[(size_t)interval(j0) * NP * NP + e];
    }
    if (j0 + 1 < j1) {
#pragma unroll
        for (int q = 0; q < EL; ++q) { const int e = tid + 256 * q; if (e < NP * NP) w[q] = Ug[(size_t)interval(j0 + 1) * NP * NP + e]; }
    }
    int cur = 0;
    for (int j = j0 + 1; j < j1; ++j) {
#pragma unroll
        for (int q = 0; q < EL; ++q) { const int e = tid + 256 * q; if (e < NP * NP) sW[row_of(e) * LD + col_of(e)] = w[q]; }
        __syncthreads();   // (also: sR[cur] complete)
        if (j + 1 < j1) {
#pragma unroll
            for (int q = 0; q < EL; ++q) { const int e = tid + 256 * q; if (e < NP * NP) w[q] = Ug[(size_t)interval(j + 1) * NP * NP + e]; }
        }
#pragma unroll
        for (int q = 0; q < EL; ++q) {
            const int e = tid + 256 * q;   // plain index here: row e / NP, column e % NP
            if (e < NP * NP) {
                const int i = e / NP, c = e % NP;
                double yr = 0.0, yi = 0.0;
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const double2 x = sW[i * LD + k], y = sR[cur][k * LD + c];
                    yr = fma(x.x, y.x, yr); yr = fma(-x.y, y.y, yr);
                    yi = fma(x.x, y.y, yi); yi = fma(x.y, y.x, yi);
                }
                sR[cur ^ 1][i * LD + c] = make_double2(yr, yi);
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    __syncthreads();
    double2 *__restrict__ out = a.tb_P + ((size_t)g * NB + b) * NP * NP;
#pragma unroll
    for (int q = 0; q < EL; ++q) {
        const int e = tid + 256 * q;
        if (e < NP * NP) out[e] = sR[cur][row_of(e) * LD + col_of(e)];
    }
}

// ---- batched propagator construction on the matrix cores, NP = 64 ------------------------------------
// U_n = exp(A), A = -i sdt H(eps_n): scaling and squaring around the same Taylor/Horner polynomial,
//   Y <- I + (A / (2^q j)) Y  (j = m..1),  then q squarings,  2^q >= ||A|| (q = 0 for the usual ||A|| <= 1),
// with every 64x64x64 complex product done as 4 real products on v_mfma_f64_16x16x4_f64 (64 cycles per
// instruction per SIMD, measured: tools/ubench/mfma64.hip = the fp64 peak, reached by one wave per SIMD).
// Wave w owns output rows 16w..16w+15; its A fragments (16 k-steps x {re, im, -im}) stay in registers for all
// rounds; Y lives in LDS as separate re / im planes (ping-pong), read as MFMA B fragments.
// Fragment maps (gfx950, checked in the micro-benchmark): A: lane l holds A[l & 15][l >> 4]; B: lane l holds
// B[l >> 4][l & 15]; D register r of lane l is D[(l >> 4) + 4 r][l & 15].
typedef double kr_d4 __attribute__((ext_vector_type(4)));
#define KR_YS 66  // row stride (doubles) of the LDS planes

template <int LMAX>
__global__ __launch_bounds__(256, 1) void kr_expm_mfma64(const KrSweepArgs a) {
    constexpr int NP = 64;
    // producer-side pack geometry (how Hpack was written for NP = 64, see KrGeom)
    using G = KrGeom<NP, LMAX, 1>;
    constexpr int JS = G::JS, JQ = G::JQ, RT = G::RT;
    constexpr int JS_A = KrApplyGeom<NP>::JS, RT_A = KrApplyGeom<NP>::RT, JQ_A = KrApplyGeom<NP>::JQA;
    extern __shared__ __attribute__((aligned(16))) double kr_lds[];
    // planes: [buffer b][re | im][64][KR_YS]
#define KR_YR(b) (kr_lds + (size_t)(b) * 2 * NP * KR_YS)
#define KR_YI(b) (kr_lds + (size_t)(b) * 2 * NP * KR_YS + NP * KR_YS)

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const double my_rinv = kr_rinv[lane <= KR_MAX_ORDER ? lane : 0];
    const double my_thr = kr_load_my_thr(a);
    const KrGen<LMAX> gx = kr_gen_ctx<LMAX>(a, blockIdx.x / a.NT);
    const int NT = a.NT, L = a.L;
    const int n = blockIdx.x % NT;
    const double sdt = (a.dir > 0) ? a.dt[n] : -a.dt[n];

    double e[LMAX];
    double theta = gx.nrm[0];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
        e[l] = (l < L) ? a.eps[(size_t)l * NT + n] : 0.0;
        theta += fabs(e[l]) * gx.nrm[1 + l];
    }
    theta *= fabs(sdt);
    int q = 0;
    while (theta > KR_THETA_MAX && q < 30) { theta *= 0.5; ++q; }
    int s_unused, m;
    kr_plan_fast(a, my_thr, theta, s_unused, m);   // theta <= 1 here: table path
    // economized polynomial of the step (KrEcon, kr_device.h): rows exist for Hermitian generators with ||H dt|| <= 1 (q = 0)
    const double ec = (a.econ && q == 0) ? a.econ[((size_t)gx.gen * NT + n) * KR_ECON_W + (lane & (KR_ECON_W - 1))] : 0.0;
    const KrEcon eco = kr_econ_select(ec, theta, s_unused, m);
    const double my_g = eco.on ? ec : my_rinv;
    const double fs = ldexp(sdt, -q);

    // A fragments: A = -i fs H  ->  Ar = fs Hi,  Ai = -fs Hr  (row 16w + li, column 4t + lk)
    double Ar[16], Ai[16];
    const int row = 16 * w + li;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int col = 4 * t + lk;
        const size_t off = (size_t)(col % JS) * RT + (size_t)row * JQ + (col / JS);
        const double2 h0 = gx.Hpack[off];
        double hr = h0.x, hi = h0.y;
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            if (l < L) {
                const double2 hl = gx.Hpack[(size_t)(1 + l) * JS * RT + off];
                hr = fma(e[l], hl.x, hr);
                hi = fma(e[l], hl.y, hi);
            }
        }
        Ar[t] = fs * hi;
        Ai[t] = -fs * hr;
    }
    int cur = 0;
    // ---- Paterson-Stockmeyer evaluation for 5 <= m <= 7 (round 3): p(A) = B0 + B1 A^4, Bk = c_4k + c_4k+1 A + c_4k+2 A^2 + c_4k+3 A^3
    //      -- four products (A^2, A^3, A^4, B1 A^4) against m - 1 = 4..6 Horner rounds.  (A second level for degrees 8..11 keeps B1 in
    //      the D layout as well and spills: those degrees -- Taylor without the economized rows -- stay with the Horner form.)
    //      Left operands are register fragments (A, A^2 and the Bk are linear combinations of the fragments of A, A^2, A^3: all
    //      powers commute, so the Bk go to the LEFT of A^4), right operands come from the two LDS buffers; products leave the
    //      matrix cores in the D layout, B0 is accumulated there as the powers appear, and a power needed as a left
    //      operand takes the round trip through LDS that the squaring rounds below take as well.  Same coefficients as the Horner
    //      form (c_0 = a_0, c_1 = a_1 resp. 1, c_j = c_{j-1} g_j), results differ by rounding only.
    const bool ps = (a.use_mfma & 2) != 0 && m >= 5 && m <= 7;
    if (ps) {
        double c[8];
        c[0] = eco.a0;
        c[1] = kr_rinv_lane(my_g, 1);
#pragma unroll
        for (int j = 2; j < 8; ++j) c[j] = (j <= m) ? c[j - 1] * kr_rinv_lane(my_g, j) : 0.0;
        // one product: (Lr + i Li) x (buffer b)  ->  sink(ct, re[4], im[4]) per column tile, D layout (row 16 w + lk + 4 r, column 16 ct + li)
        // (round 5: THREE real products per complex one -- re = Lr Br - Li Bi, im = (Lr + Li)(Br + Bi) - Lr Br - Li Bi --, the operand
        //  sums are vector adds in the shadow of the matrix cores: a quarter fewer MFMAs, which bound this kernel)
        auto product = [&](const double (&Lr)[16], const double (&Li)[16], const int b, auto sink) {
            double Ls[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) Ls[t] = Lr[t] + Li[t];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                kr_d4 pa = {0, 0, 0, 0}, pb = {0, 0, 0, 0}, pc = {0, 0, 0, 0};
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const double br = KR_YR(b)[(4 * t + lk) * KR_YS + 16 * ct + li];
                    const double bi = KR_YI(b)[(4 * t + lk) * KR_YS + 16 * ct + li];
                    const double bs = br + bi;
                    pa = __builtin_amdgcn_mfma_f64_16x16x4f64(Lr[t], br, pa, 0, 0, 0);
                    pb = __builtin_amdgcn_mfma_f64_16x16x4f64(Li[t], bi, pb, 0, 0, 0);
                    pc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ls[t], bs, pc, 0, 0, 0);
                }
                sink(ct, pa - pb, (pc - pa) - pb);
            }
        };
        auto store_d = [&](const int b, const int ct, const kr_d4 vr, const kr_d4 vi) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                KR_YR(b)[(16 * w + lk + 4 * r) * KR_YS + 16 * ct + li] = vr[r];
                KR_YI(b)[(16 * w + lk + 4 * r) * KR_YS + 16 * ct + li] = vi[r];
            }
        };
        auto load_frag = [&](const int b, double (&Fr)[16], double (&Fi)[16]) {
#pragma unroll
            for (int t = 0; t < 16; ++t) { Fr[t] = KR_YR(b)[row * KR_YS + 4 * t + lk]; Fi[t] = KR_YI(b)[row * KR_YS + 4 * t + lk]; }
        };
        // A -> buffer 0 (fragment positions)
#pragma unroll
        for (int t = 0; t < 16; ++t) { KR_YR(0)[row * KR_YS + 4 * t + lk] = Ar[t]; KR_YI(0)[row * KR_YS + 4 * t + lk] = Ai[t]; }
        __syncthreads();
        // B0 in the D layout: identity and A terms
        kr_d4 b0r[4], b0i[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int orow = 16 * w + lk + 4 * r, ocol = 16 * ct + li;
                const double ar = KR_YR(0)[orow * KR_YS + ocol], ai = KR_YI(0)[orow * KR_YS + ocol];
                const double dg = (orow == ocol) ? 1.0 : 0.0;
                b0r[ct][r] = fma(c[1], ar, c[0] * dg); b0i[ct][r] = c[1] * ai;
            }
        }
        // A^2 = A A -> buffer 1
        product(Ar, Ai, 0, [&](const int ct, const kr_d4 vr, const kr_d4 vi) {
            store_d(1, ct, vr, vi);
            b0r[ct] += c[2] * vr; b0i[ct] += c[2] * vi;
        });
        __syncthreads();
        double A2r[16], A2i[16];
        load_frag(1, A2r, A2i);
        // A^3 = A A^2 -> buffer 0 (A itself is no longer needed there: its readers were the product above)
        product(Ar, Ai, 1, [&](const int ct, const kr_d4 vr, const kr_d4 vi) {
            store_d(0, ct, vr, vi);
            b0r[ct] += c[3] * vr; b0i[ct] += c[3] * vi;
        });
        __syncthreads();
        double Fr[16], Fi[16];   // A^3 fragments, then the left operand B1
        load_frag(0, Fr, Fi);
        __syncthreads();         // everybody holds its A^3 fragments: buffer 0 may take A^4
        // A^4 = A^2 A^2 -> buffer 0
        product(A2r, A2i, 1, [&](const int ct, const kr_d4 vr, const kr_d4 vi) { store_d(0, ct, vr, vi); });
        // left operand of the last product: B1 = c_4 + c_5 A + c_6 A^2 + c_7 A^3, as fragments
        {
            constexpr int kb = 4;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const double dg = (row == 4 * t + lk) ? 1.0 : 0.0;
                const double fr = fma(c[kb + 3], Fr[t], fma(c[kb + 2], A2r[t], fma(c[kb + 1], Ar[t], c[kb] * dg)));
                const double fi = fma(c[kb + 3], Fi[t], fma(c[kb + 2], A2i[t], c[kb + 1] * Ai[t]));
                Fr[t] = fr; Fi[t] = fi;
            }
        }
        __syncthreads();         // A^4 complete in buffer 0
        // U = B0 + B1 A^4 -> buffer 1 (its A^2 has been read by everybody: the barrier above)
        product(Fr, Fi, 0, [&](const int ct, const kr_d4 vr, const kr_d4 vi) { store_d(1, ct, b0r[ct] + vr, b0i[ct] + vi); });
        __syncthreads();
        cur = 1;
    } else
    // Horner round j = m needs no product: Y_{m-1} = c I + f_m A, written from the A fragments (c = a_0 when it is also the last)
    {
        const double fm = kr_rinv_lane(my_g, m), idm = (m == 1) ? eco.a0 : 1.0;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int col = 4 * t + lk;
            KR_YR(0)[row * KR_YS + col] = fma(fm, Ar[t], (row == col) ? idm : 0.0);
            KR_YI(0)[row * KR_YS + col] = fm * Ai[t];
        }
    }
    __syncthreads();

    const int rounds = (m - 1) + q;
    for (int rd = ps ? (m - 1) : 0; rd < rounds; ++rd) {   // (Paterson-Stockmeyer has done the polynomial: squarings only)
        const bool horner = (rd < m - 1);
        const int j = m - 1 - rd;  // Horner index m-1..1
        const double f = horner ? kr_rinv_lane(my_g, j) : 1.0;
        const double idc = (horner && j == 1) ? eco.a0 : 1.0;
        if (!horner) {  // squaring: the A fragments become this wave's row tile of Y
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const double yr = KR_YR(cur)[row * KR_YS + 4 * t + lk], yi = KR_YI(cur)[row * KR_YS + 4 * t + lk];
                Ar[t] = yr; Ai[t] = yi;
            }
        }
        double As[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) As[t] = Ar[t] + Ai[t];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            kr_d4 pa = {0, 0, 0, 0}, pb = {0, 0, 0, 0}, pc = {0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const double br = KR_YR(cur)[(4 * t + lk) * KR_YS + 16 * ct + li];
                const double bi = KR_YI(cur)[(4 * t + lk) * KR_YS + 16 * ct + li];
                const double bs = br + bi;
                pa = __builtin_amdgcn_mfma_f64_16x16x4f64(Ar[t], br, pa, 0, 0, 0);
                pb = __builtin_amdgcn_mfma_f64_16x16x4f64(Ai[t], bi, pb, 0, 0, 0);
                pc = __builtin_amdgcn_mfma_f64_16x16x4f64(As[t], bs, pc, 0, 0, 0);
            }
            const kr_d4 cr = pa - pb, ci = (pc - pa) - pb;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int orow = 16 * w + lk + 4 * r, ocol = 16 * ct + li;
                double vr = cr[r] * f, vi = ci[r] * f;
                if (horner && orow == ocol) vr += idc;
                KR_YR(cur ^ 1)[orow * KR_YS + ocol] = vr;
                KR_YI(cur ^ 1)[orow * KR_YS + ocol] = vi;
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    // U_n -> the consumer's operand order.  u_layout = 1 (kr_mf_apply_sweep, kr_device_mfma.h): MFMA A operands
    //   Umf[wave][slot][lane] = {-Ui, Ur} of U[16 wave + 4 blk + f0][4 ((4 wave + slot) & 15) + f2],  lane = (f0, blk, f2)
    double2 *Un = a.U + ((size_t)gx.gen * NT + n) * NP * NP;
    if (a.u_layout == 1) {
        for (int e = tid; e < 4096; e += 256) {
            const int l = e & 63, sl = (e >> 6) & 15, wv = e >> 10;
            const int r = 16 * wv + 4 * ((l >> 2) & 3) + (l & 3), cc = 4 * ((4 * wv + sl) & 15) + (l >> 4);
            Un[e] = make_double2(-KR_YI(cur)[r * KR_YS + cc], KR_YR(cur)[r * KR_YS + cc]);
        }
        return;
    }
    if (a.u_layout == 2) {   // fragment order of the time-blocked sweep (kr_device_tblock.h): [(16 w' + t) * 64 + lane] = {Ur, Ui} of U[16 w' + (lane & 15)][4 t + (lane >> 4)]
        for (int e = tid; e < 4096; e += 256) {
            const int l = e & 63, t = (e >> 6) & 15, wt = e >> 10;
            const int r = 16 * wt + (l & 15), cc = 4 * t + (l >> 4);
            Un[e] = make_double2(KR_YR(cur)[r * KR_YS + cc], KR_YI(cur)[r * KR_YS + cc]);
        }
        return;
    }
    // u_layout = 0: thread-major pack of kr_sweep_apply: Upack[jj * 256 + i*4 + jq] = U[i][jq*16 + jj]
    for (int t2 = tid; t2 < RT_A; t2 += 256) {   // consumer thread t2 = (i, jq)
        const int i = t2 / JQ_A, jq = t2 % JQ_A;
#pragma unroll
        for (int jj = 0; jj < JS_A; ++jj) {
            const int c = jq * JS_A + jj;
            Un[(size_t)jj * RT_A + t2] = make_double2(KR_YR(cur)[i * KR_YS + c], KR_YI(cur)[i * KR_YS + c]);
        }
    }
}
