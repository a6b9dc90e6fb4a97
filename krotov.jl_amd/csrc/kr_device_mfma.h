// Update sweep on the matrix cores for NP = 64: v_mfma_f64_4x4x4_4b_f64 (four 4x4x4 blocks per instruction).
//
// Why this instruction: it issues every 16-17 cycles per SIMD from one wave with >= 2 independent accumulators
// (tools/ubench/mfma4x4.hip; 22 on one dependent chain) = 512 flop / 16 cycles = the fp64 peak, the same rate as
// v_mfma_f64_16x16x4_f64 -- but its tile is 16 rows x 4 columns, so ONE workgroup needs only 2 trajectories to fill it.
// The register-resident VALU sweep (kr_device.h) reaches that rate only on paper: next to the 32 FMAs of a row slice it
// spends 8 LDS reads, 12 DPP moves and 6 adds per trajectory and round, and with two or four trajectories per team
// (K = 512 / 1024 per GPU at N = 64) or more than two control operators it spills.
//
// Packed complex layout.  A 4-column tile carries 2 trajectories: column j = 2 t + c holds component c (0 = re, 1 = im) of
// trajectory t.  For a complex matrix H = Hr + i Hi and a vector w,   packed(H w) = Hr * packed(w) + Hi * packed(i w),
// so one complex product is two real MFMA chains and the result comes out packed again.  The products are taken of
// z = -i w, i.e.   D = Hr * packed(-i w) + Hi * packed(w) = packed(-i H w),   which is exactly what the Taylor recurrence
// needs:   y_{j-1} = psi + (fs / j) (-i H y_j)   is then LANE-LOCAL (each lane holds one real component of one element: no
// cross-lane traffic, no complex arithmetic), and so are the update integrand Im<chi|H_l|psi> = sum chi_c (-i H_l psi)_c
// and the first-application shortcut.
// Every vector lives in LDS as an array of double2 W[row * 4 + column] = {packed(-i w), packed(w)}: the two B operands of
// a k-block come with ONE 16-byte read.  (With two separate 8-byte arrays hipcc merges neighbouring k-blocks into
// ds_read2_b64, which moves 128 B/clk against 256 B/clk for ds_read_b128 -- MI355X_MICROARCH.md, LDS table -- and the four
// waves' operand reads then take as long as the MFMAs themselves: tools/ubench/mfround.hip.)
//
// Mapping: 256 threads = 4 waves, one per SIMD (up to 512 registers per lane: the A operands of H_1..H_L and H(eps), 64
// registers each, stay resident for the whole sweep, also for L = 3, 4; the drift H0 lives in LDS, 64 KB).  Wave w owns
// rows 16 w .. 16 w + 15 (its four 4x4 blocks), lane l = (f0 = l & 3, blk = (l >> 2) & 3, f2 = l >> 4):
//      A operand, k-block kb : H[16 w + 4 blk + f0][4 kb + f2]          (lane map measured: tools/ubench/mfma4x4.hip)
//      B operand, k-block kb : X[4 kb + f2][column f0]                  (the same for all four blocks: LDS broadcast)
//      D                     : Y[16 w + 4 blk + f2][column f0]
// A round = 16 k-blocks x 2 chains = 32 MFMAs per tile and wave (544 cycles), one 16-byte LDS read per pair of MFMAs, two
// 8-byte LDS writes per tile, one s_barrier.  TILES tiles per workgroup (KT = 2 TILES trajectories).
//
// Reference semantics: kr_sweep_update (kr_device.h), i.e. /root/reference/src/optimize.jl:321-370.
#pragma once
#include "kr_device.h"

__device__ __forceinline__ double kr_mfma4(const double a, const double b, const double c) {
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

#define KR_MF_TW 256  // doubles per packed tile array: 64 rows x 4 columns

// D[m][t] = packed(-i H_m w)[this lane] for NM matrices that share the B operands: W = the TILES tiles' operand arrays
// (double2 {packed(-i w), packed(w)}, KR_MF_TW elements per tile).
// k-blocks are visited in a per-wave ROTATED order: slot s = 0..15 stands for k-block (kb0 + s) & 15 with kb0 = 4 * wave, so
// that slots 0..3 are the rows the wave itself produces.  All A-operand storage is in slot order (Ar[m][s], Ai[m][s] and
// the LDS copy of the drift, A0[s * 256 + tid] = {re, im}); matrix 0 comes from A0 when LDSA0, matrices 1.. then from
// Ar[m - 1], Ai[m - 1].
// OWNFIRST (the Taylor rounds): the caller has just written its own rows of the vector into W and NOT yet synchronised.
// The operands of slots 0..3 are requested at once (same-wave LDS write -> read needs no barrier), then comes the workgroup
// barrier, then the requests for the other waves' rows -- whose latency the MFMAs of the own slots cover.
// Software pipeline written out because hipcc sinks the LDS reads next to their MFMAs when registers are tight (2.2k
// instead of 1.1k cycles per 64-MFMA round): the loop runs over PAIRS of slots; the operands of pair p + PD are requested
// before the MFMAs of pair p issue, and a scheduling barrier after every pair keeps that order.
template <int TILES, int NM, bool LDSA0, bool OWNFIRST>
__device__ __forceinline__ void kr_mf_multi(const double2 *__restrict__ A0, const double (*__restrict__ Ar)[16],
                                            const double (*__restrict__ Ai)[16], const double2 *__restrict__ W,
                                            const int boff, const int kb0, double (&d)[NM][TILES]) {
    constexpr int PER = 4 * TILES * NM;                           // MFMAs per pair of slots (16-17 cycles each)
    constexpr int PD = (PER >= 16) ? 1 : ((PER >= 8) ? 2 : 3);    // pairs in flight ahead
    constexpr int P0 = OWNFIRST ? 2 : PD;                         // pairs requested before the first MFMA
    double pr[NM][TILES], pi[NM][TILES];
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int t = 0; t < TILES; ++t) { pr[m][t] = 0.0; pi[m][t] = 0.0; }
    double2 b[8][TILES][2], a0[8][2];
    // pair p = slots 2p, 2p + 1 = k-blocks kb, kb + 1 with kb = (kb0 + 2p) & 15 (kb0 is a multiple of 4: a pair never wraps)
#define KR_MF_REQUEST(p)                                                                                               \
    {                                                                                                                  \
        const int pb_ = (((kb0 + 2 * (p)) & 15) << 4) + boff;                                                          \
        _Pragma("unroll") for (int u_ = 0; u_ < 2; ++u_) {                                                             \
            _Pragma("unroll") for (int t_ = 0; t_ < TILES; ++t_) b[p][t_][u_] = W[t_ * KR_MF_TW + pb_ + 16 * u_];      \
            if (LDSA0) a0[p][u_] = A0[(2 * (p) + u_) * 256 + threadIdx.x];                                             \
        }                                                                                                              \
    }
#pragma unroll
    for (int p = 0; p < P0; ++p) KR_MF_REQUEST(p)
    if (OWNFIRST) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = P0; p < P0 + PD - 1; ++p) KR_MF_REQUEST(p)
    }
    constexpr int AHEAD = OWNFIRST ? (P0 + PD - 1) : PD;          // pairs requested so far
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        if (p + AHEAD < 8) KR_MF_REQUEST(p + AHEAD)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const int mr = LDSA0 ? (m > 0 ? m - 1 : 0) : m;   // register matrices follow the LDS one
                const double ar = (LDSA0 && m == 0) ? a0[p][u].x : Ar[mr][2 * p + u];
                const double ai = (LDSA0 && m == 0) ? a0[p][u].y : Ai[mr][2 * p + u];
#pragma unroll
                for (int t = 0; t < TILES; ++t) {
                    pr[m][t] = kr_mfma4(ar, b[p][t][u].x, pr[m][t]);
                    pi[m][t] = kr_mfma4(ai, b[p][t][u].y, pi[m][t]);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#undef KR_MF_REQUEST
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int t = 0; t < TILES; ++t) d[m][t] = pr[m][t] + pi[m][t];
}

// one matrix from registers; OWNFIRST: see kr_mf_multi (the workgroup barrier of the round is INSIDE)
template <int TILES, bool OWNFIRST = false>
__device__ __forceinline__ void kr_mf_apply(const double (&Ar)[16], const double (&Ai)[16], const double2 *__restrict__ W,
                                            const int boff, const int kb0, double (&d)[TILES]) {
    double dd[1][TILES];
    kr_mf_multi<TILES, 1, false, OWNFIRST>(nullptr, &Ar, &Ai, W, boff, kb0, dd);
#pragma unroll
    for (int t = 0; t < TILES; ++t) d[t] = dd[0][t];
}

// this lane's component yv of a vector w -> its two places in the operand array: packed(w) here, packed(-i w) = (w_i, -w_r)
// in the neighbouring column
__device__ __forceinline__ void kr_mf_put(double2 *W, const int doff, const int zoff, const double zs, const double yv) {
    ((double *)W)[2 * doff + 1] = yv;
    ((double *)W)[2 * zoff] = zs * yv;
}

// CAL = 1: the same kernel under another name for the short sweeps of calibrate_exchange (krotov_hip.hip), so that profiler
// statistics of kr_mf_update<..., 0> hold the caller's launches only.
// SETS > 1 (round 3): the workgroup time-multiplexes SETS sets of TILES tiles within a step -- K beyond 4 trajectories per CU
// (K = 2048 / 4096 per GPU at N = 64) without leaving the persistent kernel: phase A runs set by set and ACCUMULATES the partial
// sums, ONE exchange and ONE pulse update / H(eps_new) serve all sets, then the polynomial rounds run set by set (the products
// D of every set stay in registers across the exchange, so the first application is free for every set).  LDS: the drift (64 KB)
// and the Horner ping-pong are shared; per set ONE psi array and ONE chi array (no parity pair: psi(t_{n+1}) overwrites psi(t_n)
// in place once the exchange's barriers lie behind every reader, chi(t_{n+1}) is handed over after the set's rounds), 16 KB per
// set at two tiles -> 4 sets in 144 KB.  The general (non-HERM) variant only.
// POLY = 1 (round 5): polynomial amplitudes f_l of degree <= 3 (kr_sweep_update, kr_device.h): a separate instantiation.
template <int LMAX, int TILES, int SCHED, int HERM, int CAL, int SETS = 1, int POLY = 0>
__global__ __launch_bounds__(256, 1) void kr_mf_update(const KrSweepArgs a) {
    static_assert(SETS == 1 || HERM == 0, "several sets per workgroup: general variant only");
    constexpr int KT = 2 * TILES, TW = KR_MF_TW, NW = 4;
    constexpr int CH = KT;  // 16-B elements per lane when the memory wave moves a whole tile (KT * 64 / 64)
    constexpr int NB = (SETS == 1) ? 2 : SETS;   // psi / chi arrays: a parity pair for one set, one array per set otherwise
    __shared__ __attribute__((aligned(16))) double2 sx[NB][TILES][TW];  // psi(t_n) as operand array (SETS = 1: by parity `cur`)
    __shared__ __attribute__((aligned(16))) double2 sy[2][TILES][TW];  // Horner vectors, ping-pong
    __shared__ __attribute__((aligned(16))) double2 sc[NB][TILES][TW];  // chi(t_k) (SETS = 1: in slot k & 1; a matvec operand when HERM)
    __shared__ __attribute__((aligned(16))) double2 sH0[16 * 256];     // A operands of the drift H0, [slot][tid] = {re, im} (64 KB)
    __shared__ double spart[LMAX * NW];
    __shared__ double stot[LMAX + 1];

    const int tid = threadIdx.x, wg = blockIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f0 = lane & 3, blk = (lane >> 2) & 3, f2 = lane >> 4;
    const int arow = 16 * wave + 4 * blk + f0;   // row of this lane's A elements
    const int drow = 16 * wave + 4 * blk + f2;   // row of this lane's D element
    const int dt = f0 >> 1, dc = f0 & 1;         // trajectory within the tile / component of this lane's D element
    const int boff = f2 * 4 + f0;                // B operand: X[4 kb + f2][f0]
    const int kb0 = 4 * wave;                    // slot s of every A-operand array = k-block (kb0 + s) & 15 (kr_mf_multi)
    const int doff = drow * 4 + f0;              // where this lane's component lives in a packed array
    const int zoff = drow * 4 + (f0 ^ 1);        // -i w: (w_i, -w_r) -> this lane's component goes to the OTHER column
    const double zs = dc ? 1.0 : -1.0;           //        with this sign
    const bool memwave = (wave == NW - 1);
    const double my_rinv = kr_rinv[lane <= KR_MAX_ORDER ? lane : 0];
    const double my_thr = kr_load_my_thr(a);
    const KrGen<LMAX> gx = kr_gen_ctx<LMAX>(a, a.wg_gen ? a.wg_gen[blockIdx.x] : 0);
    const int N = a.N, NT = a.NT, L = a.L;
    const size_t slab = (size_t)a.Kp * N;
    const size_t tile_base = (size_t)wg * SETS * KT * N;  // the workgroup's trajectories are contiguous in every [Kp][N] slab, set after set
    const size_t set_stride = (size_t)KT * N;
    const int tile_elems = KT * N;
    const bool row_ok = drow < N;
    // which psi / chi array: one set -> parity; several sets -> the set's own
#define KR_SX_RD(st) sx[SETS == 1 ? cur : (st)]
#define KR_SX_WR(st) sx[SETS == 1 ? (cur ^ 1) : (st)]
#define KR_SC_AT(st, k) sc[SETS == 1 ? ((k) & 1) : (st)]

    // A operands (dense row-major [m][64][64], zero padded): 16 k-blocks x (re, im) per matrix; H_1..H_L in registers,
    // the drift H0 in LDS
    double Hr[LMAX][16], Hi[LMAX][16];
    {
        const double2 *Hd = a.Hdense + (size_t)gx.gen * a.dense_stride;
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {   // kb = slot; its k-block is kq
            const int kq = (kb0 + kb) & 15;
            const double2 h0 = Hd[(size_t)arow * 64 + 4 * kq + f2];
            sH0[kb * 256 + tid] = h0;
#pragma unroll
            for (int l = 0; l < LMAX; ++l) {
                double2 h = make_double2(0.0, 0.0);
                if (l < L) h = Hd[((size_t)(1 + l) * 64 + arow) * 64 + 4 * kq + f2];
                Hr[l][kb] = h.x;
                Hi[l][kb] = h.y;
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
        m_pos[c] = (tq >> 1) * TW + i * 4 + 2 * (tq & 1);   // columns 2 (tq & 1) and + 1 of row i: re and im of the element
    }
    double2 cpf[CH];  // chi prefetch registers (memory wave)

    // initial state into registers (D layout) and LDS (packed psi and -i psi); rows >= N stay zero
    double psi[SETS][TILES];
#pragma unroll
    for (int st = 0; st < SETS; ++st) {
        const size_t sb = tile_base + st * set_stride;
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            double v = 0.0;
            if (row_ok) v = ((const double *)a.init)[(sb + (size_t)(2 * t + dt) * N + drow) * 2 + dc];
            psi[st][t] = v;
            kr_mf_put(sx[SETS == 1 ? 0 : st][t], doff, zoff, zs, v);
            double c0 = 0.0, c1 = 0.0;
            if (row_ok) {
                c0 = ((const double *)a.X)[(sb + (size_t)(2 * t + dt) * N + drow) * 2 + dc];              // chi(t_0)
                if (HERM && NT > 1) c1 = ((const double *)a.X)[(slab + sb + (size_t)(2 * t + dt) * N + drow) * 2 + dc];  // chi(t_1)
            }
            kr_mf_put(sc[SETS == 1 ? 0 : st][t], doff, zoff, zs, c0);
            if (SETS == 1) kr_mf_put(sc[1][t], doff, zoff, zs, c1);
        }
        if (memwave && a.store) {
#pragma unroll
            for (int c = 0; c < CH; ++c)
                if (m_ok[c]) a.store[sb + lane + 64 * c] = a.init[sb + lane + 64 * c];
        }
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
        if (HERM && l < L) kr_mf_apply<TILES>(Hr[l], Hi[l], &sc[0][0][0], boff, kb0, cl[l]);
    }

    const bool stamp = (a.debug != nullptr) && (wg == 0);
    unsigned long long tA = 0, tX = 0, tB = 0, tS = 0, tB1 = 0, tB2 = 0, t_prev = 0, rt0 = 0;
    if (stamp) { t_prev = __builtin_amdgcn_s_memtime(); rt0 = kr_realtime(); }
    int cur = 0;
    double eg_nx[LMAX], al_nx[LMAX], dt_nx = a.dt[0];
    double fp_nx[POLY ? LMAX : 1], ag_nx[POLY ? LMAX : 1];   // POLY: f_l'(eps_guess), f_l(eps_guess) of the next step
    double pc2[POLY ? LMAX : 1], pc3[POLY ? LMAX : 1];       //       the quadratic and cubic coefficients of f_l
    if (POLY) {
#pragma unroll
        for (int l = 0; l < LMAX; ++l) { pc2[l] = (l < L) ? a.poly[4 * l + 2] : 0.0; pc3[l] = (l < L) ? a.poly[4 * l + 3] : 0.0; }
    }
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
        eg_nx[l] = (l < L) ? a.eps[(size_t)l * NT] : 0.0;
        al_nx[l] = (l < L) ? a.alpha[(size_t)l * NT] : 0.0;
        if (POLY) { fp_nx[l] = (l < L) ? a.fprime_g[(size_t)l * NT] : 0.0; ag_nx[l] = (l < L) ? a.amp_g[(size_t)l * NT] : 0.0; }
    }
    const double *econ_rows = a.econ ? a.econ + ((size_t)gx.gen * NT) * KR_ECON_W + (lane & (KR_ECON_W - 1)) : nullptr;
    double ec_nx = econ_rows ? econ_rows[0] : 0.0;
    double *const dulog = a.dulog + (size_t)wg * NT * LMAX;   // (kr_sweep_update, kr_device.h)
    for (int n = 0; n < NT; ++n) {
        const double dtn = dt_nx;
        const double ec = ec_nx;
        double eg[LMAX], al[LMAX];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) { eg[l] = eg_nx[l]; al[l] = al_nx[l]; }
        double fpg[POLY ? LMAX : 1], ag[POLY ? LMAX : 1];
        if (POLY) {
#pragma unroll
            for (int l = 0; l < LMAX; ++l) { fpg[l] = fp_nx[l]; ag[l] = ag_nx[l]; }
        }
        if (n + 1 < NT) {
            dt_nx = a.dt[n + 1];
            if (econ_rows) ec_nx = econ_rows[(size_t)(n + 1) * KR_ECON_W];
#pragma unroll
            for (int l = 0; l < LMAX; ++l) {
                if (l < L) { eg_nx[l] = a.eps[(size_t)l * NT + n + 1]; al_nx[l] = a.alpha[(size_t)l * NT + n + 1]; }
                if (POLY && l < L) { fp_nx[l] = a.fprime_g[(size_t)l * NT + n + 1]; ag_nx[l] = a.amp_g[(size_t)l * NT + n + 1]; }
            }
        }
        // D[set][m][t] = packed(-i H_m psi)[this lane], m = 0 (drift), 1..L (controls)
        double D[SETS][LMAX + 1][TILES];
        double part[LMAX];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) part[l] = 0.0;
        auto products = [&](const int st) {   // (L is wave-uniform; LMAX = 4 covers L = 3 with a zero matrix)
            if (L == 1) {
                double dd[2][TILES];
                kr_mf_multi<TILES, 2, true, false>(sH0, Hr, Hi, &KR_SX_RD(st)[0][0], boff, kb0, dd);
#pragma unroll
                for (int t = 0; t < TILES; ++t) {
                    D[st][0][t] = dd[0][t]; D[st][1][t] = dd[1][t];
#pragma unroll
                    for (int l = 1; l < LMAX; ++l) D[st][1 + l][t] = 0.0;
                }
            } else {
                kr_mf_multi<TILES, 1 + LMAX, true, false>(sH0, Hr, Hi, &KR_SX_RD(st)[0][0], boff, kb0, D[st]);
            }
        };
        // the same split in two: the control products (all the partial sums need) and the drift product (which only the free first
        // application needs: it moves into the exchange's window -- round 3; it used to sit in front of the exchange)
        auto products_ctrl = [&](const int st) {
            if (L == 1) {
                double dd[1][TILES];
                kr_mf_multi<TILES, 1, false, false>(nullptr, Hr, Hi, &KR_SX_RD(st)[0][0], boff, kb0, dd);
#pragma unroll
                for (int t = 0; t < TILES; ++t) {
                    D[st][1][t] = dd[0][t];
#pragma unroll
                    for (int l = 1; l < LMAX; ++l) D[st][1 + l][t] = 0.0;
                }
            } else {
                double dc[LMAX][TILES];
                kr_mf_multi<TILES, LMAX, false, false>(nullptr, Hr, Hi, &KR_SX_RD(st)[0][0], boff, kb0, dc);
#pragma unroll
                for (int l = 0; l < LMAX; ++l)
#pragma unroll
                    for (int t = 0; t < TILES; ++t) D[st][1 + l][t] = dc[l][t];
            }
        };
        auto product_drift = [&](const int st) {
            double dd[1][TILES];
            kr_mf_multi<TILES, 1, true, false>(sH0, Hr, Hi, &KR_SX_RD(st)[0][0], boff, kb0, dd);
#pragma unroll
            for (int t = 0; t < TILES; ++t) D[st][0][t] = dd[0][t];
        };
        if (!HERM) {
            // ---- phase A: Im<chi|H_l|psi> = Im(conj(chi) a), a = H_l psi = i D  ->  chi_r D_r + chi_i D_i, lane by lane;
            //      set after set into the same partial sums ----
#pragma unroll
            for (int st = 0; st < SETS; ++st) {
                products_ctrl(st);
#pragma unroll
                for (int t = 0; t < TILES; ++t) {
                    const double chc = KR_SC_AT(st, n)[t][doff].y;
#pragma unroll
                    for (int l = 0; l < LMAX; ++l)
                        if (l < L && ((gx.present >> l) & 1u)) part[l] = fma(chc, D[st][1 + l][t], part[l]);
                }
            }
        } else {
            // ---- Im<H_l chi|psi> = Im(conj(c) psi), c = H_l chi = i cl  ->  -(cl_r psi_r + cl_i psi_i) ----
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
#pragma unroll
                for (int l = 0; l < LMAX; ++l)
                    if (l < L && ((gx.present >> l) & 1u)) part[l] = fma(-cl[l][t], psi[0][t], part[l]);
            }
        }
        kr_wave_sum_all<LMAX>(part, L);   // fixed order: bitwise reproducible
        auto window = [&](auto &mid) {
            if (HERM) {
                products(0);   // (no early poll pass here -- mid() of kr_exchange: measured +0.2 % at two tiles, +0.9 % at one)
                (void)mid;
                // c_l(n+1) = -i H_l chi(t_{n+1}): the control matrices share their B operands (one pass over chi; round 3 -- two
                // separate products cost 2 x 1.0 k / 1.8 k cycles at one / two tiles, together 1.3 k / 2.4 k)
                if (n + 1 < NT) {
                    if (L == 1) {
                        kr_mf_apply<TILES>(Hr[0], Hi[0], &sc[(n + 1) & 1][0][0], boff, kb0, cl[0]);
                    } else {
                        double dc[LMAX][TILES];
                        kr_mf_multi<TILES, LMAX, false, false>(nullptr, Hr, Hi, &sc[(n + 1) & 1][0][0], boff, kb0, dc);
#pragma unroll
                        for (int l = 0; l < LMAX; ++l)
#pragma unroll
                            for (int t = 0; t < TILES; ++t) if (l < L) cl[l][t] = dc[l][t];
                    }
                }
            } else {
                (void)mid;
#pragma unroll
                for (int st = 0; st < SETS; ++st) product_drift(st);
            }
        };
        if (stamp) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tA += t - t_prev; t_prev = t; }

        // ---- global sum over all trajectories of all workgroups (and GPUs): one value per wave goes in ----
        double du[LMAX];
        if (!kr_exchange<LMAX, NW, SCHED>(a, n, wg, lane == 0, wave, part, du, spart, stot, window)) return;
        if (tid == 128) {
#pragma unroll
            for (int l = 0; l < LMAX; ++l) dulog[(size_t)n * LMAX + l] = (l == 0 && a.rs_fault == 2 && wg == 1 && n == 3) ? -du[l] : du[l];
        }
        if (stamp) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tX += t - t_prev; t_prev = t; }

        // ---- pulse update (optimize.jl:351-358), identical on every workgroup ----
        double e[LMAX];
        double theta = gx.nrm[0];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            e[l] = 0.0;
            if (l < L) {
                const double alpha = al[l];
                // POLY: mu_l = f_l'(eps_guess) H_l (optimize.jl:337,345); e[l] then holds the AMPLITUDE f_l(eps_new) the matrices are combined
                // with (the pulse value itself is formed behind the sweep, kr_pulse_finish)
                const double dul = POLY ? __dmul_rn(fpg[l], du[l]) : du[l];
                const double en = __dadd_rn(eg[l], __dmul_rn(alpha, dul));
                e[l] = POLY ? (ag[l] + kr_poly3_delta(en - eg[l], fpg[l], fma(3.0 * pc3[l], eg[l], pc2[l]), pc3[l])) : en;
                const double ad = fabs(dul);
                gint[l] = __dadd_rn(gint[l], __dmul_rn(__dmul_rn(alpha, __dmul_rn(ad, ad)), dtn));
                theta += fabs(e[l]) * gx.nrm[1 + l];
            }
        }
        theta *= fabs(dtn);
        int s, m;
        kr_plan_fast(a, my_thr, theta, s, m);
        const KrEcon eco = kr_econ_select(ec, theta, s, m);   // economized step polynomial (kr_device.h) where the step's row allows it
        const double my_g = eco.on ? ec : my_rinv;
        const double fs = (s == 1) ? dtn : dtn / (double)s;

        // ---- psi <- exp(-i H(eps_new) dt) psi.  -i H(eps_new) psi = D_0 + sum e_l D_l is known already, so the first
        //      sub-step costs m - 1 products:  psi' = psi + y_1,  y_m = v,  y_{j-1} = v + (fs / j)(-i H y_j),  v = fs (-i H psi)
        double Er[16], Ei[16];   // A operands of H(eps_new): formed ONCE per step (behind the first set's first LDS write), whatever the number of sets
        // chi(t_{n+1}) (HERM: chi(t_{n+2})) from the prefetch registers into LDS (a macro, not a lambda: captured by
        // reference the prefetch registers would become a stack array)
#define KR_MF_HAND_OVER_CHI(st_)                                                                                       \
        if (memwave && n + 1 + HERM < NT) {                                                                            \
            double2 *scw_ = &KR_SC_AT(st_, n + 1 + HERM)[0][0];                                                        \
            _Pragma("unroll") for (int c = 0; c < CH; ++c) {                                                           \
                if (m_ok[c]) {   /* columns re, im of the element: {(-i chi)_c, chi_c} */                             \
                    scw_[m_pos[c]] = make_double2(cpf[c].y, cpf[c].x);                                                 \
                    scw_[m_pos[c] + 1] = make_double2(-cpf[c].x, cpf[c].y);                                            \
                }                                                                                                      \
            }                                                                                                          \
        }
#pragma unroll
        for (int st = 0; st < SETS; ++st) {
            const size_t sb = tile_base + st * set_stride;
            double v[TILES], y[TILES];
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                double w = D[st][0][t];
#pragma unroll
                for (int l = 0; l < LMAX; ++l) w = fma(e[l], D[st][1 + l][t], w);
                v[t] = fs * w;
                y[t] = v[t];
                if (m > 1) kr_mf_put(sy[0][t], doff, zoff, zs, v[t]);
            }
            if (st == 0) {
#pragma unroll
                for (int kb = 0; kb < 16; ++kb) {
                    const double2 h0 = sH0[kb * 256 + tid];
                    double r = h0.x, im = h0.y;
#pragma unroll
                    for (int l = 0; l < LMAX; ++l) { r = fma(e[l], Hr[l][kb], r); im = fma(e[l], Hi[l][kb], im); }
                    Er[kb] = r;
                    Ei[kb] = im;
                }
            }
            // memory wave: start fetching this set's chi(t_{n+1}) (HERM: chi(t_{n+2})); it lands while the set's rounds run
            if (memwave && n + 1 + HERM < NT) {
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    cpf[c] = a.X[(size_t)(n + 1 + HERM) * slab + sb + min(lane + 64 * c, tile_elems - 1)];
            }
            // Rounds j = m .. 2.  Every wave has written its own rows of the current Horner vector; the workgroup barrier of the
            // round sits inside kr_mf_apply<.., OWNFIRST> (behind the requests for the wave's own rows).  No hazard on the
            // ping-pong buffers: a wave writes buffer wb in round j only after that barrier, which the others reach after their
            // reads of the same buffer in round j + 1.
            if (stamp && st == 0) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tB1 += t - t_prev; }
            int wb = 1;
            for (int j = m; j >= 2; --j) {
                const double f = fs * kr_rinv_lane(my_g, j);
                double d[TILES];
                kr_mf_apply<TILES, true>(Er, Ei, &sy[wb ^ 1][0][0], boff, kb0, d);   // (own rows first: 1016 -> 969 cycles per round at one tile, a wash at two)
#pragma unroll
                for (int t = 0; t < TILES; ++t) {
                    y[t] = fma(f, d[t], v[t]);
                    if (j > 2) kr_mf_put(sy[wb][t], doff, zoff, zs, y[t]);
                }
                wb ^= 1;
            }
            if (stamp && st == SETS - 1) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tB2 += t - t_prev; }
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                psi[st][t] = fma(eco.a1, y[t], eco.a0 * psi[st][t]);   // (Taylor: a0 = a1 = 1)
                kr_mf_put(KR_SX_WR(st)[t], doff, zoff, zs, psi[st][t]);   // (several sets: in place -- its last readers were phase A)
            }
            if (s == 1) { KR_MF_HAND_OVER_CHI(st) }
            __syncthreads();   // the set's psi / chi are complete; the Horner ping-pong is free for the next set
            if (SETS == 1) cur ^= 1;
            for (int sub = 1; sub < s; ++sub) {  // further sub-steps (only when ||H dt|| > KR_THETA_MAX): plain Horner from psi
                int w2 = 0;
                for (int j = m; j >= 1; --j) {
                    const double f = fs * kr_rinv_lane(my_rinv, j);
                    double d[TILES];
                    if (j == m) kr_mf_apply<TILES>(Er, Ei, &KR_SX_RD(st)[0][0], boff, kb0, d);
                    else kr_mf_apply<TILES>(Er, Ei, &sy[w2 ^ 1][0][0], boff, kb0, d);
                    if (SETS > 1 && m == 1) __syncthreads();   // in-place psi array: every wave has read it before anyone overwrites
#pragma unroll
                    for (int t = 0; t < TILES; ++t) {
                        const double yn = fma(f, d[t], psi[st][t]);
                        if (j > 1) kr_mf_put(sy[w2][t], doff, zoff, zs, yn);
                        else { psi[st][t] = yn; kr_mf_put(KR_SX_WR(st)[t], doff, zoff, zs, yn); }
                    }
                    if (j == 1 && sub == s - 1) { KR_MF_HAND_OVER_CHI(st) }
                    __syncthreads();
                    w2 ^= 1;
                }
                if (SETS == 1) cur ^= 1;
            }
        }
#undef KR_MF_HAND_OVER_CHI
        if (stamp) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tB += t - t_prev; t_prev = t; }

        // memory wave: forward storage slot n+1 <- psi(t_{n+1}) (coalesced 16-B stores from LDS)
        if (memwave && a.store) {
#pragma unroll
            for (int st = 0; st < SETS; ++st) {
                const double2 *sxr = &KR_SX_RD(st)[0][0];
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    if (m_ok[c]) a.store[(size_t)(n + 1) * slab + tile_base + st * set_stride + lane + 64 * c] = make_double2(sxr[m_pos[c]].y, sxr[m_pos[c] + 1].y);
            }
        }
        if (stamp) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tS += t - t_prev; t_prev = t; }
    }
    if (stamp && tid == 0) {
        a.debug[0] = tA; a.debug[1] = tX; a.debug[2] = tB; a.debug[3] = tS;
        a.debug[4] = kr_realtime() - rt0; a.debug[5] = tB1; a.debug[6] = tB2;
    }
    if (a.fin && row_ok) {
#pragma unroll
        for (int st = 0; st < SETS; ++st)
#pragma unroll
            for (int t = 0; t < TILES; ++t)
                ((double *)a.fin)[(tile_base + st * set_stride + (size_t)(2 * t + dt) * N + drow) * 2 + dc] = psi[st][t];
    }
    if (wg == 0 && tid == 0) {
#pragma unroll
        for (int l = 0; l < LMAX; ++l) if (l < L) a.g_a_int[l] = gint[l];
    }
#undef KR_SX_RD
#undef KR_SX_WR
#undef KR_SC_AT
}

// =============================================================================================
// Known-pulse sweep with precomputed propagators on the matrix cores: psi <- U_n psi, one MFMA round per step (the VALU
// form is kr_sweep_apply, kr_device_prop.h).  packed(U psi) = Ur packed(psi) - Ui packed(-i psi), so with the operand
// pairing of kr_mf_multi (first operand set times packed(-i w), second times packed(w)) the A operands are {-Ui, Ur}.
// kr_expm_mfma64 writes them in exactly the order the lanes read them (KrSweepArgs::u_layout = 1):
//      Umf[n][wave][slot][lane] = {-Ui, Ur} of U_n[16 wave + 4 blk + f0][4 ((4 wave + slot) & 15) + f2]      (64 KB per step)
// so a step's operands are 16 coalesced 16-byte loads per lane, requested one step ahead.
// =============================================================================================
template <int TILES>
__global__ __launch_bounds__(256, 1) void kr_mf_apply_sweep(const KrSweepArgs a) {
    constexpr int KT = 2 * TILES, TW = KR_MF_TW, NW = 4, CH = KT;
    __shared__ __attribute__((aligned(16))) double2 sx[2][TILES][TW];
    const int tid = threadIdx.x, wg = blockIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f0 = lane & 3, blk = (lane >> 2) & 3, f2 = lane >> 4;
    const int drow = 16 * wave + 4 * blk + f2, dt = f0 >> 1, dc = f0 & 1;
    const int boff = f2 * 4 + f0, kb0 = 4 * wave, doff = drow * 4 + f0, zoff = drow * 4 + (f0 ^ 1);
    const double zs = dc ? 1.0 : -1.0;
    const bool memwave = (wave == NW - 1);
    const int N = a.N, NT = a.NT;
    const size_t slab = (size_t)a.Kp * N, tile_base = (size_t)wg * KT * N;
    const int tile_elems = KT * N;
    const bool row_ok = drow < N;
    const double2 *__restrict__ Umf = a.U + (size_t)(a.wg_gen ? a.wg_gen[blockIdx.x / (a.gen_div > 0 ? a.gen_div : 1)] : 0) * NT * 4096;

    int m_pos[CH];
    bool m_ok[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int e = lane + 64 * c;
        m_ok[c] = e < tile_elems;
        const int tq = m_ok[c] ? e / N : 0, i = m_ok[c] ? e % N : 0;
        m_pos[c] = (tq >> 1) * TW + i * 4 + 2 * (tq & 1);
    }
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        double v = 0.0;
        if (row_ok) v = ((const double *)a.init)[(tile_base + (size_t)(2 * t + dt) * N + drow) * 2 + dc];
        kr_mf_put(sx[0][t], doff, zoff, zs, v);
    }
    if (memwave && a.store) {
        const int slot0 = (a.dir > 0) ? 0 : NT;
#pragma unroll
        for (int c = 0; c < CH; ++c)
            if (m_ok[c]) a.store[(size_t)slot0 * slab + tile_base + lane + 64 * c] = a.init[tile_base + lane + 64 * c];
    }
    // A operands: two register sets, each REFILLED at the end of the step that used it (for the step after next), as asm loads
    // with written-out waits (round 3).  With compiler-visible loads the set in use was copied out of the way of its own refill --
    // and a copy of registers with loads in flight is a vmcnt(0): however many steps were "in flight", the loads of step n + 1
    // were only issued once those of step n had arrived (1.9 us per step at two tiles against ~0.5 us of MFMA work; a third set
    // bought 2 %).  vmcnt(16): everything but the youngest set has landed (the memory wave's stores only make that stricter).
    typedef double kr_v2d __attribute__((ext_vector_type(2)));   // (a vector type: asm register operands cannot be structs)
    kr_v2d ua[16], ub[16];
#define KR_MF_LD4(set_, g_, p_)                                                                                        \
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:1024\n\t"                 \
                 "global_load_dwordx4 %2, %4, off offset:2048\n\tglobal_load_dwordx4 %3, %4, off offset:3072"          \
                 : "=&v"(set_[4 * (g_)]), "=&v"(set_[4 * (g_) + 1]), "=&v"(set_[4 * (g_) + 2]), "=&v"(set_[4 * (g_) + 3]) : "v"(p_) : "memory");
#define KR_MF_FILL(set_, step_)                                                                                        \
    {                                                                                                                  \
        const int sc_ = ((step_) < NT) ? (step_) : (NT - 1);   /* past the end: the last step's operands again */         \
        const int nn_ = (a.dir > 0) ? sc_ : (NT - 1 - sc_);                                                            \
        const double2 *U_ = Umf + (size_t)nn_ * 4096 + (size_t)wave * 1024 + lane;                                     \
        const double2 *U1_ = U_ + 256, *U2_ = U_ + 512, *U3_ = U_ + 768;                                               \
        KR_MF_LD4(set_, 0, U_) KR_MF_LD4(set_, 1, U1_) KR_MF_LD4(set_, 2, U2_) KR_MF_LD4(set_, 3, U3_)                 \
    }
#define KR_MF_WAIT(s_)                                                                                                 \
    asm volatile("s_waitcnt vmcnt(16)" : "+v"(s_[0]), "+v"(s_[1]), "+v"(s_[2]), "+v"(s_[3]), "+v"(s_[4]), "+v"(s_[5]), "+v"(s_[6]), "+v"(s_[7]), \
                 "+v"(s_[8]), "+v"(s_[9]), "+v"(s_[10]), "+v"(s_[11]), "+v"(s_[12]), "+v"(s_[13]), "+v"(s_[14]), "+v"(s_[15]) :: "memory");
    KR_MF_FILL(ua, 0)
    KR_MF_FILL(ub, 1)
    __syncthreads();
    int cur = 0;
    double psi[TILES];
#define KR_MF_APPLY_STEP(step_, un_)                                                                                   \
    {                                                                                                                  \
        const int n = (a.dir > 0) ? (step_) : (NT - 1 - (step_));                                                      \
        KR_MF_WAIT(un_)                                                                                                \
        double Ar[16], Ai[16];                                                                                         \
        _Pragma("unroll") for (int sl = 0; sl < 16; ++sl) { Ar[sl] = un_[sl].x; Ai[sl] = un_[sl].y; }                  \
        kr_mf_apply<TILES>(Ar, Ai, &sx[cur][0][0], boff, kb0, psi);                                                    \
        _Pragma("unroll") for (int t = 0; t < TILES; ++t) kr_mf_put(sx[cur ^ 1][t], doff, zoff, zs, psi[t]);           \
        KR_MF_FILL(un_, (step_) + 2)                                                                                   \
        __syncthreads();                                                                                               \
        cur ^= 1;                                                                                                      \
        if (memwave && a.store) {                                                                                      \
            const int slot = (a.dir > 0) ? (n + 1) : n;                                                                \
            _Pragma("unroll") for (int c = 0; c < CH; ++c)                                                             \
                if (m_ok[c]) a.store[(size_t)slot * slab + tile_base + lane + 64 * c] =                                \
                    make_double2((&sx[cur][0][0])[m_pos[c]].y, (&sx[cur][0][0])[m_pos[c] + 1].y);                      \
        }                                                                                                              \
    }
    for (int step = 0; step < NT; step += 2) {
        KR_MF_APPLY_STEP(step, ua)
        if (step + 1 < NT) KR_MF_APPLY_STEP(step + 1, ub)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the last fills: nothing reads them)
#undef KR_MF_APPLY_STEP
#undef KR_MF_FILL
#undef KR_MF_LD4
#undef KR_MF_WAIT
    if (a.fin && memwave) {
#pragma unroll
        for (int c = 0; c < CH; ++c)
            if (m_ok[c]) a.fin[tile_base + lane + 64 * c] = make_double2((&sx[cur][0][0])[m_pos[c]].y, (&sx[cur][0][0])[m_pos[c] + 1].y);
    }
}

// kind = 1 | SCHED << 4 | CAL << 5 | HERM << 6 (as kr_launch_np); tiles = trajectories per workgroup / 2.
// One translation unit per LMAX (kr_mf64_l2.hip, kr_mf64_l4.hip).
template <int LM>
static hipError_t kr_launch_mfma64(int tiles, int kind, int grid, hipStream_t s, const KrSweepArgs *a) {
    if (kind == 3) {   // known-pulse sweep on precomputed propagators (LM-independent: instantiated in the LM = 2 unit only)
        if constexpr (LM == 2) {
            if (tiles == 1) { kr_launch_kernel((kr_mf_apply_sweep<1>), grid, 256, 0, s, *a); return hipGetLastError(); }
            if (tiles == 2) { kr_launch_kernel((kr_mf_apply_sweep<2>), grid, 256, 0, s, *a); return hipGetLastError(); }
        }
        return hipErrorInvalidValue;
    }
    if ((kind & 15) != 1) return hipErrorInvalidValue;
    const int v = ((kind >> 4) & 1) | (((kind >> 6) & 1) << 1) | (((kind >> 5) & 1) << 2);   // SCHED | HERM << 1 | CAL << 2
#define KR_CASE_M(TL)                                                                                              \
    if (tiles == TL) {                                                                                             \
        switch (v) {                                                                                               \
        case 0: kr_launch_kernel((kr_mf_update<LM, TL, 0, 0, 0>), grid, 256, 0, s, *a); break;       \
        case 1: kr_launch_kernel((kr_mf_update<LM, TL, 1, 0, 0>), grid, 256, 0, s, *a); break;       \
        case 2: kr_launch_kernel((kr_mf_update<LM, TL, 0, 1, 0>), grid, 256, 0, s, *a); break;       \
        case 3: kr_launch_kernel((kr_mf_update<LM, TL, 1, 1, 0>), grid, 256, 0, s, *a); break;       \
        case 4: kr_launch_kernel((kr_mf_update<LM, TL, 0, 0, 1>), grid, 256, 0, s, *a); break;       \
        case 5: kr_launch_kernel((kr_mf_update<LM, TL, 1, 0, 1>), grid, 256, 0, s, *a); break;       \
        case 6: kr_launch_kernel((kr_mf_update<LM, TL, 0, 1, 1>), grid, 256, 0, s, *a); break;       \
        default: kr_launch_kernel((kr_mf_update<LM, TL, 1, 1, 1>), grid, 256, 0, s, *a); break;      \
        }                                                                                                          \
        return hipGetLastError();                                                                                  \
    }
    // non-linear controls (kind bit 256): both schedules, both variants, one set
#define KR_CASE_P(TL)                                                                                              \
    if ((kind & 256) && tiles == TL) {                                                                             \
        switch (v & 3) {                                                                                           \
        case 0: kr_launch_kernel((kr_mf_update<LM, TL, 0, 0, 0, 1, 1>), grid, 256, 0, s, *a); break;     \
        case 1: kr_launch_kernel((kr_mf_update<LM, TL, 1, 0, 0, 1, 1>), grid, 256, 0, s, *a); break;     \
        case 2: kr_launch_kernel((kr_mf_update<LM, TL, 0, 1, 0, 1, 1>), grid, 256, 0, s, *a); break;     \
        default: kr_launch_kernel((kr_mf_update<LM, TL, 1, 1, 0, 1, 1>), grid, 256, 0, s, *a); break;    \
        }                                                                                                          \
        return hipGetLastError();                                                                                  \
    }
    // several sets per workgroup (tiles argument = tiles | sets << 8): two tiles, general variant only
#define KR_CASE_S(ST)                                                                                              \
    if (tiles == (2 | (ST << 8))) {                                                                                \
        switch (v & 5) {                                                                                           \
        case 0: kr_launch_kernel((kr_mf_update<LM, 2, 0, 0, 0, ST>), grid, 256, 0, s, *a); break;    \
        case 1: kr_launch_kernel((kr_mf_update<LM, 2, 1, 0, 0, ST>), grid, 256, 0, s, *a); break;    \
        case 4: kr_launch_kernel((kr_mf_update<LM, 2, 0, 0, 1, ST>), grid, 256, 0, s, *a); break;    \
        default: kr_launch_kernel((kr_mf_update<LM, 2, 1, 0, 1, ST>), grid, 256, 0, s, *a); break;   \
        }                                                                                                          \
        return hipGetLastError();                                                                                  \
    }
#ifdef KR_MF_ONLY_TILES
    KR_CASE_M(KR_MF_ONLY_TILES)
#else
    KR_CASE_P(1) KR_CASE_P(2)
    KR_CASE_M(1) KR_CASE_M(2)   // (4 tiles: the vector arrays + H0 exceed the CU's LDS and the accumulators spill)
    KR_CASE_S(2) KR_CASE_S(4)
#endif
#undef KR_CASE_M
#undef KR_CASE_P
#undef KR_CASE_S
    return hipErrorInvalidValue;
}
