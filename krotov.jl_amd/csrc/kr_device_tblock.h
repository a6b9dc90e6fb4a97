// Known-pulse sweeps in BLOCKS OF TIME (round 5; NP = 64).
//
// The sequential apply sweep (kr_sweep_apply / kr_mf_apply_sweep) spreads the TRAJECTORIES over the CUs, so every CU streams every step
// propagator U_n (64 KB) out of L2 in every step and the N_T steps are a chain of dependent latencies: 0.75 ms per sweep at C3 against
// 0.11 ms of arithmetic.  With known pulses (optimize.jl:247-265 initial forward sweep, :303-317 backward sweep through the adjoint
// generator, workspace.jl:69,149-161) nothing but the state itself couples the steps, so TIME can be spread over the CUs instead:
//
//   (i)   step propagators W_s, s = 0 .. N_T - 1 in sweep order (kr_expm_mfma64, as before; written as MFMA A fragments, u_layout 2);
//   (ii)  block propagators P_b = W_(bS+S-1) ... W_(bS) for blocks of S steps            kr_tb_products, one workgroup per block;
//   (iii) the states at the block boundaries, y_((b+1)S) = P_b y_(bS): the sequential apply sweep over the N_T / S block propagators
//         (the existing kernels, a.NT = number of blocks, storing into the boundary buffer);
//   (iv)  every block in parallel from its boundary state: S steps of a 64 x 64 by 64 x 64-trajectory product, U read ONCE per step and
//         workgroup                                                                     kr_tb_apply, this file.
//
// Storage slots as the sequential sweeps fill them: sweep step s (state AFTER s steps) goes to slot s (forward, optimize.jl:254,263) or
// N_T - s (backward, optimize.jl:307,315).  Results differ from the sequential order by rounding only (products of S matrices of norm ~1).
//
// Matrix-core mapping of (iv).  v_mfma_f64_16x16x4_f64: A 16 x 4 (lane l holds A[l & 15][l >> 4]), B 4 x 16 (lane l holds B[l >> 4][l & 15]),
// D 16 x 16 in four registers (register r of lane l is D[(l >> 4) + 4 r][l & 15]).  Register r of the D tile of rows 16 w' .. 16 w' + 15
// therefore holds rows 4 (4 w' + r) + (l >> 4) -- EXACTLY the B operand of k-block t = 4 w' + r.  A wave that computes all four row tiles
// of its 16 columns holds the new state as the next step's B operands: the state never leaves the registers, the waves of a workgroup
// share nothing but the LDS copy of U_s and meet once per step.  Columns = trajectories; real and imaginary parts are separate operand
// sets (y_r = U_r x_r - U_i x_i, y_i = (U_r + U_i)(x_r + x_i) - U_r x_r - U_i x_i: three real products per k-block, three independent
// accumulator chains).  A wave carries 16 trajectories, a workgroup 64; U_s (64 KB as {U_r, U_i} fragment pairs) is copied global -> LDS
// by LDS-DMA one step ahead (two buffers); per step and wave 192 MFMAs (12.3 k cycles at the fp64 matrix rate) against 64 ds_read_b128.
#pragma once
#include "kr_common.h"

typedef double kr_tb_d4 __attribute__((ext_vector_type(4)));
#define KR_TB_YS 66   // row stride (doubles) of the LDS planes of kr_tb_products (as KR_YS of kr_expm_mfma64)

// U in fragment order (u_layout 2): Ufr[(16 w' + t) * 64 + lane] = {U_r, U_i} of U[16 w' + (lane & 15)][4 t + (lane >> 4)]
__device__ __forceinline__ size_t kr_tb_frag_index(const int wt, const int t, const int lane) { return ((size_t)(16 * wt + t)) * 64 + lane; }

// ---- (ii) block propagators ----------------------------------------------------------------------------------------
// One workgroup per (generator, block): P = W_(s1-1) ... W_(s0), left factors as register fragments straight from the fragment-ordered
// U (coalesced 1-KB loads), the running product in LDS planes (re / im, ping-pong) read as B operands -- the product routine of
// kr_expm_mfma64.  Output in the operand order of the sequential apply kernel that consumes it (a.u_layout: 0 kr_sweep_apply,
// 1 kr_mf_apply_sweep), at a.tb_bound-independent address Pout[(gen * NB + b) * 4096].
template <int DUMMY>
__global__ __launch_bounds__(256, 1) void kr_tb_products(const KrSweepArgs a, double2 *__restrict__ Pout) {
    extern __shared__ __attribute__((aligned(16))) double kr_tbp_lds[];
#define KR_TBR(b) (kr_tbp_lds + (size_t)(b) * 2 * 64 * KR_TB_YS)
#define KR_TBI(b) (kr_tbp_lds + (size_t)(b) * 2 * 64 * KR_TB_YS + 64 * KR_TB_YS)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int NT = a.tb_NT, S = a.tb_S, NB = (NT + S - 1) / S;
    const int gen = blockIdx.x / NB, b = blockIdx.x % NB;
    const int s0 = b * S, s1 = min(s0 + S, NT);
    const double2 *Ug = a.U + (size_t)gen * NT * 4096;
    const int row = 16 * w + li;
    auto frag_of = [&](const int s) { return Ug + (size_t)((a.dir > 0) ? s : (NT - 1 - s)) * 4096 + (size_t)(16 * w) * 64 + lane; };
    // W_(s0) -> buffer 0 at its fragment positions
    {
        const double2 *F = frag_of(s0);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const double2 u = F[(size_t)t * 64];
            KR_TBR(0)[row * KR_TB_YS + 4 * t + lk] = u.x;
            KR_TBI(0)[row * KR_TB_YS + 4 * t + lk] = u.y;
        }
    }
    __syncthreads();
    int cur = 0;
    for (int s = s0 + 1; s < s1; ++s) {
        // complex product in THREE real ones (round 5):  re = Ar Br - Ai Bi,  im = (Ar + Ai)(Br + Bi) - Ar Br - Ai Bi.  The sums of the
        // operands are vector adds in the shadow of the matrix cores; a quarter of the MFMAs -- which bound these kernels -- is saved,
        // at the price of one more rounding of the imaginary part's operands (errors stay ~1e-16 of the matrices' size).
        double Ar[16], Ai[16], As[16];
        const double2 *F = frag_of(s);
#pragma unroll
        for (int t = 0; t < 16; ++t) { const double2 u = F[(size_t)t * 64]; Ar[t] = u.x; Ai[t] = u.y; As[t] = u.x + u.y; }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            kr_tb_d4 pa = {0, 0, 0, 0}, pb = {0, 0, 0, 0}, pc = {0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const double br = KR_TBR(cur)[(4 * t + lk) * KR_TB_YS + 16 * ct + li];
                const double bi = KR_TBI(cur)[(4 * t + lk) * KR_TB_YS + 16 * ct + li];
                const double bs = br + bi;
                pa = __builtin_amdgcn_mfma_f64_16x16x4f64(Ar[t], br, pa, 0, 0, 0);
                pb = __builtin_amdgcn_mfma_f64_16x16x4f64(Ai[t], bi, pb, 0, 0, 0);
                pc = __builtin_amdgcn_mfma_f64_16x16x4f64(As[t], bs, pc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int orow = 16 * w + lk + 4 * r, ocol = 16 * ct + li;
                KR_TBR(cur ^ 1)[orow * KR_TB_YS + ocol] = pa[r] - pb[r];
                KR_TBI(cur ^ 1)[orow * KR_TB_YS + ocol] = (pc[r] - pa[r]) - pb[r];
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    double2 *Pn = Pout + ((size_t)gen * NB + b) * 4096;
    if (a.u_layout == 1) {   // MFMA A operands of kr_mf_apply_sweep (kr_device_mfma.h): {-Ui, Ur}, lane order (f0, blk, f2)
        for (int e = tid; e < 4096; e += 256) {
            const int l = e & 63, sl = (e >> 6) & 15, wv = e >> 10;
            const int r = 16 * wv + 4 * ((l >> 2) & 3) + (l & 3), cc = 4 * ((4 * wv + sl) & 15) + (l >> 4);
            Pn[e] = make_double2(-KR_TBI(cur)[r * KR_TB_YS + cc], KR_TBR(cur)[r * KR_TB_YS + cc]);
        }
        return;
    }
    // thread-major pack of kr_sweep_apply: Upack[jj * RT + i * JQ + jq] = U[i][jq * JS + jj]
    constexpr int JQ_A = KR_APPLY_JQ64, JS_A = 64 / JQ_A, RT_A = 64 * JQ_A;
    for (int t2 = tid; t2 < RT_A; t2 += 256) {
        const int i = t2 / JQ_A, jq = t2 % JQ_A;
#pragma unroll
        for (int jj = 0; jj < JS_A; ++jj) {
            const int c = jq * JS_A + jj;
            Pn[(size_t)jj * RT_A + t2] = make_double2(KR_TBR(cur)[i * KR_TB_YS + c], KR_TBI(cur)[i * KR_TB_YS + c]);
        }
    }
#undef KR_TBR
#undef KR_TBI
}

// ---- (ii) as a tree of pairwise products (S a power of two) ------------------------------------------------------------------
// One level: out[j] = in[2 j + 1] in[2 j] (sweep order: the later factor on the left), both in fragment order; a last matrix without a
// partner is copied.  grid = generators x ceil(n_in / 2).  log2(S) dependent launches of ~8 us replace the chain of S - 1 products
// inside one workgroup (which also leaves half the chip idle at S = 8): S = 16 costs what S = 8 cost, and the sequential sweep (iii)
// over the block boundaries -- a chain of latencies -- halves.  `final`: the level's output feeds the sequential apply kernel and is
// written in ITS operand order (a.u_layout 0 / 1); otherwise fragment order for the next level.
template <int DUMMY>
__global__ __launch_bounds__(256, 1) void kr_tb_pair(const KrSweepArgs a, const double2 *__restrict__ in, double2 *__restrict__ out, const int n_in, const int level0,
                                                     const int final) {
    extern __shared__ __attribute__((aligned(16))) double kr_tbq_lds[];   // planes [re | im][64][KR_TB_YS] of the right factor, then of the product
#define KR_TQR (kr_tbq_lds)
#define KR_TQI (kr_tbq_lds + 64 * KR_TB_YS)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int n_out = (n_in + 1) / 2;
    const int gen = blockIdx.x / n_out, j = blockIdx.x % n_out;
    const int NT = a.tb_NT;
    const int row = 16 * w + li;
    // level 0 reads the step propagators in SWEEP order (backward sweeps run down the grid); later levels read what this kernel wrote
    auto mat = [&](const int idx) {
        const int n = level0 ? ((a.dir > 0) ? idx : (NT - 1 - idx)) : idx;
        return in + ((size_t)gen * n_in + n) * 4096;
    };
    const bool paired = (2 * j + 1) < n_in;
    {   // right factor (or the lone matrix) -> planes
        const double2 *F = mat(2 * j) + (size_t)(16 * w) * 64 + lane;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const double2 u = F[(size_t)t * 64];
            KR_TQR[row * KR_TB_YS + 4 * t + lk] = u.x;
            KR_TQI[row * KR_TB_YS + 4 * t + lk] = u.y;
        }
    }
    __syncthreads();
    if (paired) {
        double Ar[16], Ai[16], As[16];
        const double2 *F = mat(2 * j + 1) + (size_t)(16 * w) * 64 + lane;
#pragma unroll
        for (int t = 0; t < 16; ++t) { const double2 u = F[(size_t)t * 64]; Ar[t] = u.x; Ai[t] = u.y; As[t] = u.x + u.y; }
        kr_tb_d4 dr[4], di[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            kr_tb_d4 pa = {0, 0, 0, 0}, pb = {0, 0, 0, 0}, pc = {0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const double br = KR_TQR[(4 * t + lk) * KR_TB_YS + 16 * ct + li];
                const double bi = KR_TQI[(4 * t + lk) * KR_TB_YS + 16 * ct + li];
                const double bs = br + bi;
                pa = __builtin_amdgcn_mfma_f64_16x16x4f64(Ar[t], br, pa, 0, 0, 0);
                pb = __builtin_amdgcn_mfma_f64_16x16x4f64(Ai[t], bi, pb, 0, 0, 0);
                pc = __builtin_amdgcn_mfma_f64_16x16x4f64(As[t], bs, pc, 0, 0, 0);
            }
            dr[ct] = pa - pb;
            di[ct] = (pc - pa) - pb;
        }
        __syncthreads();   // every wave has read the right factor: the planes take the product (D layout positions)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int orow = 16 * w + lk + 4 * r, ocol = 16 * ct + li;
                KR_TQR[orow * KR_TB_YS + ocol] = dr[ct][r];
                KR_TQI[orow * KR_TB_YS + ocol] = di[ct][r];
            }
        __syncthreads();
    }
    double2 *Pn = out + ((size_t)gen * n_out + j) * 4096;
    if (!final) {   // fragment order: [(16 w' + t) * 64 + lane] = {re, im} of M[16 w' + (lane & 15)][4 t + (lane >> 4)]
        for (int e = tid; e < 4096; e += 256) {
            const int l = e & 63, t = (e >> 6) & 15, wt = e >> 10;
            const int r = 16 * wt + (l & 15), cc = 4 * t + (l >> 4);
            Pn[e] = make_double2(KR_TQR[r * KR_TB_YS + cc], KR_TQI[r * KR_TB_YS + cc]);
        }
    } else if (a.u_layout == 1) {   // MFMA A operands of kr_mf_apply_sweep: {-Ui, Ur}, lane order (f0, blk, f2)
        for (int e = tid; e < 4096; e += 256) {
            const int l = e & 63, sl = (e >> 6) & 15, wv = e >> 10;
            const int r = 16 * wv + 4 * ((l >> 2) & 3) + (l & 3), cc = 4 * ((4 * wv + sl) & 15) + (l >> 4);
            Pn[e] = make_double2(-KR_TQI[r * KR_TB_YS + cc], KR_TQR[r * KR_TB_YS + cc]);
        }
    } else {                        // thread-major pack of kr_sweep_apply
        constexpr int JQ_A = KR_APPLY_JQ64, JS_A = 64 / JQ_A, RT_A = 64 * JQ_A;
        for (int t2 = tid; t2 < RT_A; t2 += 256) {
            const int i = t2 / JQ_A, jq = t2 % JQ_A;
#pragma unroll
            for (int jj = 0; jj < JS_A; ++jj) {
                const int c = jq * JS_A + jj;
                Pn[(size_t)jj * RT_A + t2] = make_double2(KR_TQR[i * KR_TB_YS + c], KR_TQI[i * KR_TB_YS + c]);
            }
        }
    }
#undef KR_TQR
#undef KR_TQI
}

// ---- (iv) all blocks in parallel --------------------------------------------------------------------------------------
// grid = NB x items; workgroup (b, item) carries trajectories [k0, k0 + cnt) (cnt <= 64, one generator) through the steps of block b.
template <int DUMMY>
__global__ __launch_bounds__(256, 1) void kr_tb_apply(const KrSweepArgs a) {
    extern __shared__ __attribute__((aligned(16))) double2 kr_tba_lds[];   // [2][4096] fragment-ordered U of two steps
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int N = a.N, NT = a.tb_NT, S = a.tb_S, NB = (NT + S - 1) / S;
    const int b = blockIdx.x / a.tb_nitems, it = blockIdx.x % a.tb_nitems;
    const int k0 = a.tb_items[4 * it], cnt = a.tb_items[4 * it + 1], gen = a.tb_items[4 * it + 2];
    const int s0 = b * S, s1 = min(s0 + S, NT);
    // the last block's last step would reproduce the final boundary state, which the sequential sweep over the block propagators has
    // delivered already (and which the caller sees as the sweep's final states): that one is stored instead, bit for bit
    const int s_end = (b == NB - 1) ? (s1 - 1) : s1;
    const size_t slab = (size_t)a.Kp * N;
    const double2 *Ug = a.U + (size_t)gen * NT * 4096;
    const int kk = k0 + 16 * w + li;
    const bool kok = (16 * w + li) < cnt;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)kr_tba_lds;

    // state as B operands: x[t] = component 4 t + lk of trajectory kk
    double xr[16], xi[16];
    {
        const double2 *src = a.tb_bound + (size_t)b * slab + (size_t)kk * N;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            double2 v = make_double2(0.0, 0.0);
            if (kok && 4 * t + lk < N) v = src[4 * t + lk];
            xr[t] = v.x; xi[t] = v.y;
        }
    }
    auto slot_of = [&](const int j) { return (a.dir > 0) ? j : (NT - j); };
    auto store_state = [&](const int j) {   // state after j sweep steps
        double2 *dst = a.store + (size_t)slot_of(j) * slab + (size_t)kk * N;
#pragma unroll
        for (int t = 0; t < 16; ++t)
            if (kok && 4 * t + lk < N) dst[4 * t + lk] = make_double2(xr[t], xi[t]);
    };
    // LDS-DMA of one step's U (64 KB): wave w copies the sixteen 1-KB pieces of row tile w (global_load_lds_dwordx4: lane i lands at
    // M0 + 16 i; scalar base + lane offset)
    auto issue_u = [&](const int buf, const int s) {
        const int n = (a.dir > 0) ? s : (NT - 1 - s);
        const double2 *base = Ug + (size_t)n * 4096 + (size_t)(16 * w) * 64;
        const unsigned voff = (unsigned)lane * 16u;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const unsigned long long *sb = kr_uniform_ptr((const unsigned long long *)(base + (size_t)t * 64));
            const unsigned ld = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)buf * 65536u + (unsigned)(16 * w + t) * 1024u));
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(sb), "s"(ld) : "memory");
        }
    };
    if (s0 < s_end) issue_u(0, s0);
    for (int s = s0; s < s_end; ++s) {
        const int buf = (s - s0) & 1;
        // U_s has landed (this wave's pieces: vmcnt; everybody's: the barrier), and every wave is done with the other buffer.  The only
        // other vector-memory operations in flight are the stores of the state below, issued a whole step (~7 us) ago.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + 1 < s_end) issue_u(buf ^ 1, s + 1);
        // the state after s steps goes to its slot while the next one is computed (block b > 0: its first state is block b - 1's last)
        if (a.store && (s > s0 || b == 0)) store_state(s);
        const double2 *Ul = kr_tba_lds + (size_t)buf * 4096 + lane;
        double yr[16], yi[16];
        // 64 operand pairs {U_r, U_i} (one ds_read_b128 each) feed 256 MFMAs through a window of eight registers pairs: pair q + 8 is
        // requested right behind the MFMAs of pair q, i.e. every read rides on 8 x 256 cycles of matrix-core work (left to itself hipcc
        // puts each read directly in front of its MFMAs and waits for it; sched_group_barrier did not move it either)
        double2 win[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) win[q] = Ul[(size_t)q * 64];
        // y = U x in THREE real products per k-block (see kr_tb_products): y_r = U_r x_r - U_i x_i, y_i = (U_r + U_i)(x_r + x_i) - U_r x_r - U_i x_i
        double xs[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) xs[t] = xr[t] + xi[t];
        kr_tb_d4 pa = {0, 0, 0, 0}, pb = {0, 0, 0, 0}, pc = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 64; ++q) {
            const int t = q & 15, wt = q >> 4;
            const double2 u = win[q & 7];
            const double us = u.x + u.y;
            __builtin_amdgcn_sched_barrier(0);
            pa = __builtin_amdgcn_mfma_f64_16x16x4f64(u.x, xr[t], pa, 0, 0, 0);
            pb = __builtin_amdgcn_mfma_f64_16x16x4f64(u.y, xi[t], pb, 0, 0, 0);
            pc = __builtin_amdgcn_mfma_f64_16x16x4f64(us, xs[t], pc, 0, 0, 0);
            if (q + 8 < 64) win[q & 7] = Ul[(size_t)(q + 8) * 64];
            __builtin_amdgcn_sched_barrier(0);
            if (t == 15) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { yr[4 * wt + r] = pa[r] - pb[r]; yi[4 * wt + r] = (pc[r] - pa[r]) - pb[r]; }
                pa = kr_tb_d4{0, 0, 0, 0}; pb = kr_tb_d4{0, 0, 0, 0}; pc = kr_tb_d4{0, 0, 0, 0};
            }
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) { xr[t] = yr[t]; xi[t] = yi[t]; }
    }
    if (a.store && s_end > s0) store_state(s_end);
    else if (a.store && b == 0 && s_end == s0) store_state(0);
    if (a.store && b == NB - 1 && kok) {   // final states of the sweep: the sequential sweep's, bit for bit
        const double2 *src = a.tb_bound + (size_t)NB * slab + (size_t)kk * N;
        double2 *dst = a.store + (size_t)slot_of(NT) * slab + (size_t)kk * N;
#pragma unroll
        for (int t = 0; t < 16; ++t)
            if (4 * t + lk < N) dst[4 * t + lk] = src[4 * t + lk];
    }
}

// one level of the tree of pairwise products: n_in matrices per generator in `in` -> ceil(n_in / 2) in `out`
static inline hipError_t kr_launch_tbpair_impl(int generators, int n_in, int level0, int final, hipStream_t s, const KrSweepArgs *a, const double2 *in, double2 *out) {
    constexpr size_t lds = (size_t)2 * 64 * KR_TB_YS * sizeof(double);
    hipError_t e = kr_raise_dynamic_lds((const void *)kr_tb_pair<0>, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kr_tb_pair<0>, dim3(generators * ((n_in + 1) / 2)), dim3(256), lds, s, *a, in, out, n_in, level0, final);
    return hipGetLastError();
}

// kind: 0 block propagators (lds: two plane pairs), 1 parallel apply (lds: two U buffers)
static inline hipError_t kr_launch_tblock_impl(int kind, int grid, hipStream_t s, const KrSweepArgs *a, double2 *Pout) {
    if (kind == 0) {
        constexpr size_t lds = (size_t)4 * 64 * KR_TB_YS * sizeof(double);
        hipError_t e = kr_raise_dynamic_lds((const void *)kr_tb_products<0>, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kr_tb_products<0>, dim3(grid), dim3(256), lds, s, *a, Pout);
    } else {
        constexpr size_t lds = (size_t)2 * 4096 * sizeof(double2);
        hipError_t e = kr_raise_dynamic_lds((const void *)kr_tb_apply<0>, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kr_tb_apply<0>, dim3(grid), dim3(256), lds, s, *a);
    }
    return hipGetLastError();
}
