// Row-split sweeps, PAIR mode (round 3): R workgroups share TWO trajectories and interleave them.
//
// Why.  In the single mode of kr_device_rowsplit.h a Horner round of a trajectory is  row-dot -> all-gather -> row-dot -> ...,
// strictly serial, and at C5 (N = 256, K = 64: R = 4 workgroups of NR = 64 rows) a round costs 2.9 k cycles of row-dot plus
// 3.5 k cycles of all-gather -- a memory round trip during which the whole group idles -- and every step streams the 256-KB
// control block twice, because the H(eps) block alone fills the registers (profiles/r02_bench_c5_summary.txt: 61.7 ms per
// update sweep, HBM traffic 12.6 x the algorithmic bytes).  With K <= 2 * 32 the same 256 workgroups can instead be cut
// into groups of R = 8 that own NR = 32 rows of TWO trajectories:
//   * while trajectory A's new elements travel through the XCD's L2, the group computes trajectory B's row-dot: the all-gather
//     latency leaves the critical path (one publish and one collect share each pair of barriers);
//   * the blocks shrink to 32 x N: H(eps) AND the control block H_1 stay in registers (2 x 64 VGPRs) for the whole sweep --
//     nothing is streamed per step (the drift block is read only when H(eps) is rebuilt, every KR_RS_REFRESH steps), and the
//     H(eps) block serves both trajectories;
//   * the LDS holds no part of the block any more (the row-dot reads x only: 16 instead of 48 ds_read_b128 per thread and round).
// Column map (differs from the single mode): thread (il, jq) holds the STRIDED columns j = jj * JQ + jq, so the JQ lanes of a row read JQ
// consecutive elements of x per instruction -- conflict-free without padding -- and the state vectors live in LDS as plain arrays x[e].
// That lets the early collect pass (below) land by LDS-DMA straight in the destination vector (1 KB = 64 consecutive elements per
// instruction), with nothing to copy afterwards.  The host packs the blocks accordingly (pack_generator_rowsplit, strided).
// One control only (LMAX = 1: with two controls the three blocks do not fit the registers; those problems keep the single mode).
//
// Reference semantics: exactly kr_rowsplit_sweep's, i.e. /root/reference/src/optimize.jl:247-265, 303-317 (known-pulse sweeps,
// true adjoint for the backward one: workspace.jl:69,149-161) and :321-370 (update sweep).
#pragma once
#include "kr_device_rowsplit.h"
#ifndef KR_PAIR_PRE
#define KR_PAIR_PRE 1   // wave 0 reads the landed vector's phase bits in front of the phase's first barrier (0: behind it)
#endif
#ifndef KR_PAIR_NOVERIFY
#define KR_PAIR_NOVERIFY 0   // 1: TIMING EXPERIMENT ONLY -- trust the early pass without validating it (wrong results when it missed)
#endif
#ifndef KR_PAIR_EARLY_AT
#define KR_PAIR_EARLY_AT 1   // where in the row-dot: 0 = in front of the first chunk of columns, 1 = behind it
#endif
#ifndef KR_PAIR_EARLY
#define KR_PAIR_EARLY 1   // issue the collect's poll pass from inside the row-dot that runs in between (0: poll in the phase only)
#endif

// The two halves of kr_group_gather, so that the publish of one trajectory and the collect of the other share a barrier pair.
//   deposit : the row owners put their new element (LSB = phase) into the LDS vector          -- before barrier 1
//   publish : wave 1 stores the workgroup's block into the group mailbox                      -- between barrier 1 and 2
//   collect : wave 0 polls the foreign elements of a vector published EARLIER into LDS        -- between barrier 1 and 2
struct KrPairBox {
    unsigned long long *base;  // this trajectory's mailbox: [2][NP] elements of 16 B
    unsigned counter;          // all-gathers so far (buffer = counter & 1, phase bit from counter)
    bool local;                // all R workgroups of the group run on ONE XCD: publish into that XCD's L2 only
};
template <int NP, int NR, int JS>
__device__ __forceinline__ void kr_pair_deposit(KrPairBox &box, const int rb, const bool owner, const int il, const double vr,
                                                const double vi, double2 *dst) {
    ++box.counter;
    const unsigned phase = (((box.counter - 1u) >> 1) & 1u) ^ 1u;
    if (owner) {
        const int e = rb * NR + il;
        dst[e] = make_double2(__longlong_as_double((long long)kr_slot_bits(vr, phase)), __longlong_as_double((long long)kr_slot_bits(vi, phase)));
    }
}
template <int NP, int NR, int JS>
__device__ __forceinline__ void kr_pair_publish(const KrPairBox &box, const int rb, const double2 *dst) {
    const int tid = threadIdx.x;
    if ((tid >> 6) == 1 && (tid & 63) < NR) {
        const int par = box.counter & 1;
        const unsigned long long *mb = box.base + (size_t)par * NP * 4;
        const int e = rb * NR + (tid & 63);
        const double2 v = dst[e];
        const unsigned long long br = (unsigned long long)__double_as_longlong(v.x), bi = (unsigned long long)__double_as_longlong(v.y);
        kr_v4u w;
        w.x = (unsigned)br; w.y = (unsigned)(br >> 32); w.z = (unsigned)bi; w.w = (unsigned)(bi >> 32);
        const unsigned long long *pe = mb + 2 * e;
        if (box.local) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(pe), "v"(w) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(pe), "v"(w) : "memory");
    }
}
// deposit AND publish by the row owners themselves, from registers (one barrier per phase instead of two: KR_PAIR_ONEBAR)
template <int NP, int NR, int JS>
__device__ __forceinline__ void kr_pair_deposit_publish(KrPairBox &box, const int rb, const bool owner, const int il, const double vr,
                                                        const double vi, double2 *dst) {
    ++box.counter;
    const unsigned phase = (((box.counter - 1u) >> 1) & 1u) ^ 1u;
    if (owner) {
        const int e = rb * NR + il;
        const unsigned long long br = kr_slot_bits(vr, phase), bi = kr_slot_bits(vi, phase);
        dst[e] = make_double2(__longlong_as_double((long long)br), __longlong_as_double((long long)bi));
        kr_v4u w;
        w.x = (unsigned)br; w.y = (unsigned)(br >> 32); w.z = (unsigned)bi; w.w = (unsigned)(bi >> 32);
        const unsigned long long *pe = box.base + (size_t)(box.counter & 1) * NP * 4 + 2 * e;
        if (box.local) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(pe), "v"(w) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(pe), "v"(w) : "memory");
    }
}
// wave 0 only; `counter` / parity / phase are those of the box's LAST deposit.  Returns false when the bounded spin gave up.
template <int NP, int NR, int JS>
__device__ __forceinline__ bool kr_pair_collect(const KrSweepArgs &a, const KrPairBox &box, const int rb, double2 *dst,
                                                unsigned long long &dbg_tpoll, unsigned long long &dbg_npass) {
    const int lane = threadIdx.x & 63;
    const int par = box.counter & 1;
    const unsigned phase = (((box.counter - 1u) >> 1) & 1u) ^ 1u;
    const unsigned long long *mbc = box.base + (size_t)par * NP * 4;
    constexpr int EC = (NP - NR + 63) / 64;
    static_assert(EC <= KR_POLL_CHUNKS, "at most 256 foreign elements");
    int el[KR_POLL_CHUNKS];
#pragma unroll
    for (int c = 0; c < KR_POLL_CHUNKS; ++c) {
        int q = min(lane + 64 * c, NP - NR - 1);
        if (q >= rb * NR) q += NR;
        el[c] = q;
    }
    kr_v4u g[KR_POLL_CHUNKS];
    const unsigned long long t0 = kr_realtime();
    const unsigned long long tm0 = a.debug ? __builtin_amdgcn_s_memtime() : 0;
    unsigned spins = 0;
    bool alive = true;
    for (;;) {
        bool ok = true;
        kr_poll_elems(mbc + 2 * el[0], mbc + 2 * el[1], mbc + 2 * el[2], mbc + 2 * el[3], EC, g);
#pragma unroll
        for (int c = 0; c < EC; ++c) ok = ok & ((g[c].x & 1u) == phase) & ((g[c].z & 1u) == phase);
        if (__all(ok)) break;
        if ((++spins & 63u) == 0u) {
            const unsigned st = __hip_atomic_load((kr_gu32 *)a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (st != 0u || kr_realtime() - t0 > a.spin_ticks) {
                if (lane == 0 && st == 0u) {
                    __hip_atomic_store((kr_gu32 *)a.status + 1, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store((kr_gu32 *)a.status + 2, box.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store((kr_gu32 *)a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                alive = false;
                break;
            }
        }
    }
    if (a.debug) { dbg_tpoll += __builtin_amdgcn_s_memtime() - tm0; dbg_npass += spins + 1; }
    if (alive) {
#pragma unroll
        for (int c = 0; c < EC; ++c) {
            if (lane + 64 * c < NP - NR) {
                const double re = __longlong_as_double((long long)(((unsigned long long)g[c].y << 32) | g[c].x));
                const double im = __longlong_as_double((long long)(((unsigned long long)g[c].w << 32) | g[c].z));
                dst[el[c]] = make_double2(re, im);
            }
        }
    }
    return alive;
}

// EARLY collect (wave 0): the poll pass of a pending vector costs one L2 round trip (~700 cycles) even when every element has
// long arrived.  So the pass is ISSUED from the middle of the row-dot that runs in between -- as LDS-DMA (global_load_lds_dwordx4:
// no result registers live across the rest of the row-dot; 1 KB per instruction, scalar base + one lane offset, M0 = LDS address)
// -- and only CHECKED in the phase; elements that had not arrived fall back to the register polling of kr_pair_collect.
template <int NP, int NR>
__device__ __forceinline__ void kr_pair_collect_issue(const unsigned long long *base, const unsigned counter, const int rb, double2 *dst) {
    const int lane = threadIdx.x & 63;
    const int par = counter & 1;
    const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) void *)dst;
    const unsigned long long *sb = kr_uniform_ptr(base + (size_t)par * NP * 4);
#pragma unroll
    for (int c = 0; c < NP / 64; ++c) {   // piece c = elements [64 c, 64 c + 64): lane i <-> element 64 c + i, the own block's lanes stay out
        const int e = 64 * c + lane;
        if (e < rb * NR || e >= rb * NR + NR) {
            const unsigned voff = (unsigned)e * 16u;
            const unsigned ld = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds + 1024u * (unsigned)c));
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(sb), "s"(ld) : "memory");
        }
    }
}
// wave 0: the early pass is retired and the landed vector READ in front of the phase's first barrier (the LDS latency of the four
// reads hides in the barrier), tested behind it; false = some element had not arrived (kr_pair_collect then polls them in)
// (only the low dwords, which carry the phase bit, are read: 2 x NP / 64 registers across the barrier)
template <int NP>
__device__ __forceinline__ void kr_pair_check_load(const double2 *dst, unsigned (&v)[2 * (NP / 64)]) {
    const int lane = threadIdx.x & 63;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned *u = (const unsigned *)dst;
#pragma unroll
    for (int c = 0; c < NP / 64; ++c) { v[2 * c] = u[(64 * c + lane) * 4]; v[2 * c + 1] = u[(64 * c + lane) * 4 + 2]; }
}
template <int NP>
__device__ __forceinline__ bool kr_pair_check_test(const KrPairBox &box, const unsigned (&v)[2 * (NP / 64)]) {
    const unsigned phase = (((box.counter - 1u) >> 1) & 1u) ^ 1u;
    unsigned all_and = 1u, all_or = 0u;
#pragma unroll
    for (int c = 0; c < 2 * (NP / 64); ++c) { all_and &= v[c]; all_or |= v[c]; }
    const bool ok = phase ? ((all_and & 1u) == 1u) : ((all_or & 1u) == 0u);
    return __all(ok);
}

struct KrNoMid { __device__ __forceinline__ void operator()() const {} };
template <int NP, int NR>
struct KrPairMid {   // what the row-dot calls after its first chunk: wave 0 issues the early pass for the other trajectory's pending vector
    const unsigned long long *base;
    unsigned counter;
    int rb;
    double2 *dst;
    bool armed;
    __device__ __forceinline__ void operator()() const { if (armed) kr_pair_collect_issue<NP, NR>(base, counter, rb, dst); }
};

// row dot of the register-resident H(eps) block with the x slice (all reads of a chunk in flight first)
// (`mid` is called once, after the first chunk: the hook for the early collect)
// xp = &x[jq]: column jj of the thread's slice is element jj * JQ + jq of the vector
template <int JS, int JQ, class M = KrNoMid>
__device__ __forceinline__ void kr_pair_row_dot(const double (&Er)[JS], const double (&Ei)[JS], const double2 *__restrict__ xp,
                                                double &sr, double &si, M mid = M()) {
    constexpr int CH = (JS >= 8) ? 8 : JS;
    double aa = 0, cc = 0;   // (two accumulators, the subtraction inside the FMA: two fp64 instructions less per row-dot and wave)
#pragma unroll
    for (int j0 = 0; j0 < JS; j0 += CH) {
        if (j0 == (KR_PAIR_EARLY_AT ? CH : 0) || (JS == CH && j0 == 0)) mid();
        double2 x[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) x[u] = xp[(j0 + u) * JQ];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const double hr = Er[j0 + u], hi = Ei[j0 + u];
            aa = fma(hr, x[u].x, aa); cc = fma(hr, x[u].y, cc);
            aa = fma(-hi, x[u].y, aa); cc = fma(hi, x[u].x, cc);
        }
    }
    sr = kr_row_sum_wide<JQ>(aa);
    si = kr_row_sum_wide<JQ>(cc);
}

// =============================================================================================
// UPDATE = false: known-pulse sweep (dir = +1 forward, -1 backward through the adjoint pack); true: update sweep.
// Grid: P = (K / 2) * R workgroups of 512 threads; workgroup -> (pair `tile`, row block rb) as in kr_rowsplit_sweep.
// =============================================================================================
// MULTI (round 3): more pairs than groups -- every group of R workgroups carries a.rs_sets pairs in turn (pair s * tiles + tile in
// set s).  The blocks of H(eps) and H_1 are resident once and serve every set (shared generator).  Known-pulse sweeps: the sets are
// independent, one whole sweep after the other.  Update sweep: per step, phase A of every set (partial sums accumulate), ONE
// exchange, then the rounds of every set; between its turns a set keeps psi(t_n) (the all-gathered vectors) in a per-workgroup
// global scratch and the phase-A products of its rows in LDS.  N = 256 with 65..256 trajectories: ~S x the C5 step instead of
// the streamed-generator kernels (8 x).  MULTI = false is the kernel of C5, unchanged.
// POLY (round 5; update sweep only): a non-linear control -- `eps` carries f(eps_guess), `amp_g` the guess pulses, `fprime_g` f'(eps_guess),
// `poly` the cubic's coefficients, as in kr_rowsplit_sweep (kr_device_rowsplit.h).  An instantiation of its own: as a run-time branch the
// same lines cost the linear kernel of C5 8 B more scratch and 0.8 % (36.71 -> 37.01 ms per update sweep).
#define KR_PAIR_MAX_SETS 4
template <int NP, int NR, bool UPDATE, bool MULTI = false, bool POLY = false>
__global__ __launch_bounds__(512, 2) void kr_rowsplit_pair_sweep(const KrSweepArgs a) {
    using G = KrGeomR<NP, NR, 1>;
    constexpr int JS = G::JS, JQ = G::JQ, R = G::R, T = G::THREADS, KP = 2;
    static_assert(G::JSL == 0 && JS <= 16, "pair mode keeps the H(eps) block in registers: NR <= 32");
    // the control block H_1: columns [0, JSH) of the thread's slice in registers, [JSH, JS) in LDS (thread-major, conflict-free).
    // It is touched three times per step (phase-A product, two H(eps) updates), not in the rounds: with all of it in registers
    // the update kernel spilled (184-208 B of scratch at NR = 32; 72-140 B with half of it in LDS).
    constexpr int JSH = (JS > 8) ? JS / 8 : JS;
    __shared__ __attribute__((aligned(16))) double2 sH1[(JS > JSH) ? (JS - JSH) * T : 1];
#define KR_H1R(jj) (((jj) < JSH) ? H1r[(jj) < JSH ? (jj) : 0] : sH1[((jj) < JSH ? 0 : (jj) - JSH) * T + tid].x)
#define KR_H1I(jj) (((jj) < JSH) ? H1i[(jj) < JSH ? (jj) : 0] : sH1[((jj) < JSH ? 0 : (jj) - JSH) * T + tid].y)
    __shared__ __attribute__((aligned(16))) double2 sx[KP][2][NP];   // full state vectors psi (or chi) of both trajectories, by step parity (plain: x[e])
    __shared__ __attribute__((aligned(16))) double2 sy[KP][2][NP];   // Horner vectors, ping-pong
    __shared__ double spart[2 * NR];
    __shared__ double stot[2 + 1];
    __shared__ int sflag[2];
    __shared__ int salive;
    __shared__ double sab[MULTI ? KR_PAIR_MAX_SETS : 1][KP][4][MULTI ? NR : 1];   // phase-A products b, a_1 of the rows, per set (update sweep)
    __shared__ unsigned scnt[MULTI ? KR_PAIR_MAX_SETS : 1][KP];                  // all-gather counters of the sets' trajectories
    const int S = MULTI ? a.rs_sets : 1;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles = a.P / R;   // pairs
    const int wg = a.xcd_group ? (((int)(blockIdx.x & 7) * (tiles >> 3) + (int)(blockIdx.x >> 3) / R) * R + (int)(blockIdx.x >> 3) % R) : (int)blockIdx.x;
    const int tile = wg / R, rb = wg % R;
    const int il = tid / JQ, jq = tid % JQ;
    const int row = rb * NR + il;
    const bool owner = (jq == 0), row_ok = (row < a.N);
    const double my_rinv = kr_rinv[lane <= KR_MAX_ORDER ? lane : 0];
    const double my_thr = kr_load_my_thr(a);
    const KrGen<1> gx = kr_gen_ctx<1>(a, a.wg_gen ? a.wg_gen[tile] : 0);
    const int xoff = jq;   // strided column map: column jj of this thread <-> element jj * JQ + jq
    const int N = a.N, NT = a.NT;
    const double *econ_rows = a.econ ? a.econ + ((size_t)gx.gen * NT) * KR_ECON_W + (lane & (KR_ECON_W - 1)) : nullptr;
    const size_t slab = (size_t)a.Kp * N;
    size_t kbase[KP];
    kbase[0] = (size_t)(2 * tile) * N; kbase[1] = (size_t)(2 * tile + 1) * N;
    // per-workgroup scratch of the sets' state vectors (MULTI): [wg][set][q][NP]
    double2 *vscr = nullptr;
    if constexpr (MULTI) vscr = a.rs_scratch + (size_t)wg * KR_PAIR_MAX_SETS * KP * NP;

    const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)gx.Hpack, 0, (int)(a.pack_stride * 16), 0x00020000);
    KrPairBox box[KP];
    unsigned long long dbg_tpoll = 0, dbg_npass = 0;
#pragma unroll
    for (int q = 0; q < KP; ++q) {
        box[q].base = a.gmail + (size_t)(2 * tile + q) * 2 * NP * 4;
        box[q].counter = 0;
        box[q].local = false;
    }
    // set s of this group: pair s * tiles + tile
    auto select_set = [&](const int s) {
        if constexpr (MULTI) {
#pragma unroll
            for (int q = 0; q < KP; ++q) {
                const size_t traj = (size_t)2 * ((size_t)s * tiles + tile) + q;
                kbase[q] = traj * N;
                box[q].base = a.gmail + traj * 2 * NP * 4;
            }
        }
    };
    if (a.xcd_group) {
        // every workgroup posts {launch tag | XCC id}; wave 0 reads its R group members: local iff all ids agree
        unsigned long long *xtab = a.gmail + (size_t)tiles * S * KP * 2 * NP * 4;
        if (tid < 64) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            const unsigned ltag = a.epoch_base2 | 0x800000u;
            if (lane == 0) __hip_atomic_store((kr_gu64 *)xtab + wg, ((unsigned long long)ltag << 32) | (xcc & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long t0 = kr_realtime();
            bool same = false;
            for (;;) {
                const unsigned long long v = __hip_atomic_load((kr_gu64 *)xtab + tile * R + (lane % R), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool got = ((unsigned)(v >> 32) == ltag);
                if (__all(got)) { same = __all((unsigned)(v & 15u) == (xcc & 15u)); break; }
                if (kr_realtime() - t0 > a.spin_ticks / 64) break;
            }
            if (lane == 0) sflag[0] = same ? 1 : 0;
        }
        __syncthreads();
        box[0].local = box[1].local = (sflag[0] != 0);
        __syncthreads();
    }

    // full initial vectors into LDS, own elements into registers
    double pr[KP], pim[KP], cr[KP], ci[KP];
    auto init_set = [&]() {
#pragma unroll
        for (int q = 0; q < KP; ++q) {
            for (int e = tid; e < NP; e += T) {
                double2 v = make_double2(0.0, 0.0);
                if (e < N) v = a.init[kbase[q] + e];
                sx[q][0][e] = v;
            }
            pr[q] = 0.0; pim[q] = 0.0; cr[q] = 0.0; ci[q] = 0.0;
            if (row_ok) {
                const double2 v = a.init[kbase[q] + row];
                pr[q] = v.x; pim[q] = v.y;
                if (UPDATE) { const double2 c = a.X[kbase[q] + row]; cr[q] = c.x; ci[q] = c.y; }
                if (a.store && owner) a.store[(size_t)((a.dir > 0) ? 0 : NT) * slab + kbase[q] + row] = v;
            }
        }
    };
    if constexpr (!(MULTI && UPDATE)) init_set();   // (update sweep with sets: every set loads its vectors at its turn, step 0 from a.init)
    else {
#pragma unroll
        for (int q = 0; q < KP; ++q) { pr[q] = 0.0; pim[q] = 0.0; cr[q] = 0.0; ci[q] = 0.0; }
    }
    if constexpr (MULTI) { if (tid < KR_PAIR_MAX_SETS * KP) scnt[tid / KP][tid % KP] = 0u; }
    // the control block H_1, resident for the whole sweep (registers + LDS); the H(eps) block: Er / Ei
    double Er[JS], Ei[JS], H1r[JSH], H1i[JSH];
#pragma unroll
    for (int jj = 0; jj < JS; ++jj) {
        const unsigned soff = (unsigned)((((1 * JS + jj) * R + rb) * T) * 16);
        const kr_u4 raw = __builtin_amdgcn_raw_buffer_load_b128(hrsrc, tid * 16, soff, 0);
        double2 d;
        __builtin_memcpy(&d, &raw, 16);
        if (jj < JSH) { H1r[jj < JSH ? jj : 0] = d.x; H1i[jj < JSH ? jj : 0] = d.y; }
        else sH1[(jj < JSH ? 0 : jj - JSH) * T + tid] = d;
        Er[jj] = 0.0; Ei[jj] = 0.0;
    }
    double gint = 0.0;
    if (tid == 0) salive = 1;
    __syncthreads();

    const bool stamp = (a.debug != nullptr) && (wg == 0);
    unsigned long long tS = 0, tD = 0, tG = 0, tX = 0, t_prev = 0, nG = 0;
    unsigned long long prA = 0, prB = 0, prC = 0; (void)prA; (void)prB; (void)prC;
    if (stamp) t_prev = __builtin_amdgcn_s_memtime();
#define KR_STAMP(acc) if (stamp) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - t_prev; t_prev = t_; }
    // One interleaving phase: deposit trajectory q's new element (vr, vi) into dst, then -- between ONE pair of barriers -- wave 1
    // publishes q's block while wave 0 collects the OTHER trajectory's vector `odst`, published one phase earlier (its latency was
    // covered by q's row-dot).  `ocollect` false: nothing is pending for the other trajectory (first phase of a step).
#ifdef KR_PAIR_PROBE
#define KR_PROBE(acc) if (stamp) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - tp_; tp_ = t_; }
#else
#define KR_PROBE(acc)
#endif
#ifndef KR_PAIR_ONEBAR
#define KR_PAIR_ONEBAR 1   // one barrier per phase: the row owners publish their elements themselves (from registers)
#endif
#if KR_PAIR_ONEBAR
// One barrier per phase (end of round 3).  The first barrier of the two-barrier phase below only served wave 1, which read the deposited
// block back from LDS to publish it; here every row owner stores its element into the mailbox itself, straight from its registers,
// next to the LDS deposit.  Wave 0 retires its early pass and reads the landed vector's phase bits BEFORE it deposits (its own
// publishing store is then the youngest memory operation it has in flight: nothing waits for its acknowledgement except the rare
// fall-back poll), tests them behind the store, and the one barrier closes the phase.  C5, same box: known-pulse step 14.87 -> 13.91,
// update step 18.98 -> 18.16 us.  (Also tried: the bits read in the row-dot's tail, in front of the lane reductions -- the early pass
// has often not landed yet there and the wave stalls inside its row-dot: 13.9 -> 14.7, 18.2 -> 19.7 us.)
#define KR_PAIR_PHASE(q_, vr_, vi_, dst_, ocollect_, odst_, early_)                                                      \
    {                                                                                                                    \
        unsigned long long tp_ = stamp ? __builtin_amdgcn_s_memtime() : 0; (void)tp_;                                    \
        unsigned chk_[2 * (NP / 64)];                                                                                    \
        if ((ocollect_) && (early_) && wave == 0) kr_pair_check_load<NP>(odst_, chk_);                                   \
        kr_pair_deposit_publish<NP, NR, JS>(box[q_], rb, owner, il, vr_, vi_, dst_);                                     \
        KR_PROBE(prA)                                                                                                    \
        if ((ocollect_) && wave == 0) {                                                                                  \
            KR_PROBE(prB)                                                                                                \
            const bool got_ = (early_) && kr_pair_check_test<NP>(box[1 - (q_)], chk_);                                   \
            KR_PROBE(prC)                                                                                                \
            if (!got_) { if (!kr_pair_collect<NP, NR, JS>(a, box[1 - (q_)], rb, odst_, dbg_tpoll, dbg_npass) && lane == 0) salive = 0; } \
        }                                                                                                                \
        __syncthreads();                                                                                                 \
    }
#else
#define KR_PAIR_PHASE(q_, vr_, vi_, dst_, ocollect_, odst_, early_)                                                      \
    {                                                                                                                    \
        unsigned long long tp_ = stamp ? __builtin_amdgcn_s_memtime() : 0; (void)tp_;                                    \
        unsigned chk_[2 * (NP / 64)];                                                                                    \
        /* the landed vector's phase bits (low dwords only: 8 registers) are read IN FRONT of the barrier, where the LDS latency  \
           hides in the barrier wait: C5 known-pulse step 16.95 -> 16.05 us, update step 19.35 -> 19.05 us (same-box A/B) */    \
        if (KR_PAIR_PRE && (ocollect_) && (early_) && wave == 0) kr_pair_check_load<NP>(odst_, chk_);                     \
        kr_pair_deposit<NP, NR, JS>(box[q_], rb, owner, il, vr_, vi_, dst_);                                             \
        __syncthreads();                                                                                                 \
        KR_PROBE(prA)                                                                                                    \
        kr_pair_publish<NP, NR, JS>(box[q_], rb, dst_);                                                                  \
        if ((ocollect_) && wave == 0) {                                                                                  \
            if (!KR_PAIR_PRE && (early_)) kr_pair_check_load<NP>(odst_, chk_);                                           \
            KR_PROBE(prB)                                                                                                \
            const bool got_ = (early_) && (KR_PAIR_NOVERIFY || kr_pair_check_test<NP>(box[1 - (q_)], chk_));             \
            KR_PROBE(prC)                                                                                                \
            if (!got_) { if (!kr_pair_collect<NP, NR, JS>(a, box[1 - (q_)], rb, odst_, dbg_tpoll, dbg_npass) && lane == 0) salive = 0; } \
        }                                                                                                                \
        __syncthreads();                                                                                                 \
    }
#endif
    // (a collect that gave up is noticed once per step, below: the status word makes every later spin of the sweep return at once)
    // Collect only (the last pending vector of a step): wave 0 polls, one barrier.
#define KR_PAIR_COLLECT(q_, dst_)                                                                                        \
    {                                                                                                                    \
        if (wave == 0) { if (!kr_pair_collect<NP, NR, JS>(a, box[q_], rb, dst_, dbg_tpoll, dbg_npass) && lane == 0) salive = 0; } \
        __syncthreads();                                                                                                 \
    }

    const int KS = (MULTI && !UPDATE) ? S : 1;   // known-pulse sweeps: one whole sweep per set
    int kset = 0;
    // Coherence word of the group (round 4; kr_rowsplit_sweep has the same): the 2R workgroups of a pair end the sweep with bitwise
    // identical copies of BOTH state vectors -- of every set they carry.  Each thread folds the final elements it can see into a
    // position-sensitive checksum (kr_pair_fold), the workgroup posts the XOR in its slot of the XCC-id table behind the mailboxes,
    // and kr_rowsplit_check (krotov_hip.hip) compares the members of every group after the launch.
    unsigned long long coh = 0ull;
#define KR_PAIR_FOLD(set_, cur_)                                                                                       \
    {                                                                                                                  \
        _Pragma("unroll") for (int q_ = 0; q_ < KP; ++q_) {                                                            \
            for (int e_ = tid; e_ < NP; e_ += T) {                                                                     \
                const double2 v_ = sx[q_][cur_][e_];                                                                   \
                const unsigned long long bx_ = (unsigned long long)__double_as_longlong(v_.x), by_ = (unsigned long long)__double_as_longlong(v_.y); \
                const int i_ = ((set_) * KP + q_) * NP + e_, r1_ = i_ & 63, r2_ = (i_ * 7 + 13) & 63;                   \
                coh ^= ((bx_ << r1_) | (r1_ ? bx_ >> (64 - r1_) : 0ull)) ^ ((by_ << r2_) | (r2_ ? by_ >> (64 - r2_) : 0ull)); \
            }                                                                                                          \
        }                                                                                                              \
    }
kr_next_set:
    {
    if constexpr (MULTI && !UPDATE) if (kset > 0) {
        __syncthreads();
        select_set(kset);
        box[0].counter = 0; box[1].counter = 0;   // (every trajectory has its own mailbox)
        init_set();
        __syncthreads();
    }
    double ecur = 0.0;
    int cur = 0;
    for (int step = 0; step < NT; ++step) {
        const int n = (a.dir > 0) ? step : (NT - 1 - step);
        const double dtn = a.dt[n];
        const double sdt = (a.dir > 0) ? dtn : -dtn;
        const double eg = a.eps[n];
        double e = eg, de = 0.0;
        // row of the step's economized polynomial (lane j: entry j), consumed behind the exchange; 0 = none, the step takes Taylor's plan
        const double ec = econ_rows ? econ_rows[(size_t)n * KR_ECON_W] : 0.0;
        // H(eps_guess) block: rebuilt from the drift every KR_RS_REFRESH steps (one streamed pass over H0; bounds the accumulated
        // rounding), otherwise advanced in registers by (eps - ecur) H_1
        if ((step % KR_RS_REFRESH) == 0) {
#pragma unroll
            for (int jj = 0; jj < JS; ++jj) {
                const unsigned soff = (unsigned)((((0 * JS + jj) * R + rb) * T) * 16);
                const kr_u4 raw = __builtin_amdgcn_raw_buffer_load_b128(hrsrc, tid * 16, soff, 0);
                double2 d;
                __builtin_memcpy(&d, &raw, 16);
                Er[jj] = fma(eg, KR_H1R(jj), d.x);
                Ei[jj] = fma(eg, KR_H1I(jj), d.y);
            }
        } else {
            const double dinc = eg - ecur;
#pragma unroll
            for (int jj = 0; jj < JS; ++jj) { Er[jj] = fma(dinc, KR_H1R(jj), Er[jj]); Ei[jj] = fma(dinc, KR_H1I(jj), Ei[jj]); }
        }
        double br[KP], bi[KP], a1r[KP], a1i[KP];
        if (UPDATE) {
            // phase A for both trajectories: b = H(eps_g) psi and a_1 = H_1 psi for the own rows, one pass over x each
            double part[2], du[2];
            part[0] = 0.0; part[1] = 0.0;
            for (int as = 0; as < S; ++as) {
            if constexpr (MULTI) {
                // set `as` takes its turn: state vectors from the scratch (step 0: the initial states), chi(t_n) of the own rows
                select_set(as);
                __syncthreads();   // the previous set's row-dots have read sx
                const int q = tid >> 8, e = tid & 255;   // 512 threads: one element of one trajectory each (NP = 256; NP = 128: half idle)
                if (e < NP) {
                    double2 v = make_double2(0.0, 0.0);
                    if (step == 0) { if (e < N) v = a.init[kbase[q] + e]; }
                    else v = vscr[((size_t)as * KP + q) * NP + e];
                    sx[q][cur][e] = v;
                }
#pragma unroll
                for (int q2 = 0; q2 < KP; ++q2) {
                    cr[q2] = 0.0; ci[q2] = 0.0;
                    if (row_ok) {
                        const double2 c = a.X[(size_t)n * slab + kbase[q2] + row];
                        cr[q2] = c.x; ci[q2] = c.y;
                        if (step == 0 && a.store && owner) a.store[kbase[q2] + row] = a.init[kbase[q2] + row];
                    }
                }
                __syncthreads();
            }
#pragma unroll
            for (int q = 0; q < KP; ++q) {
                {   // ONE pass over the x slice for both blocks, four columns at a time
                    const double2 *xp = &sx[q][cur][xoff];
                    double aa[2] = {0, 0}, bb[2] = {0, 0}, cc[2] = {0, 0}, dd[2] = {0, 0};
                    constexpr int CH = (JS >= 4) ? 4 : JS;
#pragma unroll
                    for (int j0 = 0; j0 < JS; j0 += CH) {
                        double2 x[CH];
#pragma unroll
                        for (int u = 0; u < CH; ++u) x[u] = xp[(j0 + u) * JQ];
#pragma unroll
                        for (int u = 0; u < CH; ++u) {
                            const double er = Er[j0 + u], ei = Ei[j0 + u], hr = KR_H1R(j0 + u), hi = KR_H1I(j0 + u);
                            aa[0] = fma(er, x[u].x, aa[0]); bb[0] = fma(ei, x[u].y, bb[0]);
                            cc[0] = fma(er, x[u].y, cc[0]); dd[0] = fma(ei, x[u].x, dd[0]);
                            aa[1] = fma(hr, x[u].x, aa[1]); bb[1] = fma(hi, x[u].y, bb[1]);
                            cc[1] = fma(hr, x[u].y, cc[1]); dd[1] = fma(hi, x[u].x, dd[1]);
                        }
                    }
                    br[q] = kr_row_sum_wide<JQ>(aa[0] - bb[0]); bi[q] = kr_row_sum_wide<JQ>(cc[0] + dd[0]);
                    a1r[q] = kr_row_sum_wide<JQ>(aa[1] - bb[1]); a1i[q] = kr_row_sum_wide<JQ>(cc[1] + dd[1]);
                }
                if ((gx.present & 1u)) part[0] += cr[q] * a1i[q] - ci[q] * a1r[q];
                if (!MULTI && n + 1 < NT && row_ok) { const double2 c = a.X[(size_t)(n + 1) * slab + kbase[q] + row]; cr[q] = c.x; ci[q] = c.y; }
                if constexpr (MULTI) if (owner) { sab[as][q][0][il] = br[q]; sab[as][q][1][il] = bi[q]; sab[as][q][2][il] = a1r[q]; sab[as][q][3][il] = a1i[q]; }
            }
            }   // sets
            KR_STAMP(tS)
            if (!kr_exchange<2, NR, KR_RS_SCHED>(a, n, wg, owner, il, part, du, spart, stot)) return;
            KR_STAMP(tX)
            const double alpha = a.alpha[n];
            double dul = du[0], pn;
            if constexpr (POLY) {   // mu = f'(eps_guess) H_1 (optimize.jl:337,345); the amplitude of the updated pulse as the cubic's Taylor form at the guess
                const double pg = a.amp_g[n], fp = a.fprime_g[n], c2 = a.poly[2], c3 = a.poly[3];
                dul = __dmul_rn(fp, du[0]);
                pn = __dadd_rn(pg, __dmul_rn(alpha, dul));
                e = eg + kr_poly3_delta(pn - pg, fp, fma(3.0 * c3, pg, c2), c3);
            } else {
                pn = __dadd_rn(eg, __dmul_rn(alpha, dul));
                e = pn;
            }
            de = e - eg;
            const double ad = fabs(dul);
            gint = __dadd_rn(gint, __dmul_rn(__dmul_rn(alpha, __dmul_rn(ad, ad)), dtn));
            // (no log of the gathered totals here, unlike every other persistent update kernel: this kernel sits at 256 VGPRs with scratch, and
            //  both placements of the one store -- lane 0 of wave 2 behind the exchange, or the eps_new store below redirected per workgroup
            //  -- cost C5 1.4 % of the iteration, the second one through the KNOWN-pulse instantiation of this very template, 28.55 -> 29.12 ms)
            if (wg == 0 && tid == 0) a.eps_new[n] = pn;
            // H(eps_new) = H(eps_g) + d eps H_1, in registers
#pragma unroll
            for (int jj = 0; jj < JS; ++jj) { Er[jj] = fma(de, KR_H1R(jj), Er[jj]); Ei[jj] = fma(de, KR_H1I(jj), Ei[jj]); }
        }
        ecur = e;
        double theta = gx.nrm[0] + fabs(e) * gx.nrm[1];
        theta *= fabs(sdt);
        int s, m;
        kr_plan_fast(a, my_thr, theta, s, m);
        // economized step polynomial (kr_device.h: KrEcon; here also for generators that are only NEARLY Hermitian -- a Liouvillian
        // with weak dissipation: kr_econ_rows_kernel bounds the truncation on an ellipse around the numerical range)
        int se = s, me = m;
        KrEcon eco = kr_econ_select(ec, theta, se, me);
        if (eco.on && (s > 1 || me < m)) { s = se; m = me; }          // (a row whose degree saves nothing is not used: the ellipse bound
        else { eco.on = false; eco.a0 = 1.0; eco.a1 = 1.0; }          //  of a generator far from Hermitian can exceed Taylor's)
        const double my_g = eco.on ? ec : my_rinv;   // Horner ratios of this step, lane j holds g_j
        const double fs = (s == 1) ? sdt : sdt / (double)s;

        for (int bs = 0; bs < ((MULTI && UPDATE) ? S : 1); ++bs) {
        if constexpr (MULTI && UPDATE) {
            // set `bs` takes its turn: its rows' phase-A products from LDS, psi(t_n) of the own rows from the scratch (it is needed
            // last -- psi + y_1 --, so the load's latency hides behind the rounds), the all-gather counters of its trajectories
            select_set(bs);
#pragma unroll
            for (int q = 0; q < KP; ++q) {
                box[q].counter = scnt[bs][q];
                br[q] = sab[bs][q][0][il]; bi[q] = sab[bs][q][1][il]; a1r[q] = sab[bs][q][2][il]; a1i[q] = sab[bs][q][3][il];
                pr[q] = 0.0; pim[q] = 0.0;
                if (row_ok) {
                    const double2 v = (step == 0) ? a.init[kbase[q] + row] : vscr[((size_t)bs * KP + q) * NP + row];
                    pr[q] = v.x; pim[q] = v.y;
                }
            }
        }
        for (int sub = 0; sub < s; ++sub) {
            // Horner as in kr_rowsplit_sweep, the two trajectories alternating: with the phase-A products known (update sweep, first
            // sub-step) the constant is v = -i fs H psi and the result psi + y_1; otherwise x0 = psi and the result is y_0.
            const bool have_v = UPDATE && (sub == 0);
            double x0r[KP], x0i[KP], yr[KP], yi[KP];
            double2 *src[KP];
            int w = 0, jlast;
            // pend1: trajectory 1's source vector src[1] has been published but not collected yet (it is collected inside
            // trajectory 0's next phase, i.e. behind trajectory 0's row-dot)
            bool pend1 = false;
            if (have_v) {
#pragma unroll
                for (int q = 0; q < KP; ++q) {
                    const double wr = fma(de, a1r[q], br[q]), wi = fma(de, a1i[q], bi[q]);
                    x0r[q] = fs * wi; x0i[q] = -fs * wr;
                    yr[q] = x0r[q]; yi[q] = x0i[q];
                }
                if (m > 1) {   // v of both trajectories -> sy[q][0]
                    KR_PAIR_PHASE(0, x0r[0], x0i[0], sy[0][0], false, sy[1][0], false)
                    KR_PAIR_PHASE(1, x0r[1], x0i[1], sy[1][0], true, sy[0][0], false)
                    pend1 = true;
                }
                w = 1; jlast = 2;
                src[0] = sy[0][0]; src[1] = sy[1][0];
            } else {
#pragma unroll
                for (int q = 0; q < KP; ++q) { x0r[q] = pr[q]; x0i[q] = pim[q]; yr[q] = pr[q]; yi[q] = pim[q]; }
                jlast = 1;
                src[0] = sx[0][cur]; src[1] = sx[1][cur];
            }
            for (int j = m; j >= jlast; --j) {
                const bool last = (j == jlast);
                // (known-pulse form, last round: a_0 psi + a_1 z y_1 -- entry 1 of the row is a_1 itself, not a ratio; Taylor: 1, 1)
                const bool scale0 = !have_v && last;
                const double f = scale0 ? fs * eco.a1 : fs * kr_rinv_lane(my_g, j);
#pragma unroll
                for (int q = 0; q < KP; ++q) {
                    double sr, si;
                    // the other trajectory's pending vector (q = 0: trajectory 1's source of this round, if pending; q = 1: what
                    // trajectory 0 has just published): its poll pass is issued from inside this row-dot
                    const bool early = KR_PAIR_EARLY && (q == 1 || pend1);
                    double2 *pending = (q == 0) ? src[1] : (last ? sx[0][cur ^ 1] : sy[0][w]);   // where the other trajectory's vector lands
                    const KrPairMid<NP, NR> mid{box[1 - q].base, box[1 - q].counter, rb, pending, early && wave == 0};
                    kr_pair_row_dot<JS, JQ>(Er, Ei, src[q] + xoff, sr, si, mid);
                    yr[q] = fma(f, si, scale0 ? eco.a0 * x0r[q] : x0r[q]);
                    yi[q] = fma(-f, sr, scale0 ? eco.a0 * x0i[q] : x0i[q]);
                    if (last && have_v) { yr[q] = fma(eco.a1, yr[q], eco.a0 * pr[q]); yi[q] = fma(eco.a1, yi[q], eco.a0 * pim[q]); }   // psi' = a_0 psi + a_1 y_1  (Taylor: 1, 1)
                    double2 *dst = last ? sx[q][cur ^ 1] : sy[q][w];
                    KR_STAMP(tD)
                    // deposit + publish q; meanwhile collect what the OTHER trajectory published one phase earlier:
                    //   q = 0: trajectory 1's pending source vector src[1] (its result of the previous round, or its v)
                    //   q = 1: trajectory 0's result of THIS round (the dst of the q = 0 phase)
                    if (q == 0) {
                        KR_PAIR_PHASE(0, yr[0], yi[0], dst, pend1, src[1], early)
                    } else {
                        double2 *od = last ? sx[0][cur ^ 1] : sy[0][w];
                        KR_PAIR_PHASE(1, yr[1], yi[1], dst, true, od, early)
                        pend1 = true;
                    }
                    KR_STAMP(tG)
                    nG++;
                }
                src[0] = sy[0][w]; src[1] = sy[1][w];
                w ^= 1;
            }
            if (have_v && m == 1) {  // psi' = psi + v : no round ran
#pragma unroll
                for (int q = 0; q < KP; ++q) { yr[q] = fma(eco.a1, x0r[q], eco.a0 * pr[q]); yi[q] = fma(eco.a1, x0i[q], eco.a0 * pim[q]); }
                KR_PAIR_PHASE(0, yr[0], yi[0], sx[0][cur ^ 1], false, sx[1][cur ^ 1], false)
                KR_PAIR_PHASE(1, yr[1], yi[1], sx[1][cur ^ 1], true, sx[0][cur ^ 1], false)
            }
            // trajectory 1's last vector is still in flight
            KR_PAIR_COLLECT(1, sx[1][cur ^ 1])
#pragma unroll
            for (int q = 0; q < KP; ++q) { pr[q] = yr[q]; pim[q] = yi[q]; }
            cur ^= 1;
            if (salive == 0) return;   // (uniform: read behind the last collect's barrier; set when a bounded spin gave up)
        }
        if (a.store && owner && row_ok) {
            const int slot = (a.dir > 0) ? (n + 1) : n;
#pragma unroll
            for (int q = 0; q < KP; ++q) a.store[(size_t)slot * slab + kbase[q] + row] = make_double2(pr[q], pim[q]);
        }
        if constexpr (MULTI && UPDATE) {
            // the set's turn ends: psi(t_{n+1}) -- complete in every workgroup of the group -- goes to this workgroup's scratch
            const int q = tid >> 8, e = tid & 255;
            if (e < NP) vscr[((size_t)bs * KP + q) * NP + e] = sx[q][cur][e];
            if (step == NT - 1) KR_PAIR_FOLD(bs, cur)
            if (tid < KP) scnt[bs][tid] = box[tid].counter;
            if (step == NT - 1 && a.fin && owner && row_ok) {
#pragma unroll
                for (int q2 = 0; q2 < KP; ++q2) a.fin[kbase[q2] + row] = make_double2(pr[q2], pim[q2]);
            }
            __syncthreads();   // (scratch stores are this workgroup's own: visible to its later loads; the barrier orders the LDS counters)
        }
        }   // sets
    }
    if (stamp && tid == 0) { a.debug[0] = tS; a.debug[1] = tD; a.debug[2] = tG; a.debug[3] = tX; a.debug[7] = nG; a.debug[5] = dbg_tpoll; a.debug[6] = dbg_npass;
#ifdef KR_PAIR_PROBE
        a.debug[4] = prA; a.debug[5] = prB; a.debug[6] = prC;
#endif
    }
#undef KR_STAMP
#undef KR_PAIR_PHASE
#undef KR_PAIR_COLLECT
#undef KR_H1R
#undef KR_H1I
    if constexpr (!(MULTI && UPDATE)) if (a.fin && owner && row_ok) {
#pragma unroll
        for (int q = 0; q < KP; ++q) a.fin[kbase[q] + row] = make_double2(pr[q], pim[q]);
    }
    if constexpr (!(MULTI && UPDATE)) {
        if (a.rs_fault && wg == a.rs_fault && tid == 0 && kset == 0) sx[1][cur][0].x = __longlong_as_double(__double_as_longlong(sx[1][cur][0].x) ^ 2ll);   // (tests: a fault to find)
        if (a.rs_fault) __syncthreads();
        KR_PAIR_FOLD(kset, cur)
    }
    }   // known-pulse sets
    if constexpr (MULTI && !UPDATE) { if (++kset < KS) goto kr_next_set; }
    (void)kset; (void)KS;
#undef KR_PAIR_FOLD
    {
        __shared__ unsigned long long scs[T / 64];
        if (MULTI && UPDATE && a.rs_fault && wg == a.rs_fault && tid == 0) coh ^= 2ull;   // (tests: a fault to find)
        for (int o = 1; o < 64; o <<= 1) coh ^= __shfl_xor(coh, o, 64);
        __syncthreads();
        if (lane == 0) scs[wave] = coh;
        __syncthreads();
        if (tid == 0) {
            unsigned long long all = 0x9E3779B97F4A7C15ull;   // (never zero by accident: a zero slot means "not posted")
            for (int w = 0; w < T / 64; ++w) all = (all << 1 | all >> 63) ^ scs[w];
            __hip_atomic_store((kr_gu64 *)(a.gmail + (size_t)tiles * S * KP * 2 * NP * 4) + wg, all | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (UPDATE && wg == 0 && tid == 0) a.g_a_int[0] = gint;
}
