// Row-split sweep kernels for 64 < N <= 1024 (NP in {128, 256}; round 3: also 512 and 1024 for few trajectories): R workgroups
// share ONE trajectory.
//
// With K < #CUs and N large, one workgroup per trajectory leaves most of the chip idle and has to stream the
// whole generator (N = 256: 1 MB per matrix) from L2 for every matvec round (kr_device_stream.h: 16 us per
// round).  Here workgroup (tile, rb) owns rows [rb*NR, (rb+1)*NR) of the generator: its block of H(eps) stays in
// registers for all rounds of a step (NR = 64, NP = 256: 128 VGPRs per thread), the blocks of H0, H_l are streamed
// once (known-pulse sweeps) or twice (update sweep) per STEP, and after every matvec round the R workgroups of a
// trajectory all-gather the N new vector elements through a mailbox (one 16-byte word {re, im} per complex element, each double
// validated by ONE bit -- its mantissa LSB = phase of the buffer's use, as in kr_exchange: a slot has one writer, written in order,
// and the mailbox is zeroed in front of every launch; before: four {tag32 | 32 bits} granules per element, twice the poll bytes --,
// double-buffered by round parity, bounded spins).  The owners put the SAME LSB-forced values into their own copy of the vector,
// so that all R workgroups of a trajectory keep bitwise identical vectors.
// The per-step global sum of the update integrands runs over all P = K*R workgroups with kr_exchange, each
// workgroup contributing the rows it owns.
//
// Thread map: T = 512 threads (1024 for NR = 64); thread (il, jq) = (tid / JQ, tid % JQ), JQ = T / NR lanes per row, holds
// columns [jq*JS, (jq+1)*JS), JS = NP / JQ, of row rb*NR + il.
#pragma once
#include "kr_device.h"

template <int NP, int NR, int LMAX>
struct KrGeomR {
    // NR = 64: 1024 threads (4 waves/SIMD, 128-VGPR budget, H(eps) block = 4*JS = 64 VGPRs); with 512 threads the
    // block alone takes 128 of 256 VGPRs and the kernel spills (measured: 8.0k cycles per row-dot round)
    static constexpr int THREADS = 512;  // (1024 threads for NR = 64 were measured slower: 67 vs 55 us per step at C5)
    static constexpr int JQ = THREADS / NR;
    static constexpr int JS = NP / JQ;
    static constexpr int R = NP / NR;
    static constexpr int NPP = NP + JQ;
    // columns of the thread's H(eps) slice kept in registers / in LDS.  NR = 64 at NP = 256 has JS = 32 (128 VGPRs for
    // the slice alone): hipcc then keeps part of it in scratch (measured 8.0k cycles per row-dot round), so the second
    // half lives in LDS instead (16 * 512 * 16 B = 128 KB of the CU's 160 KB), thread-major and conflict-free.
    static constexpr int JSR = (JS > 16) ? 16 : JS;
    static constexpr int JSL = JS - JSR;
    static constexpr int CHUNK = (JS >= 16) ? 4 : ((JS >= 8) ? 8 : JS);  // columns per streamed chunk (register budget: the H(eps) block takes 4*JS VGPRs)
    // NR = NP = 128 (round 3): R = 1, the whole trajectory in ONE workgroup -- no all-gather, but the same resident H(eps) (half of it
    // in LDS, as at NR = 64 / NP = 256) and the same one or two streamed passes per STEP instead of the streamed-generator kernels'
    // (1 + L) matrices per ROUND: N = 128 with 129..256 trajectories
    // NP = 512 (round 3): 256 < N <= 512 with K * R <= #CUs, R = 32 / 64 workgroups per trajectory (NR = 16 / 8 rows each: 64 / 32
    // registers for the H(eps) block); the all-gather is polled by two waves.  NP = 1024: R = 128 (8 rows each, 64 registers), four
    // polling waves, at most two trajectories.
    static_assert((NR == 8 && NP <= 1024) || (NR == 16 && NP <= 512) || ((NR == 32 || NR == 64) && NP <= 256) || (NR == 128 && NP == 128),
                  "NR in {8,16,32,64} (128 at NP = 128; <= 16 at NP = 512; 8 at NP = 1024)");
    static_assert(NP == 128 || NP == 256 || NP == 512 || NP == 1024, "NP in {128,256,512,1024}");
    static_assert(JS >= 2, "at least two columns per thread");
};

// sum over the JQ consecutive lanes of a row (JQ in {8,16,32,64}); every lane gets the sum
template <int JQ>
__device__ __forceinline__ double kr_row_sum_wide(double v) {
    v += kr_dpp_mov<0xB1>(v);
    v += kr_dpp_mov<0x4E>(v);
    if (JQ >= 8) v += kr_dpp_mov<0x141>(v);   // 8 lanes
    if (JQ >= 16) v += kr_dpp_mov<0x140>(v);  // row_mirror: 16 lanes
    if (JQ >= 32) v += __shfl_xor(v, 16, 64);
    if (JQ >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

typedef unsigned int kr_u4 __attribute__((ext_vector_type(4)));

struct KrGroupBox {
    unsigned long long *base;  // this tile's mailbox: [2][NP] elements of 16 B (2 x [NP][4] words are reserved per tile)
    unsigned tag_base;
    unsigned counter;
    unsigned long long tpoll, npass;  // diagnostics (a.debug): cycles spent polling and number of poll passes, wave 0
    bool local;  // all R workgroups of the tile run on ONE XCD (checked at kernel start): publish into that XCD's L2 only
};

// One poll pass of the all-gather: up to KR_POLL_CHUNKS elements per lane, 16 B each, sc1 loads; the loads and their wait are ONE asm
// block (the compiler neither counts nor reorders asm loads).
__device__ __forceinline__ void kr_poll_elems(const unsigned long long *p0, const unsigned long long *p1, const unsigned long long *p2,
                                              const unsigned long long *p3, const int nch, kr_v4u (&g)[KR_POLL_CHUNKS]) {
    if (nch > 2)
        asm volatile(KR_PL(0, 4, 0) KR_PL(1, 5, 0) KR_PL(2, 6, 0) KR_PL(3, 7, 0) "s_waitcnt vmcnt(0)"
                     : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
    else if (nch > 1)
        asm volatile(KR_PL(0, 2, 0) KR_PL(1, 3, 0) "s_waitcnt vmcnt(0)" : "=&v"(g[0]), "=&v"(g[1]) : "v"(p0), "v"(p1) : "memory");
    else
        asm volatile(KR_PL(0, 1, 0) "s_waitcnt vmcnt(0)" : "=&v"(g[0]) : "v"(p0) : "memory");
}

// Publish this workgroup's NR new elements (row owners hold them) and gather the other blocks of the tile into
// the LDS vector `dst` (padded layout).  Returns false on abort.  All threads of the workgroup call it.
template <int NP, int NR, int JS>
__device__ __forceinline__ bool kr_group_gather(const KrSweepArgs &a, KrGroupBox &box, const int rb, const bool owner,
                                                const int il, const double vr, const double vi, double2 *dst,
                                                int *sflag) {
    const int tid = threadIdx.x;
    ++box.counter;
    const int par = box.counter & 1;
    const unsigned phase = (((box.counter - 1u) >> 1) & 1u) ^ 1u;   // first use of either buffer expects 1: zeroed memory is invalid
    kr_gu64 *mb = (kr_gu64 *)(box.base + (size_t)par * NP * 4);
    // Row owners put their element into the LDS vector; after a barrier WAVE 1 publishes the whole block (16-byte asm
    // stores the compiler does not count: nobody waits for their acknowledgement) while wave 0, which therefore never
    // stores, polls: vmcnt counts loads and stores together, and a wave that stored first waits for the acknowledgement
    // from memory with its first poll pass (1.2-1.9 us against 0.75 us for a pass of loads alone).
    if (owner) {
        const int e = rb * NR + il;
#ifdef KR_GATHER_SEQ   // diagnostic build (tools/build_diag.sh): bits 15..1 of every double carry the gather counter, see below
        const unsigned long long sq = (unsigned long long)((box.counter & 0x7fffu) << 1);
        dst[e + e / JS] = make_double2(__longlong_as_double((long long)((kr_slot_bits(vr, phase) & ~0xfffeull) | sq)), __longlong_as_double((long long)((kr_slot_bits(vi, phase) & ~0xfffeull) | sq)));
#else
        dst[e + e / JS] = make_double2(__longlong_as_double((long long)kr_slot_bits(vr, phase)), __longlong_as_double((long long)kr_slot_bits(vi, phase)));
#endif
    }
    __syncthreads();
    if constexpr (NR == NP) { (void)sflag; (void)mb; return true; }   // R = 1: the workgroup owns every row
    if ((tid >> 6) == 1 && (tid & 63) < NR) {
        const int e = rb * NR + (tid & 63);
        const double2 v = dst[e + e / JS];
        const unsigned long long br = (unsigned long long)__double_as_longlong(v.x), bi = (unsigned long long)__double_as_longlong(v.y);
        kr_v4u w;  // one 16-byte store per element: {re, im}, each double validated by its LSB
        w.x = (unsigned)br; w.y = (unsigned)(br >> 32); w.z = (unsigned)bi; w.w = (unsigned)(bi >> 32);
        const unsigned long long *pe = (const unsigned long long *)mb + 2 * e;
        // local: sc0 stores stay in the XCD's L2, where the group's sc1 loads (which bypass only the vector L1) find them
        // (292 ns per hop against 626 across XCDs, tools/ubench/pingpong.hip); otherwise sc1 stores write through to memory
        if (box.local) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(pe), "v"(w) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(pe), "v"(w) : "memory");
    }
    // wave 0 gathers the NP - NR foreign elements: lane handles elements lane, lane + 64, ...; every load of a pass
    // is in flight before the first tag is looked at (no workgroup barrier inside the spin); 16-byte sc1 loads, the
    // loads of a pass and their wait in one asm block (KrPoll, kr_device.h)
    // (more than 256 foreign elements, NP = 512: windows of 256 elements, window w polled by wave 0 (w = 0) / wave w + 1 -- wave 1
    // publishes --, each with its own bounded spin; the verdicts meet in sflag[par * 8 + w])
    bool alive = true;
    constexpr int NWIN = (NP - NR + 255) / 256;
    static_assert(NWIN <= 7, "one polling wave per window of 256 foreign elements");
    const int pwin = ((tid >> 6) == 0) ? 0 : (tid >> 6) - 1;
    if ((tid >> 6) != 1 && pwin < NWIN) {
        constexpr int EC = (NWIN == 1) ? (NP - NR + 63) / 64 : KR_POLL_CHUNKS;
        static_assert(EC <= KR_POLL_CHUNKS, "at most 256 foreign elements per window");
        const int lane = tid & 63;
        int el[KR_POLL_CHUNKS];
#pragma unroll
        for (int c = 0; c < KR_POLL_CHUNKS; ++c) {
            int q = min(pwin * 256 + lane + 64 * c, NP - NR - 1);
            if (q >= rb * NR) q += NR;
            el[c] = q;
        }
        const unsigned long long *mbc = (const unsigned long long *)mb;
        kr_v4u g[KR_POLL_CHUNKS];
        for (int w = 0; w < (box.local ? a.gather_delay_local : a.gather_delay); ++w) __builtin_amdgcn_s_sleep(2);
        const unsigned long long t0 = kr_realtime();
        const unsigned long long tm0 = a.debug ? __builtin_amdgcn_s_memtime() : 0;
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
            kr_poll_elems(mbc + 2 * el[0], mbc + 2 * el[1], mbc + 2 * el[2], mbc + 2 * el[3], EC, g);
#pragma unroll
            for (int c = 0; c < EC; ++c)
                ok = ok & ((g[c].x & 1u) == phase) & ((g[c].z & 1u) == phase);
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
        if (a.debug) { box.tpoll += __builtin_amdgcn_s_memtime() - tm0; box.npass += spins + 1; }
#ifdef KR_GATHER_SEQ
        // Did the one-bit validity accept an element of an EARLIER use of this buffer?  (Every element carries its gather counter in
        // this build; a.debug[4] counts the elements whose counter is not this gather's, a.debug[8..11] keep the last such event.)
        if (alive && a.debug) {
#pragma unroll
            for (int c = 0; c < EC; ++c) {
                if (pwin * 256 + lane + 64 * c < NP - NR) {
                    const unsigned want = box.counter & 0x7fffu, sr = (g[c].x >> 1) & 0x7fffu, si = (g[c].z >> 1) & 0x7fffu;
                    if (sr != want || si != want) {
                        atomicAdd(a.debug + 4, 1ull);
                        a.debug[8] = (unsigned long long)blockIdx.x; a.debug[9] = box.counter; a.debug[10] = (unsigned long long)el[c]; a.debug[11] = ((unsigned long long)sr << 32) | si;
                    }
                }
            }
        }
#endif
        if (alive) {
#pragma unroll
            for (int c = 0; c < EC; ++c) {
                if (pwin * 256 + lane + 64 * c < NP - NR) {
                    const double re = __longlong_as_double((long long)(((unsigned long long)g[c].y << 32) | g[c].x));
                    const double im = __longlong_as_double((long long)(((unsigned long long)g[c].w << 32) | g[c].z));
                    dst[el[c] + el[c] / JS] = make_double2(re, im);
                }
            }
        }
        if (lane == 0) sflag[par * 8 + pwin] = alive ? 0 : 1;   // two sets of slots by parity: the next gather writes the other one
    }
    __syncthreads();
    bool dead = false;
#pragma unroll
    for (int w = 0; w < NWIN; ++w) dead = dead || (sflag[par * 8 + w] != 0);
    return !dead;
}

// One streamed pass over this workgroup's blocks of H0, H_1..H_L (thread-major, [m][jj][rb][tid]):
//   MODE 0 (known sweeps): Heps = H0 + sum e_l H_l
//   MODE 1 (update, phase A): Heps = H0 + sum e_l H_l, b = Heps x, a_l = H_l x        (x = LDS vector, padded)
//   MODE 2 (update, correction): Heps += sum e_l H_l                                   (e = delta eps)
//   INC (MODE 1): start from the resident Heps instead of H0, e = eps - (pulse folded into Heps): the H0 block is not read
// H0 and H_l never change, so between two steps only sum (eps' - eps) H_l has to be added: MODE 2 / MODE 1 + INC stream L
// blocks instead of 1 + L (the pass is bound by the CU's L1 fill rate: 256 KB per block at N = 256, NR = 64).  Every
// KR_RS_REFRESH-th step rebuilds Heps from H0, which bounds the accumulated rounding (<= 32 additions of 1 ulp each).
#define KR_RS_REFRESH 32
#ifndef KR_RS_SCHED
#define KR_RS_SCHED 1   // exchange schedule of the row-split update sweep (kr_exchange): 1 = publisher wave + delayed poll (a.poll_delay x 128
                        // cycles; measured at C5, us per step: schedule 0 42.6; schedule 1 with delay 0 / 4 / 8 / 12 / 16 / 24: 41.5 / 41.2 / 41.1 / 40.9 / 40.9 / 40.9)
#endif
template <int NP, int NR, int LMAX, int MODE, bool INC = false>
__device__ __forceinline__ void kr_rs_stream(const __amdgpu_buffer_rsrc_t rsrc, const int rb, const int L, const double (&e)[LMAX],
                                             double (&Er)[KrGeomR<NP, NR, LMAX>::JSR], double (&Ei)[KrGeomR<NP, NR, LMAX>::JSR],
                                             double2 *sHl /* LDS part [JSL][512], or unused */,
                                             const double2 *xp, double (&outr)[1 + LMAX], double (&outi)[1 + LMAX]) {
    using G = KrGeomR<NP, NR, LMAX>;
    constexpr int JS = G::JS, JSR = G::JSR, R = G::R, T = G::THREADS, JQ = G::JQ;
    constexpr int CH = (MODE == 1 && G::CHUNK > 2 && JS >= 16 && LMAX > 1) ? 2 : G::CHUNK;  // phase A carries 4(1+L) accumulators: smaller chunks
                                                                                               // (one control, LMAX = 1: room for 4 columns in flight)
    const int tid = threadIdx.x;
    double ar[1 + LMAX], br[1 + LMAX], ai[1 + LMAX], bi[1 + LMAX];
#pragma unroll
    for (int o = 0; o <= LMAX; ++o) { ar[o] = 0; br[o] = 0; ai[o] = 0; bi[o] = 0; }
#pragma unroll
    for (int j0 = 0; j0 < JS; j0 += CH) {
        double2 h[LMAX + 1][CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
#pragma unroll
            for (int m = ((MODE == 2 || INC) ? 1 : 0); m <= LMAX; ++m) {
                h[m][u] = make_double2(0.0, 0.0);
                if (m <= L) {
                    // buffer load: resource (SGPRs) + one per-lane byte offset + a scalar offset per element.  With
                    // flat/global addressing hipcc hoists 96 per-load 64-bit addresses out of the step loop and
                    // spills them (368 VGPRs of scratch at NR = 64).
                    const unsigned soff = (unsigned)((((m * JS + j0 + u) * R + rb) * T) * 16);
                    const kr_u4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, tid * 16, soff, 0);
                    double2 d;
                    __builtin_memcpy(&d, &raw, 16);
                    h[m][u] = d;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int jj = j0 + u;
            double cr, ci;
            if (MODE == 2 || INC) {
                if (jj < JSR) { cr = Er[jj < JSR ? jj : 0]; ci = Ei[jj < JSR ? jj : 0]; }
                else { const double2 o = sHl[(jj - JSR) * T + tid]; cr = o.x; ci = o.y; }
            } else { cr = h[0][u].x; ci = h[0][u].y; }
#pragma unroll
            for (int l = 0; l < LMAX; ++l) { cr = fma(e[l], h[1 + l][u].x, cr); ci = fma(e[l], h[1 + l][u].y, ci); }
            if (jj < JSR) { Er[jj < JSR ? jj : 0] = cr; Ei[jj < JSR ? jj : 0] = ci; }
            else sHl[(jj - JSR) * T + tid] = make_double2(cr, ci);
            if (MODE == 1) {
                const double2 x = xp[j0 + u];
                ar[0] = fma(cr, x.x, ar[0]); br[0] = fma(ci, x.y, br[0]);
                ai[0] = fma(cr, x.y, ai[0]); bi[0] = fma(ci, x.x, bi[0]);
#pragma unroll
                for (int l = 0; l < LMAX; ++l) {
                    if (l < L) {
                        const double2 hl = h[1 + l][u];
                        ar[1 + l] = fma(hl.x, x.x, ar[1 + l]); br[1 + l] = fma(hl.y, x.y, br[1 + l]);
                        ai[1 + l] = fma(hl.x, x.y, ai[1 + l]); bi[1 + l] = fma(hl.y, x.x, bi[1 + l]);
                    }
                }
            }
        }
        if (NR >= 32) __builtin_amdgcn_sched_barrier(0);  // keep the loads of later chunks out of this one (registers)
    }
    if (MODE == 1) {
#pragma unroll
        for (int o = 0; o <= LMAX; ++o) {
            outr[o] = kr_row_sum_wide<JQ>(ar[o] - br[o]);
            outi[o] = kr_row_sum_wide<JQ>(ai[o] + bi[o]);
        }
    }
}

// row dot with this thread's slice of H(eps): columns [0, JSR) from registers, [JSR, JS) from LDS (thread-major)
template <int JS, int JSR, int JQ, int CH, int T>
__device__ __forceinline__ void kr_rs_row_dot(const double (&Er)[JSR], const double (&Ei)[JSR], const double2 *sHl,
                                              const double2 *xp, double &sr, double &si) {
    const int tid = threadIdx.x;
    double a = 0, b = 0, c = 0, d = 0;
#pragma unroll
    for (int j0 = 0; j0 < JS; j0 += CH) {
        double2 x[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) x[u] = xp[j0 + u];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int jj = j0 + u;
            double hr, hi;
            if (jj < JSR) { hr = Er[jj < JSR ? jj : 0]; hi = Ei[jj < JSR ? jj : 0]; }
            else { const double2 o = sHl[(jj - JSR) * T + tid]; hr = o.x; hi = o.y; }
            a = fma(hr, x[u].x, a); b = fma(hi, x[u].y, b);
            c = fma(hr, x[u].y, c); d = fma(hi, x[u].x, d);
        }
        if (CH <= 4) __builtin_amdgcn_sched_barrier(0);
    }
    sr = kr_row_sum_wide<JQ>(a - b);
    si = kr_row_sum_wide<JQ>(c + d);
}

// =============================================================================================
template <int NP, int NR, int LMAX, bool UPDATE>
__global__ __launch_bounds__((KrGeomR<NP, NR, LMAX>::THREADS), (KrGeomR<NP, NR, LMAX>::THREADS / 256)) void kr_rowsplit_sweep(const KrSweepArgs a) {
    using G = KrGeomR<NP, NR, LMAX>;
    constexpr int JS = G::JS, JSR = G::JSR, JSL = G::JSL, JQ = G::JQ, R = G::R, NPP = G::NPP;
    __shared__ __attribute__((aligned(16))) double2 sHl[JSL > 0 ? JSL * G::THREADS : 1];  // LDS part of the H(eps) block
    __shared__ __attribute__((aligned(16))) double2 sx[2][NPP];   // full state vector psi (or chi), by step parity
    __shared__ __attribute__((aligned(16))) double2 sy[2][NPP];   // Horner vectors, ping-pong
    constexpr int XL = (LMAX < 2) ? 2 : LMAX;   // slot width of the per-step exchange (kr_exchange: 2 or 4 doubles)
    __shared__ double spart[XL * NR];
    __shared__ double stot[XL + 1];
    __shared__ int sflag[16];   // [parity][polling wave] (kr_group_gather)

    // Workgroup -> (trajectory tile, row block).  With a.xcd_group (K % 8 == 0) the R workgroups of a tile have equal
    // blockIdx % 8: one XCD under the dispatcher's round-robin placement, which is CHECKED below through HW_REG_XCC_ID;
    // only then does the all-gather stay inside that XCD's L2 (speed; the other path is correct for any placement).
    const int tid = threadIdx.x, lane = tid & 63;
    const int tiles = a.P / R;
    const int wg = a.xcd_group ? (((int)(blockIdx.x & 7) * (tiles >> 3) + (int)(blockIdx.x >> 3) / R) * R + (int)(blockIdx.x >> 3) % R) : (int)blockIdx.x;
    const int tile = wg / R, rb = wg % R;
    const int il = tid / JQ, jq = tid % JQ;
    const int row = rb * NR + il;
    const bool owner = (jq == 0), row_ok = (row < a.N);
    const double my_rinv = kr_rinv[lane <= KR_MAX_ORDER ? lane : 0];
    const double my_thr = kr_load_my_thr(a);
    const KrGen<LMAX> gx = kr_gen_ctx<LMAX>(a, a.wg_gen ? a.wg_gen[tile] : 0);
    const int xoff = jq * (JS + 1);
    const int N = a.N, NT = a.NT, L = a.L;
    const double *econ_rows = a.econ ? a.econ + ((size_t)gx.gen * NT) * KR_ECON_W + (lane & (KR_ECON_W - 1)) : nullptr;
    const size_t slab = (size_t)a.Kp * N;
    const size_t kbase = (size_t)tile * N;

    // generator blocks are read through a buffer resource (raw buffer loads, see kr_rs_stream)
    const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)gx.Hpack, 0, (int)(a.pack_stride * 16), 0x00020000);
    KrGroupBox box;
    box.base = a.gmail + (size_t)tile * 2 * NP * 4;
    box.tag_base = a.epoch_base2;
    box.counter = 0;
    box.local = false; box.tpoll = 0; box.npass = 0;
    if (a.xcd_group) {
        // every workgroup posts {launch tag | XCC id}; wave 0 reads its R group members: local iff all ids agree
        unsigned long long *xtab = a.gmail + (size_t)tiles * 2 * NP * 4;
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
                if (kr_realtime() - t0 > a.spin_ticks / 64) break;  // never arrived: stay on the placement-independent path
            }
            if (lane == 0) sflag[0] = same ? 1 : 0;
        }
        __syncthreads();
        box.local = (sflag[0] != 0);
        __syncthreads();
    }

    // full initial vector into LDS, own element into registers
    for (int e = tid; e < NP; e += G::THREADS) {
        double2 v = make_double2(0.0, 0.0);
        if (e < N) v = a.init[kbase + e];
        sx[0][e + e / JS] = v;
    }
    double pr = 0.0, pim = 0.0, cr = 0.0, ci = 0.0;
    if (row_ok) {
        const double2 v = a.init[kbase + row];
        pr = v.x; pim = v.y;
        if (UPDATE) { const double2 c = a.X[kbase + row]; cr = c.x; ci = c.y; }
        if (a.store && owner) a.store[(size_t)((a.dir > 0) ? 0 : NT) * slab + kbase + row] = v;
    }
    double gint[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) gint[l] = 0.0;
    __syncthreads();

    const bool stamp = (a.debug != nullptr) && (wg == 0);
    unsigned long long tS = 0, tD = 0, tG = 0, tX = 0, t_prev = 0, nG = 0;
    if (stamp) t_prev = __builtin_amdgcn_s_memtime();
#define KR_STAMP(acc) if (stamp) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - t_prev; t_prev = t_; }
    double Er[JSR], Ei[JSR];
    double ecur[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) ecur[l] = 0.0;
    int cur = 0;
    for (int step = 0; step < NT; ++step) {
        const int n = (a.dir > 0) ? step : (NT - 1 - step);
        const double dtn = a.dt[n];
        const double sdt = (a.dir > 0) ? dtn : -dtn;
        // e / eg / de / ecur are the AMPLITUDES the control blocks are combined with.  Linear controls: the pulses themselves.  Non-linear
        // controls (UPDATE with KrSweepArgs::poly, round 5; kr_sweep_update<.., POLY> in kr_device.h): the host hands over a.eps = f_l(eps_guess)
        // as it does for the known-pulse sweeps; the guess PULSES the update is added to ride in a.amp_g (sic: the row-split kernels' use of
        // that field) and f_l'(eps_guess) in a.fprime_g, both fetched behind the exchange -- nothing more is live across it, the linear path
        // is the code it was.
        double e[LMAX], eg[LMAX], de[LMAX];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) { eg[l] = (l < L) ? a.eps[(size_t)l * NT + n] : 0.0; e[l] = eg[l]; de[l] = 0.0; }
        // row of the step's economized polynomial (lane j: entry j; kr_device.h: KrEcon), consumed behind the exchange; 0 = Taylor's plan
        const double ec = econ_rows ? econ_rows[(size_t)n * KR_ECON_W] : 0.0;
        double orr[1 + LMAX], oi[1 + LMAX];
        const bool refresh = (step % KR_RS_REFRESH) == 0;   // rebuild H(eps) from H0; otherwise add (eps - ecur) H_l
        double dinc[LMAX];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) dinc[l] = eg[l] - ecur[l];
        if (!UPDATE) {
            if (refresh) kr_rs_stream<NP, NR, LMAX, 0>(hrsrc, rb, L, eg, Er, Ei, sHl, nullptr, orr, oi);
            else kr_rs_stream<NP, NR, LMAX, 2>(hrsrc, rb, L, dinc, Er, Ei, sHl, nullptr, orr, oi);
            KR_STAMP(tS)
        } else {
            // phase A: H(eps_guess) block into registers, b = H(eps_g) psi, a_l = H_l psi for the own rows
            if (refresh) kr_rs_stream<NP, NR, LMAX, 1>(hrsrc, rb, L, eg, Er, Ei, sHl, &sx[cur][xoff], orr, oi);
            else kr_rs_stream<NP, NR, LMAX, 1, true>(hrsrc, rb, L, dinc, Er, Ei, sHl, &sx[cur][xoff], orr, oi);
            KR_STAMP(tS)
            double part[XL], du[XL];
#pragma unroll
            for (int l = 0; l < XL; ++l) {
                part[l] = 0.0;
                if (l < LMAX && l < L && ((gx.present >> l) & 1u)) part[l] = cr * oi[1 + (l < LMAX ? l : 0)] - ci * orr[1 + (l < LMAX ? l : 0)];
            }
            if (n + 1 < NT && row_ok) { const double2 c = a.X[(size_t)(n + 1) * slab + kbase + row]; cr = c.x; ci = c.y; }
            if (!kr_exchange<XL, NR, KR_RS_SCHED>(a, n, wg, owner, il, part, du, spart, stot)) return;
            if (tid == 128 && a.dulog) {   // the totals this workgroup gathered, for kr_pulse_finish's comparison (kr_sweep_update, kr_device.h)
#pragma unroll
                for (int l = 0; l < LMAX; ++l) if (l < L) a.dulog[((size_t)wg * NT + n) * KR_LMAX_CAP + l] = (l == 0 && a.rs_fault == 1000 + wg && n == 1) ? -du[l] : du[l];
            }
            KR_STAMP(tX)
#pragma unroll
            for (int l = 0; l < LMAX; ++l) {
                if (l < L) {
                    const double alpha = a.alpha[(size_t)l * NT + n];
                    double dul = du[l], pn;
                    if (a.poly) {   // mu_l = f_l'(eps_guess) H_l (optimize.jl:337,345); the amplitude of the updated pulse: f_l(pg + d) as the cubic's Taylor form at pg
                        const double pg = a.amp_g[(size_t)l * NT + n], fp = a.fprime_g[(size_t)l * NT + n], c2 = a.poly[4 * l + 2], c3 = a.poly[4 * l + 3];
                        dul = __dmul_rn(fp, du[l]);
                        pn = __dadd_rn(pg, __dmul_rn(alpha, dul));
                        e[l] = eg[l] + kr_poly3_delta(pn - pg, fp, fma(3.0 * c3, pg, c2), c3);
                    } else {
                        pn = __dadd_rn(eg[l], __dmul_rn(alpha, dul));
                        e[l] = pn;
                    }
                    de[l] = e[l] - eg[l];
                    const double ad = fabs(dul);
                    gint[l] = __dadd_rn(gint[l], __dmul_rn(__dmul_rn(alpha, __dmul_rn(ad, ad)), dtn));
                    if (wg == 0 && tid == 0) a.eps_new[(size_t)l * NT + n] = pn;
                }
            }
            // H(eps_new) = H(eps_g) + sum_l d eps_l H_l  (second streamed pass over the control blocks only)
            kr_rs_stream<NP, NR, LMAX, 2>(hrsrc, rb, L, de, Er, Ei, sHl, nullptr, orr, oi);
            KR_STAMP(tS)
        }
#pragma unroll
        for (int l = 0; l < LMAX; ++l) ecur[l] = e[l];   // the pulse now folded into the resident H(eps) block
        double theta = gx.nrm[0];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) theta += fabs(e[l]) * gx.nrm[1 + l];
        theta *= fabs(sdt);
        int s, m;
        kr_plan_fast(a, my_thr, theta, s, m);
        int se = s, me = m;
        KrEcon eco = kr_econ_select(ec, theta, se, me);
        if (eco.on && (s > 1 || me < m)) { s = se; m = me; }          // (a row whose degree saves nothing is not used)
        else { eco.on = false; eco.a0 = 1.0; eco.a1 = 1.0; }
        const double my_g = eco.on ? ec : my_rinv;   // Horner ratios of this step, lane j holds g_j
        const double fs = (s == 1) ? sdt : sdt / (double)s;

        for (int sub = 0; sub < s; ++sub) {
            // Horner: y_m = x0 ... ; with the phase-A product known (update sweep, first sub-step) the constant is
            // v = -i fs H psi and the result is psi + y_1; otherwise x0 = psi and the result is y_0.
            const bool have_v = UPDATE && (sub == 0);
            double x0r, x0i;
            int w = 0, jlast;
            const double2 *src;
            if (have_v) {
                double wr = orr[0], wi = oi[0];
#pragma unroll
                for (int l = 0; l < LMAX; ++l) { wr = fma(de[l], orr[1 + l], wr); wi = fma(de[l], oi[1 + l], wi); }
                x0r = fs * wi; x0i = -fs * wr;
                if (m > 1) { if (!kr_group_gather<NP, NR, JS>(a, box, rb, owner, il, x0r, x0i, sy[0], sflag)) return; }
                w = 1; jlast = 2; src = sy[0];
            } else {
                x0r = pr; x0i = pim; jlast = 1; src = sx[cur];
            }
            double yr = x0r, yi = x0i;
            for (int j = m; j >= jlast; --j) {
                const bool last = (j == jlast);
                // (known-pulse form, last round: a_0 psi + a_1 z y_1 -- entry 1 of the row is a_1 itself, not a ratio; Taylor: 1, 1)
                const bool scale0 = !have_v && last;
                const double f = scale0 ? fs * eco.a1 : fs * kr_rinv_lane(my_g, j);
                double sr, si;
                kr_rs_row_dot<JS, JSR, JQ, (JS >= 8 ? 8 : JS), G::THREADS>(Er, Ei, sHl, src + xoff, sr, si);
                yr = fma(f, si, scale0 ? eco.a0 * x0r : x0r);
                yi = fma(-f, sr, scale0 ? eco.a0 * x0i : x0i);
                if (last && have_v) { yr = fma(eco.a1, yr, eco.a0 * pr); yi = fma(eco.a1, yi, eco.a0 * pim); }   // psi' = a_0 psi + a_1 y_1  (Taylor: 1, 1)
                double2 *dst = last ? sx[cur ^ 1] : sy[w];
                KR_STAMP(tD)
                if (!kr_group_gather<NP, NR, JS>(a, box, rb, owner, il, yr, yi, dst, sflag)) return;
                KR_STAMP(tG)
                nG++;
                src = sy[w];
                w ^= 1;
            }
            if (have_v && m == 1) {  // psi' = psi + v : no round ran
                yr = fma(eco.a1, x0r, eco.a0 * pr); yi = fma(eco.a1, x0i, eco.a0 * pim);
                if (!kr_group_gather<NP, NR, JS>(a, box, rb, owner, il, yr, yi, sx[cur ^ 1], sflag)) return;
            }
            pr = yr; pim = yi;
            cur ^= 1;
        }
        if (a.store && owner && row_ok) {
            const int slot = (a.dir > 0) ? (n + 1) : n;
            a.store[(size_t)slot * slab + kbase + row] = make_double2(pr, pim);
        }
    }
    if (stamp && tid == 0) { a.debug[0] = tS; a.debug[1] = tD; a.debug[2] = tG; a.debug[3] = tX; a.debug[7] = nG; a.debug[5] = box.tpoll; a.debug[6] = box.npass; }
#undef KR_STAMP
    if constexpr (R > 1) {
        // Coherence word of the group (round 4): the R workgroups of a trajectory must end the sweep with bitwise identical copies
        // of its state vector -- every element they hold went through the all-gather.  Each posts a position-sensitive checksum of
        // its LDS copy behind the mailboxes (the XCC-id table's slots: their job ended at kernel start); kr_rowsplit_check
        // (krotov_hip.hip) compares them after the launch and turns a difference into a status instead of a silently wrong sweep.
        __shared__ unsigned long long scs[G::THREADS / 64];
        if (a.rs_fault && wg == a.rs_fault && tid == 0) { sx[cur][0].x = __longlong_as_double(__double_as_longlong(sx[cur][0].x) ^ 2ll); }   // (tests: a fault to find)
        if (a.rs_fault) __syncthreads();
        unsigned long long cs = 0ull;
        for (int e = tid; e < NP; e += G::THREADS) {
            const double2 v = sx[cur][e + e / JS];
            const unsigned long long bx = (unsigned long long)__double_as_longlong(v.x), by = (unsigned long long)__double_as_longlong(v.y);
            const int r1 = e & 63, r2 = (e * 7 + 13) & 63;
            cs ^= ((bx << r1) | (r1 ? bx >> (64 - r1) : 0ull)) ^ ((by << r2) | (r2 ? by >> (64 - r2) : 0ull));
        }
        for (int o = 1; o < 64; o <<= 1) cs ^= __shfl_xor(cs, o, 64);
        if (lane == 0) scs[tid >> 6] = cs;
        __syncthreads();
        if (tid == 0) {
            unsigned long long all = 0x9E3779B97F4A7C15ull;   // (never zero by accident: a zero slot means "not posted")
            for (int w = 0; w < G::THREADS / 64; ++w) all = (all << 1 | all >> 63) ^ scs[w];
            __hip_atomic_store((kr_gu64 *)(a.gmail + (size_t)tiles * 2 * NP * 4) + wg, all | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (a.fin && owner && row_ok) a.fin[kbase + row] = make_double2(pr, pim);
    if (UPDATE && wg == 0 && tid == 0) {
#pragma unroll
        for (int l = 0; l < LMAX; ++l) if (l < L) a.g_a_int[l] = gint[l];
    }
}

// =============================================================================================
// Update sweep IN TURNS (round 3; NP >= 512, and NP = 256 with two controls): more trajectories than groups of R workgroups -- every group carries a.rs_sets
// trajectories, set s = trajectory s * groups + group.  The H(eps) block is resident once and serves every set (shared generator).
// Per step: phase A of every set in turn (its vector from the workgroup's global scratch into LDS, one streamed pass over the
// control blocks for b = H(eps_g) psi and a_l = H_l psi, the partial sums accumulate), ONE exchange, the correction pass to
// H(eps_new), then the Horner rounds of every set in turn (vector from the scratch again, the phase-A products of its rows from LDS,
// its own all-gather mailbox and counter), psi(t_n+1) back into the scratch.  The known-pulse sweeps of such a handle are batches of
// kr_rowsplit_sweep launches (krotov_hip.hip).  A separate kernel on purpose: kr_rowsplit_sweep's code generation does not move.
template <int NP, int NR, int LMAX>
__global__ __launch_bounds__((KrGeomR<NP, NR, LMAX>::THREADS), (KrGeomR<NP, NR, LMAX>::THREADS / 256)) void kr_rowsplit_turns_update(const KrSweepArgs a) {
    using G = KrGeomR<NP, NR, LMAX>;
    constexpr int JS = G::JS, JSR = G::JSR, JQ = G::JQ, R = G::R, NPP = G::NPP, T = G::THREADS;
    static_assert(G::JSL == 0, "the H(eps) block in registers");
    __shared__ __attribute__((aligned(16))) double2 sx[2][NPP];
    __shared__ __attribute__((aligned(16))) double2 sy[2][NPP];
    constexpr int XL = (LMAX < 2) ? 2 : LMAX;
    __shared__ double spart[XL * NR];
    __shared__ double stot[XL + 1];
    __shared__ int sflag[16];
    __shared__ double sab[KR_RS_MAX_TURNS][2 * (1 + LMAX)][NR];   // phase-A products b, a_l of the rows, per set
    __shared__ unsigned scnt[KR_RS_MAX_TURNS];                    // all-gather counters of the sets' trajectories
    double2 *sHl = nullptr;

    const int tid = threadIdx.x, lane = tid & 63;
    const int S = a.rs_sets;
    const int tiles = a.P / R;   // groups
    const int wg = a.xcd_group ? (((int)(blockIdx.x & 7) * (tiles >> 3) + (int)(blockIdx.x >> 3) / R) * R + (int)(blockIdx.x >> 3) % R) : (int)blockIdx.x;
    const int tile = wg / R, rb = wg % R;
    const int il = tid / JQ, jq = tid % JQ;
    const int row = rb * NR + il;
    const bool owner = (jq == 0), row_ok = (row < a.N);
    const double my_rinv = kr_rinv[lane <= KR_MAX_ORDER ? lane : 0];
    const double my_thr = kr_load_my_thr(a);
    const KrGen<LMAX> gx = kr_gen_ctx<LMAX>(a, 0);
    const int xoff = jq * (JS + 1);
    const int N = a.N, NT = a.NT, L = a.L;
    const double *econ_rows = a.econ ? a.econ + (lane & (KR_ECON_W - 1)) : nullptr;
    const size_t slab = (size_t)a.Kp * N;
    double2 *vscr = a.rs_scratch + (size_t)wg * KR_RS_MAX_TURNS * NP;   // this workgroup's copies of the sets' state vectors

    const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)gx.Hpack, 0, (int)(a.pack_stride * 16), 0x00020000);
    KrGroupBox box;
    box.base = a.gmail;
    box.tag_base = a.epoch_base2;
    box.counter = 0;
    box.local = false; box.tpoll = 0; box.npass = 0;
    if (a.xcd_group) {
        unsigned long long *xtab = a.gmail + (size_t)tiles * S * 2 * NP * 4;
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
        box.local = (sflag[0] != 0);
        __syncthreads();
    }
    if (tid < KR_RS_MAX_TURNS) scnt[tid] = 0u;
    unsigned long long coh = 0ull;   // this thread's share of the group's coherence word (posted at the end)
    double gint[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) gint[l] = 0.0;
    __syncthreads();

    double Er[JSR], Ei[JSR];
    double ecur[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) ecur[l] = 0.0;
    // the state vector of set s at the start of step n: the initial states at n = 0, the workgroup's scratch afterwards
    auto load_set = [&](const int s, const int n, const size_t kbase, double &pr, double &pim) {
        const double2 *srcv = (n == 0) ? a.init + kbase : vscr + (size_t)s * NP;
        for (int e = tid; e < NP; e += T) {
            double2 v = make_double2(0.0, 0.0);
            if (e < N) v = srcv[e];
            sx[0][e + e / JS] = v;
        }
        pr = 0.0; pim = 0.0;
        if (row_ok) { const double2 v = srcv[row]; pr = v.x; pim = v.y; }
    };
    for (int n = 0; n < NT; ++n) {
        const double dtn = a.dt[n];
        double e[LMAX], eg[LMAX], de[LMAX], dinc[LMAX], zero[LMAX];   // (amplitudes; non-linear controls as in kr_rowsplit_sweep)
#pragma unroll
        for (int l = 0; l < LMAX; ++l) { eg[l] = (l < L) ? a.eps[(size_t)l * NT + n] : 0.0; e[l] = eg[l]; de[l] = 0.0; dinc[l] = eg[l] - ecur[l]; zero[l] = 0.0; }
        const double ec = econ_rows ? econ_rows[(size_t)n * KR_ECON_W] : 0.0;
        const bool refresh = (n % KR_RS_REFRESH) == 0;
        double part[XL], du[XL];
#pragma unroll
        for (int l = 0; l < XL; ++l) part[l] = 0.0;
        // ---- phase A of every set
        for (int s = 0; s < S; ++s) {
            const size_t kbase = ((size_t)s * tiles + tile) * N;
            double pr, pim;
            load_set(s, n, kbase, pr, pim);
            double cr = 0.0, ci = 0.0;
            if (row_ok) { const double2 c = a.X[(size_t)n * slab + kbase + row]; cr = c.x; ci = c.y; }
            if (n == 0 && a.store && owner && row_ok) a.store[kbase + row] = make_double2(pr, pim);
            __syncthreads();
            double orr[1 + LMAX], oi[1 + LMAX];
            if (s == 0) {
                if (refresh) kr_rs_stream<NP, NR, LMAX, 1>(hrsrc, rb, L, eg, Er, Ei, sHl, &sx[0][xoff], orr, oi);
                else kr_rs_stream<NP, NR, LMAX, 1, true>(hrsrc, rb, L, dinc, Er, Ei, sHl, &sx[0][xoff], orr, oi);
            } else {   // the block already is H(eps_g): add nothing, take the products
                kr_rs_stream<NP, NR, LMAX, 1, true>(hrsrc, rb, L, zero, Er, Ei, sHl, &sx[0][xoff], orr, oi);
            }
#pragma unroll
            for (int l = 0; l < LMAX; ++l)
                if (l < L && ((gx.present >> l) & 1u)) part[l] += cr * oi[1 + l] - ci * orr[1 + l];
            if (owner) {
#pragma unroll
                for (int o = 0; o <= LMAX; ++o) { sab[s][2 * o][il] = orr[o]; sab[s][2 * o + 1][il] = oi[o]; }
            }
            __syncthreads();   // (the next set overwrites sx[0])
        }
        if (!kr_exchange<XL, NR, KR_RS_SCHED>(a, n, wg, owner, il, part, du, spart, stot)) return;
        if (tid == 128 && a.dulog) {
#pragma unroll
            for (int l = 0; l < LMAX; ++l) if (l < L) a.dulog[((size_t)wg * NT + n) * KR_LMAX_CAP + l] = (l == 0 && a.rs_fault == 1000 + wg && n == 1) ? -du[l] : du[l];
        }
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            if (l < L) {
                const double alpha = a.alpha[(size_t)l * NT + n];
                double dul = du[l], pn;
                if (a.poly) {
                    const double pg = a.amp_g[(size_t)l * NT + n], fp = a.fprime_g[(size_t)l * NT + n], c2 = a.poly[4 * l + 2], c3 = a.poly[4 * l + 3];
                    dul = __dmul_rn(fp, du[l]);
                    pn = __dadd_rn(pg, __dmul_rn(alpha, dul));
                    e[l] = eg[l] + kr_poly3_delta(pn - pg, fp, fma(3.0 * c3, pg, c2), c3);
                } else {
                    pn = __dadd_rn(eg[l], __dmul_rn(alpha, dul));
                    e[l] = pn;
                }
                de[l] = e[l] - eg[l];
                const double ad = fabs(dul);
                gint[l] = __dadd_rn(gint[l], __dmul_rn(__dmul_rn(alpha, __dmul_rn(ad, ad)), dtn));
                if (wg == 0 && tid == 0) a.eps_new[(size_t)l * NT + n] = pn;
            }
        }
        {   // H(eps_new) = H(eps_g) + sum_l d eps_l H_l
            double orr[1 + LMAX], oi[1 + LMAX];
            kr_rs_stream<NP, NR, LMAX, 2>(hrsrc, rb, L, de, Er, Ei, sHl, nullptr, orr, oi);
        }
#pragma unroll
        for (int l = 0; l < LMAX; ++l) ecur[l] = e[l];
        double theta = gx.nrm[0];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) theta += fabs(e[l]) * gx.nrm[1 + l];
        theta *= fabs(dtn);
        int sn, m;
        kr_plan_fast(a, my_thr, theta, sn, m);
        int se = sn, me = m;
        KrEcon eco = kr_econ_select(ec, theta, se, me);
        if (eco.on && (sn > 1 || me < m)) { sn = se; m = me; }
        else { eco.on = false; eco.a0 = 1.0; eco.a1 = 1.0; }
        const double my_g = eco.on ? ec : my_rinv;
        const double fs = (sn == 1) ? dtn : dtn / (double)sn;
        // ---- Horner rounds of every set
        for (int s = 0; s < S; ++s) {
            const size_t traj = (size_t)s * tiles + tile;
            const size_t kbase = traj * N;
            box.base = a.gmail + traj * 2 * NP * 4;
            box.counter = scnt[s];
            double pr, pim;
            load_set(s, n, kbase, pr, pim);
            double orr[1 + LMAX], oi[1 + LMAX];
#pragma unroll
            for (int o = 0; o <= LMAX; ++o) { orr[o] = sab[s][2 * o][il]; oi[o] = sab[s][2 * o + 1][il]; }
            __syncthreads();
            int cur = 0;
            for (int sub = 0; sub < sn; ++sub) {
                const bool have_v = (sub == 0);
                double x0r, x0i;
                int w = 0, jlast;
                const double2 *src;
                if (have_v) {
                    double wr = orr[0], wi = oi[0];
#pragma unroll
                    for (int l = 0; l < LMAX; ++l) { wr = fma(de[l], orr[1 + l], wr); wi = fma(de[l], oi[1 + l], wi); }
                    x0r = fs * wi; x0i = -fs * wr;
                    if (m > 1) { if (!kr_group_gather<NP, NR, JS>(a, box, rb, owner, il, x0r, x0i, sy[0], sflag)) return; }
                    w = 1; jlast = 2; src = sy[0];
                } else {
                    x0r = pr; x0i = pim; jlast = 1; src = sx[cur];
                }
                double yr = x0r, yi = x0i;
                for (int j = m; j >= jlast; --j) {
                    const bool last = (j == jlast);
                    const bool scale0 = !have_v && last;
                    const double f = scale0 ? fs * eco.a1 : fs * kr_rinv_lane(my_g, j);
                    double sr, si;
                    kr_rs_row_dot<JS, JSR, JQ, (JS >= 8 ? 8 : JS), T>(Er, Ei, sHl, src + xoff, sr, si);
                    yr = fma(f, si, scale0 ? eco.a0 * x0r : x0r);
                    yi = fma(-f, sr, scale0 ? eco.a0 * x0i : x0i);
                    if (last && have_v) { yr = fma(eco.a1, yr, eco.a0 * pr); yi = fma(eco.a1, yi, eco.a0 * pim); }
                    double2 *dst = last ? sx[cur ^ 1] : sy[w];
                    if (!kr_group_gather<NP, NR, JS>(a, box, rb, owner, il, yr, yi, dst, sflag)) return;
                    src = sy[w];
                    w ^= 1;
                }
                if (have_v && m == 1) {
                    yr = fma(eco.a1, x0r, eco.a0 * pr); yi = fma(eco.a1, x0i, eco.a0 * pim);
                    if (!kr_group_gather<NP, NR, JS>(a, box, rb, owner, il, yr, yi, sx[cur ^ 1], sflag)) return;
                }
                pr = yr; pim = yi;
                cur ^= 1;
            }
            if (a.store && owner && row_ok) a.store[(size_t)(n + 1) * slab + kbase + row] = make_double2(pr, pim);
            for (int e = tid; e < NP; e += T) if (e < N) vscr[(size_t)s * NP + e] = sx[cur][e + e / JS];
            if (n == NT - 1) {   // coherence word of the group (see kr_rowsplit_sweep): the final vector of every set this group carries
                for (int e = tid; e < NP; e += T) {
                    const double2 v = sx[cur][e + e / JS];
                    const unsigned long long bx = (unsigned long long)__double_as_longlong(v.x), by = (unsigned long long)__double_as_longlong(v.y);
                    const int i = s * NP + e, r1 = i & 63, r2 = (i * 7 + 13) & 63;
                    coh ^= ((bx << r1) | (r1 ? bx >> (64 - r1) : 0ull)) ^ ((by << r2) | (r2 ? by >> (64 - r2) : 0ull));
                }
            }
            if (tid == 0) scnt[s] = box.counter;
            if (n == NT - 1 && a.fin && owner && row_ok) a.fin[kbase + row] = make_double2(pr, pim);
            // The scratch vector is written by all threads here and read back by OTHER threads of the same workgroup in load_set
            // (twice per step).  Workgroup-scope release / acquire around the barrier, written out: the waves of a workgroup share
            // one CU and its vector L1 (no threadgroup-split mode), so a workgroup-scope fence is what this hand-over needs --
            // unlike the cross-workgroup mailboxes, which go through sc0 / sc1 accesses.
            __threadfence_block();
            __syncthreads();
            __threadfence_block();
        }
    }
    if (wg == 0 && tid == 0) {
#pragma unroll
        for (int l = 0; l < LMAX; ++l) if (l < L) a.g_a_int[l] = gint[l];
    }
    {
        __shared__ unsigned long long scs[T / 64];
        if (a.rs_fault && wg == a.rs_fault && tid == 0) coh ^= 2ull;   // (tests: a fault to find)
        for (int o = 1; o < 64; o <<= 1) coh ^= __shfl_xor(coh, o, 64);
        if (lane == 0) scs[tid >> 6] = coh;
        __syncthreads();
        if (tid == 0) {
            unsigned long long all = 0x9E3779B97F4A7C15ull;
            for (int w = 0; w < T / 64; ++w) all = (all << 1 | all >> 63) ^ scs[w];
            __hip_atomic_store((kr_gu64 *)(a.gmail + (size_t)tiles * S * 2 * NP * 4) + wg, all | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// kind: 0 known sweep, 1 update sweep; nr = rows per workgroup
template <int NP>
static hipError_t kr_launch_rowsplit_np(int lmax, int nr, int kind, int grid, hipStream_t s, const KrSweepArgs *a) {
#define KR_CASE_R(LM, NRv)                                                                             \
    if (lmax == LM && nr == NRv) {                                                                     \
        if (kind == 0) kr_launch_kernel((kr_rowsplit_sweep<NP, NRv, LM, false>), grid, KrGeomR<NP, NRv, LM>::THREADS, 0, s, *a); \
        else kr_launch_kernel((kr_rowsplit_sweep<NP, NRv, LM, true>), grid, KrGeomR<NP, NRv, LM>::THREADS, 0, s, *a);             \
        return hipGetLastError();                                                                      \
    }
    if constexpr (NP >= 256) {   // kind 5: the update sweep in turns (kr_rowsplit_turns_update), one or two controls
        if (kind == 5) {
            if (a->rs_sets < 1 || a->rs_sets > KR_RS_MAX_TURNS || a->rs_scratch == nullptr) return hipErrorInvalidValue;
#define KR_CASE_T(LM, NRv)                                                                             \
            if (lmax == LM && nr == NRv) {                                                             \
                kr_launch_kernel((kr_rowsplit_turns_update<NP, NRv, LM>), grid, KrGeomR<NP, NRv, LM>::THREADS, 0, s, *a); \
                return hipGetLastError();                                                              \
            }
            if constexpr (NP == 256) { KR_CASE_T(2, 32) }   // (one control at N <= 256: pair mode with sets)
            else { KR_CASE_T(1, 8) KR_CASE_T(2, 8) }
            if constexpr (NP == 512) { KR_CASE_T(4, 8) }
            if constexpr (NP == 512) { KR_CASE_T(1, 16) KR_CASE_T(2, 16) }
#undef KR_CASE_T
            return hipErrorInvalidValue;
        }
    }
    if constexpr (NP == 128) { KR_CASE_R(1, 128) KR_CASE_R(2, 128) }   // R = 1: one workgroup per trajectory (L <= 2: LMAX = 4 spills)
    if constexpr (NP == 1024) {  // 512 < N <= 1024: 8 rows per workgroup (128 workgroups per trajectory), one or two controls
        KR_CASE_R(1, 8) KR_CASE_R(2, 8)
    } else if constexpr (NP == 512) {   // 256 < N <= 512: 8 or 16 rows per workgroup, one or two controls
        KR_CASE_R(1, 8) KR_CASE_R(1, 16)     // (NR = 32: the LDS half of the block + the vectors exceed 160 KB)
        KR_CASE_R(2, 8) KR_CASE_R(2, 16)
        KR_CASE_R(4, 8)
    } else {
    KR_CASE_R(1, 8) KR_CASE_R(1, 16) KR_CASE_R(1, 32) KR_CASE_R(1, 64)   // one control: C5's shape
    KR_CASE_R(2, 8) KR_CASE_R(2, 16) KR_CASE_R(2, 32) KR_CASE_R(2, 64)
    KR_CASE_R(4, 8) KR_CASE_R(4, 16) KR_CASE_R(4, 32) KR_CASE_R(4, 64)
    }
#undef KR_CASE_R
    return hipErrorInvalidValue;
}
