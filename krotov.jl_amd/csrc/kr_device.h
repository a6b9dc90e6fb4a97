// Device code of the Krotov sweeps for gfx950 (CDNA4).  Register-resident-generator variant,
// state dimension padded to NP in {8,16,32,64}.
//
// Mapping (one persistent workgroup per CU over all N_T time steps; 256 threads, 512 for NP = 64):
//   team  = JQ*NP threads that own ONE copy of the generator slices in registers (JQ = 4; JQ = 8 for NP = 64):
//           thread (i, jq) holds row i, columns [jq*JS, (jq+1)*JS), JS = NP/JQ, of every matrix
//           (for NP = 64 and L <= 2 the drift slice H0 lives in LDS instead);
//   a workgroup has SG = THREADS/(JQ*NP) teams; each team propagates KT trajectories, so one workgroup
//   carries KTILE = SG*KT trajectories and the generator is read from HBM/L2 ONCE per sweep, not per step.
//   A matvec round y = H x issues all JS 16-B LDS reads of the x slice back to back (broadcast across the
//   rows of a wave, padded so the JQ column segments fall on different banks), does 4*JS fp64 FMAs per thread
//   in four independent chains, and sums the JQ column segments of a row inside the lane group with DPP
//   (no LDS, no wave-wide shuffle); one LDS write per row and one s_barrier end the round.
//   In the update sweep the step propagator exp(-i H(eps_n) dt) is never formed: it is applied as a
//   Horner-form Taylor polynomial of order m (chosen per step from ||H(eps_n)|| dt so that the truncation
//   bound is below tol), with H(eps_n) = H0 + sum_l eps_l[n] H_l combined in registers.  The known-pulse
//   sweeps normally run on precomputed propagators instead (kr_device_prop.h); kr_sweep_known is their
//   Taylor form (KROTOV_HIP_PRECOMP=0, or when the propagator buffer would be too large).
//
// Reference semantics restated (paths under /root/reference):
//   kr_sweep_known  : optimize.jl:247-265 (initial forward sweep) and :303-317 (backward sweep
//                     through the adjoint generator, workspace.jl:69,149-161)
//   kr_sweep_update : optimize.jl:321-370 -- per step: du_l = sum_k Im<chi_k|H_l|psi_k> (:339-349),
//                     eps_new = eps + S/lambda du, g_a_int += S/lambda |du|^2 dt (:351-358), then
//                     psi_k <- U_n(eps_new) psi_k (:360-368).
#pragma once
#include "kr_common.h"
#include <cstdlib>

template <int NP, int LMAX, int KT>
struct KrGeom {
    // NP = 64: 512-thread teams (2 waves/SIMD, JQ = 8: measured 795 vs 966 cycles per matvec round against
    // 256 threads, tools/ubench/round.hip).  For L <= 2 the drift slice H0 additionally lives in LDS (64 KB,
    // read once per step when H(eps) is formed) so that the register file keeps room for all LDS reads of a
    // round in flight; for L > 2 all 1+L+1 slices stay in registers.
#ifdef KR_NO_H0LDS
    static constexpr bool H0LDS = false;
#else
    static constexpr bool H0LDS = (NP >= 64) && (LMAX <= 2);
#endif
    static constexpr int THREADS = (NP >= 64) ? 512 : 256;
    static constexpr int JQ = (THREADS == 512) ? 8 : 4;  // lanes that share one row
    static constexpr int NW = THREADS / 64;
    static constexpr int JS = NP / JQ;
    static constexpr int RT = NP * JQ;
    static constexpr int SG = THREADS / RT;
    static constexpr int KTILE = SG * KT;
    static constexpr int NPP = NP + JQ;  // one 16-B pad slot per column segment
    static_assert(NP >= 8 && NP <= 64 && (NP & (NP - 1)) == 0, "NP in {8,16,32,64}");
};

// Synchronisation between the Horner rounds of one team.  A team of RT = NP * JQ threads that fits into one wavefront
// (NP <= 16) only exchanges its vector through LDS with ITSELF: LDS instructions of a wave execute in order, so the
// reads of the next round see the writes of this one without a workgroup barrier (204 cycles per round saved; the
// fence only stops the compiler from moving LDS accesses across this point).  Larger teams span waves: s_barrier.
template <int RT>
__device__ __forceinline__ void kr_team_sync() {
    if (RT <= 64) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

// ---- lane-quad sum (lanes 4r..4r+3 hold the four column segments of row r) -----------------
__device__ __forceinline__ double kr_dpp_quad_swap(double v, const int which) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    if (which == 1) {  // quad_perm:[1,0,3,2]
        lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true);
        hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true);
    } else {  // quad_perm:[2,3,0,1]
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true);
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true);
    }
    return __hiloint2double(hi, lo);
}
// sum over the JQ (4 or 8) consecutive lanes that share a row; every lane gets the sum
template <int JQ>
__device__ __forceinline__ double kr_row_sum(double v) {
    v += kr_dpp_quad_swap(v, 1);
    v += kr_dpp_quad_swap(v, 2);
    if (JQ == 8) {  // row_half_mirror: lane j <-> 7-j inside each group of 8
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, true);
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, true);
        v += __hiloint2double(hi, lo);
    }
    return v;
}

template <int CTRL>
__device__ __forceinline__ double kr_dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double kr_readlane_f64(const double v, const int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
// wave-wide sum in a fixed order (bitwise reproducible), result uniform in every lane:
// butterflies inside each row of 16 lanes with DPP, then the four row sums in row order.
__device__ __forceinline__ double kr_wave_sum(double v) {
    v += kr_dpp_mov<0xB1>(v);   // quad_perm:[1,0,3,2]
    v += kr_dpp_mov<0x4E>(v);   // quad_perm:[2,3,0,1]
    v += kr_dpp_mov<0x141>(v);  // row_half_mirror
    v += kr_dpp_mov<0x140>(v);  // row_mirror
    const double r0 = kr_readlane_f64(v, 0), r1 = kr_readlane_f64(v, 16);
    const double r2 = kr_readlane_f64(v, 32), r3 = kr_readlane_f64(v, 48);
    return ((r0 + r1) + r2) + r3;
}

// Two wave-wide sums at once: the same butterflies and the same order per value as kr_wave_sum (bitwise identical results), the two
// dependent chains interleaved stage by stage -- the exchange sums L = 2 partials three times per step on its critical path, and
// one chain after the other (what the compiler emits for two calls) costs twice the latency.
__device__ __forceinline__ void kr_wave_sum2(double &a, double &b) {
    double ta, tb;
    ta = kr_dpp_mov<0xB1>(a);  tb = kr_dpp_mov<0xB1>(b);  a += ta; b += tb;
    ta = kr_dpp_mov<0x4E>(a);  tb = kr_dpp_mov<0x4E>(b);  a += ta; b += tb;
    ta = kr_dpp_mov<0x141>(a); tb = kr_dpp_mov<0x141>(b); a += ta; b += tb;
    ta = kr_dpp_mov<0x140>(a); tb = kr_dpp_mov<0x140>(b); a += ta; b += tb;
    const double a0 = kr_readlane_f64(a, 0), b0 = kr_readlane_f64(b, 0), a1 = kr_readlane_f64(a, 16), b1 = kr_readlane_f64(b, 16);
    const double a2 = kr_readlane_f64(a, 32), b2 = kr_readlane_f64(b, 32), a3 = kr_readlane_f64(a, 48), b3 = kr_readlane_f64(b, 48);
    double ra = a0 + a1, rb = b0 + b1;
    ra += a2; rb += b2;
    ra += a3; rb += b3;
    a = ra; b = rb;
}
// all LMAX (2 or 4) values of an exchange; entries l >= L must hold finite numbers (zeros): pairs are summed together
template <int LMAX>
__device__ __forceinline__ void kr_wave_sum_all(double (&v)[LMAX], const int L) {
    static_assert(LMAX == 2 || LMAX == 4, "pairs");
    kr_wave_sum2(v[0], v[1]);
    if (LMAX == 4) { if (L > 2) kr_wave_sum2(v[LMAX - 2], v[LMAX - 1]); }
}

// 1/j for a wave-uniform j without a memory access: lane j of every wave keeps 1/j in `mine`
__device__ __forceinline__ double kr_rinv_lane(const double mine, const int j) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(mine), j);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(mine), j);
    return __hiloint2double(hi, lo);
}

// Economized step polynomial (round 2).  For a Hermitian generator the step exp(-i H fs) psi needs a polynomial that is accurate
// on the segment z = -i fs [-||H||, ||H||] only; truncating the Chebyshev expansion of exp(-i theta_c x) at degree m and rewriting
// it in powers of z gives real coefficients a_j(theta_c), and the Horner loop of the kernels evaluates it unchanged with the ratios
// g_j = a_j / a_{j-1} in place of Taylor's 1 / j:   psi' = a_0 psi + a_1 y_1,  y_m = v,  y_{j-1} = v + g_j z y_j,  v = z psi.
// Same truncation bound (2 sum_{k>m} |J_k(theta_c)| <= tol), one to three degrees -- i.e. matvec rounds -- fewer than Taylor
// (theta = 0.12: 9 against 10; 0.5: 12 against 14; C3's 0.067: 7 against 8).  theta_c = 1.01 x the bound of ||H(eps_guess) dt||
// of the step, known before the sweep; a step whose updated pulse leaves that margin takes the Taylor plan (always valid).
// Lane j (mod 32) of every wave holds entry j of the step's row (KR_ECON_W), fetched one step ahead.
struct KrEcon { bool on; double a0, a1; };
__device__ __forceinline__ KrEcon kr_econ_select(const double ec, const double theta, int &s, int &m) {
    KrEcon r;
    r.on = false; r.a0 = 1.0; r.a1 = 1.0;
    const int me = (int)kr_rinv_lane(ec, 31);
    if (me > 0 && theta <= kr_rinv_lane(ec, 30)) {
        r.on = true; m = me; s = 1;
        r.a0 = kr_rinv_lane(ec, 0); r.a1 = kr_rinv_lane(ec, 1);
    }
    return r;
}

__device__ __forceinline__ unsigned long long kr_realtime() { return __builtin_amdgcn_s_memrealtime(); }

typedef __attribute__((address_space(1))) unsigned long long kr_gu64;
typedef __attribute__((address_space(1))) unsigned kr_gu32;

// y_i = sum_j E_ij x_j for this thread's column segment; returns the quad-reduced row sum.
template <int JS, int JQ>
__device__ __forceinline__ void kr_row_dot(const double (&Er)[JS], const double (&Ei)[JS],
                                           const double2 *__restrict__ xp, double &sr, double &si) {
    // issue every LDS read of the slice first so that all of them are in flight behind one another
    // (LDS latency is ~100+ cycles; with only two or three reads in flight the round is latency-bound)
    double2 x[JS];
#pragma unroll
    for (int jj = 0; jj < JS; ++jj) x[jj] = xp[jj];
    if constexpr (JS <= 4) {
        // short slices (N <= 16): two accumulators, the real part's subtraction inside the FMA (negated operand).  One wave issues an
        // fp64 instruction only every ~9.5 cycles (tools/ubench/dpprate.hip), so the two combining adds of four accumulators are ~19
        // cycles of every round, and alternating the two chains keeps consecutive FMAs of a chain two issue slots apart (> the FMA
        // latency).  C2 3.32 -> 3.23 us per step; at JS = 8 (N = 32, 64) the four-accumulator form measured 1 % faster and stays.
        double a = 0, c = 0;
#pragma unroll
        for (int jj = 0; jj < JS; ++jj) {
            a = fma(Er[jj], x[jj].x, a);
            c = fma(Er[jj], x[jj].y, c);
            a = fma(-Ei[jj], x[jj].y, a);
            c = fma(Ei[jj], x[jj].x, c);
        }
        sr = kr_row_sum<JQ>(a);
        si = kr_row_sum<JQ>(c);
    } else {
        double a = 0, b = 0, c = 0, d = 0;
#pragma unroll
        for (int jj = 0; jj < JS; ++jj) {
            a = fma(Er[jj], x[jj].x, a);
            b = fma(Ei[jj], x[jj].y, b);
            c = fma(Er[jj], x[jj].y, c);
            d = fma(Ei[jj], x[jj].x, d);
        }
        sr = kr_row_sum<JQ>(a - b);
        si = kr_row_sum<JQ>(c + d);
    }
}

// =============================================================================================
// Known-pulse sweep: forward (dir=+1) or backward through the adjoint generator (dir=-1).
// No inter-workgroup communication: every workgroup runs its own trajectories to the end.
// =============================================================================================
template <int NP, int LMAX, int KT>
__global__ __launch_bounds__((KrGeom<NP, LMAX, KT>::THREADS), (KrGeom<NP, LMAX, KT>::THREADS / 256)) void kr_sweep_known(const KrSweepArgs a) {
    using G = KrGeom<NP, LMAX, KT>;
    constexpr int JS = G::JS, JQ = G::JQ, RT = G::RT, SG = G::SG, KTILE = G::KTILE, NPP = G::NPP;
    __shared__ __attribute__((aligned(16))) double2 sx[2][KTILE][NPP];
    __shared__ __attribute__((aligned(16))) double2 sy[2][KTILE][NPP];
    __shared__ __attribute__((aligned(16))) double2 sH0[G::H0LDS ? JS * RT : 1];

    const int tid = threadIdx.x, wg = blockIdx.x;
    const int team = tid / RT, tt = tid % RT, i = tt / JQ, jq = tt % JQ;
    const bool writer = (jq == 0), row_ok = (i < a.N);
    const double my_rinv = kr_rinv[(tid & 63) <= KR_MAX_ORDER ? (tid & 63) : 0];
    const double my_thr = kr_load_my_thr(a);
    const KrGen<LMAX> gx = kr_gen_ctx<LMAX>(a, a.wg_gen ? a.wg_gen[blockIdx.x / (a.gen_div > 0 ? a.gen_div : 1)] : 0);
    const int pi_ = i + i / JS;          // padded position of element i
    const int xoff = jq * (JS + 1);      // padded start of this thread's column segment
    const int N = a.N, NT = a.NT, L = a.L;

    // generator slices: H_1..H_L always in registers; H0 in registers or (H0LDS) in LDS, thread-major
    double Hr[LMAX + 1][JS], Hi[LMAX + 1][JS];
#pragma unroll
    for (int m = 0; m <= LMAX; ++m) {
#pragma unroll
        for (int jj = 0; jj < JS; ++jj) {
            double2 h = make_double2(0.0, 0.0);
            if (m <= L) h = gx.Hpack[((size_t)m * JS + jj) * RT + tt];
            if (G::H0LDS && m == 0) {
                sH0[jj * RT + tt] = h;
                Hr[m][jj] = 0.0; Hi[m][jj] = 0.0;  // dead: never read when H0LDS
            } else {
                Hr[m][jj] = h.x;
                Hi[m][jj] = h.y;
            }
        }
    }

    double pr[KT], pim[KT];
    size_t kbase[KT];
#pragma unroll
    for (int q = 0; q < KT; ++q) {
        const int k = wg * KTILE + q * SG + team;
        kbase[q] = (size_t)k * N;
        double2 v = make_double2(0.0, 0.0);
        if (row_ok) v = a.init[kbase[q] + i];
        pr[q] = v.x;
        pim[q] = v.y;
        if (writer) sx[0][q * SG + team][pi_] = v;
        if (a.store && writer && row_ok) {
            const int slot0 = (a.dir > 0) ? 0 : NT;
            a.store[(size_t)slot0 * a.Kp * N + kbase[q] + i] = v;
        }
    }
    __syncthreads();

    int cur = 0;
    // pulse values and dt of the NEXT step are fetched one step ahead (off the critical path)
    double e_nx[LMAX], dt_nx;
    {
        const int n0 = (a.dir > 0) ? 0 : (NT - 1);
        dt_nx = a.dt[n0];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) e_nx[l] = (l < L) ? a.eps[(size_t)l * NT + n0] : 0.0;
    }
    for (int step = 0; step < NT; ++step) {
        const int n = (a.dir > 0) ? step : (NT - 1 - step);
        const double sdt = (a.dir > 0) ? dt_nx : -dt_nx;
        double e[LMAX];
        double theta = gx.nrm[0];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            e[l] = e_nx[l];
            if (l < L) theta += fabs(e[l]) * gx.nrm[1 + l];
        }
        theta *= fabs(sdt);
        if (step + 1 < NT) {
            const int n1 = (a.dir > 0) ? (step + 1) : (NT - 2 - step);
            dt_nx = a.dt[n1];
#pragma unroll
            for (int l = 0; l < LMAX; ++l) if (l < L) e_nx[l] = a.eps[(size_t)l * NT + n1];
        }
        int s, m;
        kr_plan_fast(a, my_thr, theta, s, m);

        double Er[JS], Ei[JS];
#pragma unroll
        for (int jj = 0; jj < JS; ++jj) {
            double r, im;
            if (G::H0LDS) { const double2 h0 = sH0[jj * RT + tt]; r = h0.x; im = h0.y; }
            else { r = Hr[0][jj]; im = Hi[0][jj]; }
#pragma unroll
            for (int l = 0; l < LMAX; ++l) {  // slices beyond L are zero and e[l] = 0 there: no select needed
                r = fma(e[l], Hr[1 + l][jj], r);
                im = fma(e[l], Hi[1 + l][jj], im);
            }
            Er[jj] = r;
            Ei[jj] = im;
        }

        const double fs = (s == 1) ? sdt : sdt / (double)s;
        for (int sub = 0; sub < s; ++sub) {
            int w = 0;
            for (int j = m; j >= 1; --j) {  // Horner: y <- psi + (-i fs / j) H y
                const double f = fs * kr_rinv_lane(my_rinv, j);
#pragma unroll
                for (int q = 0; q < KT; ++q) {
                    const int tq = q * SG + team;
                    const double2 *src = (j == m) ? &sx[cur][tq][xoff] : &sy[w ^ 1][tq][xoff];
                    double sr, si;
                    kr_row_dot<JS, JQ>(Er, Ei, src, sr, si);
                    const double yr = fma(f, si, pr[q]);
                    const double yi = fma(-f, sr, pim[q]);
                    if (j > 1) {
                        if (writer) sy[w][tq][pi_] = make_double2(yr, yi);
                    } else {
                        pr[q] = yr;
                        pim[q] = yi;
                        if (writer) sx[cur ^ 1][tq][pi_] = make_double2(yr, yi);
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
            for (int q = 0; q < KT; ++q)
                a.store[(size_t)slot * a.Kp * N + kbase[q] + i] = make_double2(pr[q], pim[q]);
        }
    }
    if (a.fin && writer && row_ok) {
#pragma unroll
        for (int q = 0; q < KT; ++q) a.fin[kbase[q] + i] = make_double2(pr[q], pim[q]);
    }
}

// =============================================================================================
// Per-step exchange of the L partial sums between the P workgroups of this GPU (and, for
// world > 1, between GPUs through peer mailboxes over xGMI).  One hop, all-to-all: every
// workgroup publishes its L partial sums (8-byte doubles, see ONE-BIT validity below) with write-through
// stores, wave 0 of every workgroup polls all P slots with loads that bypass the vector L1 (sc1)
// and sums them in slot order, so every workgroup of every GPU obtains the bitwise identical
// total.  Slots are double-buffered by step parity.  Spins are bounded.
// Wave 0 issues NO other vector-memory operation in the update kernel (state loads/stores belong
// to the last wave): vmcnt retires in order and counts loads and stores together, so an HBM load or
// a store acknowledgement queued in front of the poll sits on the critical path of every step.
// Two schedules (template parameter SCHED; chosen per handle by calibrate_exchange in krotov_hip.hip):
//   0    wave 0 publishes, does its share of the window work, then polls;
//   1    wave 1 publishes (asm store, never waited for), wave 0 sleeps a.poll_delay x 128 cycles and
//        then polls.
// Returns false when the exchange was aborted (status[0] set).
// Where the time goes (tools/exchange_arrivals.py on a -DKR_ARRIVALS build, C3, P = 256): the workgroups publish within
// 1.1-1.2 us of each other (median 0.3-0.45 us after the first; whole XCDs lag, and a late workgroup stays late: it still
// does its 0.84 us of window work before it polls, while the others are done with theirs and complete 0.7-1.0 us after
// the LAST publish); a workgroup completes 1.65 us (median) after its own publish; poll passes: 1 (median), 2 (max).
// =============================================================================================
// ONE-BIT validity (round 2; before: slot = 2*LMAX granules {tag32 | half of a double}, 32 B for LMAX = 2).  A slot is written by
// one wave only, in step order, and a reader that has completed step n - 2 has
// seen the slot's value of step n - 2 (same buffer: parity of n), so at step n it can only find that value or the new one --
// one bit tells them apart.  The bit is the mantissa LSB of each partial sum, forced to the phase ((n >> 1) & 1) ^ 1 (the
// mailbox is zeroed in front of every launch, so the first use of either buffer finds invalid slots): every 8-byte double
// validates itself, a slot is LMAX doubles (16 B for LMAX = 2), a poll pass moves half
// the bytes (tools/ubench/exchange.hip variant 11, P = 256: 1.51 against 1.87-2.12 us per exchange; same-box A/B of the update
// sweep, us per step: C3 5.97 -> 5.86, C4 11.40 -> 11.12, K = 512 7.53 -> 7.32, C5 40.9 -> 40.3).  Every workgroup reads the
// same slots -- its own included -- so all of them still form bitwise identical totals; the forced bit moves a partial sum by
// at most one ulp.  The mailbox between GPUs (xmail, 8 slots) keeps its 32-bit tags: nothing zeroes it between launches.
// (Slot width, measured at P = 256 with the tagged granules: 32-B slots 8.0k cycles per exchange, 64-B 8.45k, 128-B 10.2k,
// granule-major 10.3k: the cost grows with the number of 128-B lines a poll pass touches.)
#ifndef KR_SLOT_GRANULES
#define KR_SLOT_GRANULES (LMAX)
#endif
// One-XCD placement (P <= 32 workgroups exchanging through ONE L2): every slot on a 128-byte line of its own.  Packed, the 8 slots of
// C2 share one line, and that line's stores and polling loads serialise in the L2 (a poll pass took 600 ns with every slot present;
// update sweep 3.67 -> 3.57 us per step with one slot per line).
#ifndef KR_SLOT_GRANULES_LOCAL
#define KR_SLOT_GRANULES_LOCAL 16
#endif
// One replica of the mailbox per XCD: every workgroup stores its slot into all 8 replicas (one store instruction, 8 x L active
// lanes), the polling wave reads the replica of the XCD it runs on (HW_REG_XCC_ID) -- 32 pollers per 4 KB of slots instead of 256.
// A poll pass with every slot present took 0.84 us at C3 (tools/exchange_arrivals.py; 2-4 x the idle load latency: 256 CUs pull
// the same 4 KB through the few memory channels that hold them); in isolation (tools/ubench/exchange.hip, P = 256, uncached
// mailbox) 1.93 -> 1.60 us per exchange with replicas 64 KB apart, 1.70 with replicas 4 KB + 256 B apart.  Any replica holds
// every partial sum, so which one a wave reads changes timing only.
// (KR_MAIL_REPLICAS, KR_MAIL_RSTRIDE: kr_common.h)
// A wave-uniform pointer the compiler may hold in vector registers, for the "s" operand of an asm statement
__device__ __forceinline__ const unsigned long long *kr_uniform_ptr(const unsigned long long *p) {
    const unsigned long long v = (unsigned long long)(size_t)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (const unsigned long long *)(size_t)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ unsigned kr_xcc_id() {
    unsigned id;
    asm("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    return id & (KR_MAIL_REPLICAS - 1);
}
#ifndef KR_POLL_SLEEP
#define KR_POLL_SLEEP 0  // measured at P = 256: 0 -> 7.48 ms per update sweep, 1 -> 7.55, 4 -> 7.60, 16 -> 7.78
#endif
#define KR_POLL_CHUNKS 4  // slots per lane gathered with all loads in flight at once (P <= 256 in one go)
// KrPoll: one pass of the ROW-SPLIT ALL-GATHER (kr_device_rowsplit.h; elements = 16-byte words {lo32, tag, hi32, tag}, i.e. two
// self-validating 8-byte granules per word, so tearing between the halves of a word is harmless) with 16-byte loads that bypass
// the vector L1 (sc1).  All loads of a pass and their wait are ONE asm block: the compiler does not count asm loads in vmcnt,
// so nothing may touch the results before the block ends (C5 known-pulse sweeps 37.9 -> 34.3 us per step).
typedef unsigned int kr_v4u __attribute__((ext_vector_type(4)));
#define KR_PL(o, p, off) "global_load_dwordx4 %" #o ", %" #p ", off offset:" #off " sc1\n\t"
template <int LMAX> struct KrPoll;
template <> struct KrPoll<2> {
    static __device__ __forceinline__ void pass(const unsigned long long *p0, const unsigned long long *p1,
                                                const unsigned long long *p2, const unsigned long long *p3, const int nch,
                                                kr_v4u (&g)[KR_POLL_CHUNKS][2]) {
        if (nch > 2) {
            asm volatile(KR_PL(0, 8, 0) KR_PL(1, 8, 16) KR_PL(2, 9, 0) KR_PL(3, 9, 16) KR_PL(4, 10, 0) KR_PL(5, 10, 16)
                         KR_PL(6, 11, 0) KR_PL(7, 11, 16) "s_waitcnt vmcnt(0)"
                         : "=&v"(g[0][0]), "=&v"(g[0][1]), "=&v"(g[1][0]), "=&v"(g[1][1]), "=&v"(g[2][0]), "=&v"(g[2][1]),
                           "=&v"(g[3][0]), "=&v"(g[3][1])
                         : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
        } else if (nch > 1) {
            asm volatile(KR_PL(0, 4, 0) KR_PL(1, 4, 16) KR_PL(2, 5, 0) KR_PL(3, 5, 16) "s_waitcnt vmcnt(0)"
                         : "=&v"(g[0][0]), "=&v"(g[0][1]), "=&v"(g[1][0]), "=&v"(g[1][1]) : "v"(p0), "v"(p1) : "memory");
        } else {
            asm volatile(KR_PL(0, 2, 0) KR_PL(1, 2, 16) "s_waitcnt vmcnt(0)" : "=&v"(g[0][0]), "=&v"(g[0][1]) : "v"(p0) : "memory");
        }
    }
};

// One poll pass over all P slots of buffer `par`: lane handles slots lane, lane + 64, ...; the loads of (up to) 4 x 64 slots
// and their wait are ONE asm block of 16-byte sc1 loads (KrPoll above: the compiler neither counts nor reorders them).
// tot = per-lane partial of the sum (slot order fixed), returns whether every slot of this lane carried `phase`.
// Addressing: scalar base (buffer of this step) + ONE 32-bit lane offset + immediate chunk offsets, so that the poll keeps one
// VGPR of addresses live instead of four 64-bit pointers (kr_sweep_update's HERM variant sits at the 256-VGPR limit).  No clamp
// for the lanes past P: the mailbox is allocated with a tail of 64 slots (krotov_hip.hip), what they read is masked out.
#define KR_PLS(o, v, sb, off) "global_load_dwordx4 %" #o ", %" #v ", %" #sb " offset:" #off " sc1\n\t"
template <int LMAX> struct KrPollLsb;
template <> struct KrPollLsb<2> {   // slot = one 16-byte word {d0, d1}; 64 slots = 1 KB
    static constexpr int WORDS = 1;
    static __device__ __forceinline__ void pass(const unsigned long long *sbase, const unsigned voff, const int nch, kr_v4u (&g)[KR_POLL_CHUNKS][1]) {
        if (nch > 2)
            asm volatile(KR_PLS(0, 4, 5, 0) KR_PLS(1, 4, 5, 1024) KR_PLS(2, 4, 5, 2048) KR_PLS(3, 4, 5, 3072) "s_waitcnt vmcnt(0)"
                         : "=&v"(g[0][0]), "=&v"(g[1][0]), "=&v"(g[2][0]), "=&v"(g[3][0]) : "v"(voff), "s"(sbase) : "memory");
        else if (nch > 1)
            asm volatile(KR_PLS(0, 2, 3, 0) KR_PLS(1, 2, 3, 1024) "s_waitcnt vmcnt(0)" : "=&v"(g[0][0]), "=&v"(g[1][0]) : "v"(voff), "s"(sbase) : "memory");
        else
            asm volatile(KR_PLS(0, 1, 2, 0) "s_waitcnt vmcnt(0)" : "=&v"(g[0][0]) : "v"(voff), "s"(sbase) : "memory");
    }
};
template <> struct KrPollLsb<4> {   // slot = two 16-byte words {d0, d1}, {d2, d3}; 64 slots = 2 KB
    static constexpr int WORDS = 2;
    static __device__ __forceinline__ void pass(const unsigned long long *sbase, const unsigned voff, const int nch, kr_v4u (&g)[KR_POLL_CHUNKS][2]) {
        if (nch > 2) {
            const unsigned voff2 = voff + 4096u;
            asm volatile(KR_PLS(0, 8, 10, 0) KR_PLS(1, 8, 10, 16) KR_PLS(2, 8, 10, 2048) KR_PLS(3, 8, 10, 2064)
                         KR_PLS(4, 9, 10, 0) KR_PLS(5, 9, 10, 16) KR_PLS(6, 9, 10, 2048) KR_PLS(7, 9, 10, 2064) "s_waitcnt vmcnt(0)"
                         : "=&v"(g[0][0]), "=&v"(g[0][1]), "=&v"(g[1][0]), "=&v"(g[1][1]), "=&v"(g[2][0]), "=&v"(g[2][1]),
                           "=&v"(g[3][0]), "=&v"(g[3][1])
                         : "v"(voff), "v"(voff2), "s"(sbase) : "memory");
        } else if (nch > 1)
            asm volatile(KR_PLS(0, 4, 5, 0) KR_PLS(1, 4, 5, 16) KR_PLS(2, 4, 5, 2048) KR_PLS(3, 4, 5, 2064) "s_waitcnt vmcnt(0)"
                         : "=&v"(g[0][0]), "=&v"(g[0][1]), "=&v"(g[1][0]), "=&v"(g[1][1]) : "v"(voff), "s"(sbase) : "memory");
        else
            asm volatile(KR_PLS(0, 2, 3, 0) KR_PLS(1, 2, 3, 16) "s_waitcnt vmcnt(0)" : "=&v"(g[0][0]), "=&v"(g[0][1]) : "v"(voff), "s"(sbase) : "memory");
    }
};
// sgran: 8-byte words between two slots (KR_SLOT_GRANULES: packed; KR_SLOT_GRANULES_LOCAL: one slot per 128-byte line, P <= 64 only)
template <int LMAX>
__device__ __forceinline__ bool kr_poll_pass(const KrSweepArgs &a, const int par, const unsigned phase, const int lane, double (&tot)[LMAX],
                                             const unsigned sgran = KR_SLOT_GRANULES, const int seq = -1) {
    static_assert(LMAX == 2 || LMAX == 4, "slot = LMAX doubles in 16-byte words");
    static_assert(LMAX <= 4, "publishing lanes: 4 per replica");
    constexpr int WORDS = KrPollLsb<LMAX>::WORDS;
    const int L = a.L, P = a.P;
    bool ok = true;
#pragma unroll
    for (int l = 0; l < LMAX; ++l) tot[l] = 0.0;
    for (int q0 = 0; q0 < P; q0 += 64 * KR_POLL_CHUNKS) {
        const int left = P - q0, nch = (left + 63) >> 6;   // wave-uniform
        const unsigned long long *sbase = kr_uniform_ptr(a.mail + ((size_t)par * KR_MAIL_REPLICAS + kr_xcc_id()) * KR_MAIL_RSTRIDE + (size_t)q0 * sgran);
        kr_v4u g[KR_POLL_CHUNKS][WORDS];
        KrPollLsb<LMAX>::pass(sbase, (unsigned)lane * (8u * sgran), nch, g);
#pragma unroll
        for (int c = 0; c < KR_POLL_CHUNKS; ++c) {
            if (c < nch) {
                const bool mine = (q0 + 64 * c + lane) < P;
#pragma unroll
                for (int l = 0; l < LMAX; ++l) {
                    if (l < L) {
                        const unsigned lo = (l & 1) ? g[c][l >> 1].z : g[c][l >> 1].x, hi = (l & 1) ? g[c][l >> 1].w : g[c][l >> 1].y;
                        ok = ok & (((lo & 1u) == phase) | !mine);
#ifdef KR_GATHER_SEQ
                        if (seq >= 0 && a.debug && mine && (lo & 1u) == phase && ((lo >> 1) & 0x7fffu) != ((unsigned)seq & 0x7fffu)) {
                            atomicAdd(a.debug + 12, 1ull);
                            a.debug[13] = ((unsigned long long)(unsigned)seq << 32) | ((lo >> 1) & 0x7fffu);
                        }
#else
                        (void)seq;
#endif
                        tot[l] += mine ? __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)) : 0.0;
                    }
                }
            }
        }
    }
    return ok;
}
// Slot value of a partial sum: its bits with the LSB forced to the phase.
__device__ __forceinline__ unsigned long long kr_slot_bits(const double v, const unsigned phase) {
    return ((unsigned long long)__double_as_longlong(v) & ~1ull) | (unsigned long long)phase;
}
// The per-step exchange's slots in a diagnostics build (-DKR_GATHER_SEQ, tools/build_diag.sh): bits 15..1 carry the step index, so that
// kr_poll_pass can count partial sums that the one-bit validity accepts although they belong to an earlier step (a.debug[12]).
__device__ __forceinline__ unsigned long long kr_slot_bits_step(const double v, const unsigned phase, const int n) {
#ifdef KR_GATHER_SEQ   // (n < 0: the positive control of tools/diag_gather_seq.py -- a slot that claims to be four steps old)
    return (kr_slot_bits(v, phase) & ~0xfffeull) | ((unsigned long long)((unsigned)(n < 0 ? -n - 4 : n) & 0x7fffu) << 1);
#else
    (void)n;
    return kr_slot_bits(v, phase);
#endif
}

// EARLY pass of the polling wave (update kernels with window work, one GPU): issued as LDS-DMA in the middle of the window work
// -- global_load_lds_dwordx4: no result registers to keep live across the rest of the window --, checked when the window work is
// done.  The workgroups that publish LAST set the pace of a step: such a workgroup spent window (0.84 us) + one pass (0.76 us)
// after its publish although every other slot was visible long before (tools/exchange_arrivals.py); with the pass in flight during
// the second half of the window it is done ~0.3 us earlier.  This workgroup's own slot comes from registers (the value it has just
// published: its store may not be visible yet).  Issued right AFTER the publish the pass never hit (0 of 256 workgroups: the last
// workgroups of a step are a whole XCD publishing within 0.2 us of each other, and a store takes ~0.7 us to become visible).
// A workgroup whose early pass misses polls as before.
// MEASURED AND COMPILED OUT (KR_EARLY_POLL = 0): in the kernel the early pass hits in 170-200 of 256 workgroups per step, yet
// the step grows from 5.73 to 6.22 us at C3 (same box, wave 0 publishing); kept for the record and the diagnostics build.
// The LDS image mirrors the replica's buffer byte for byte (1 KB pieces: scalar base, lane offset, M0 = LDS address of the piece).
template <int LMAX>
__device__ __forceinline__ void kr_poll_issue_lds(const unsigned long long *sbase, const int pieces, unsigned long long *spoll, const int lane) {
    const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) void *)spoll;
    const unsigned voff = (unsigned)lane * 16u;
    for (int c = 0; c < pieces; ++c) {   // wave-uniform trip count
        unsigned keep;
        const unsigned long long *sb = kr_uniform_ptr(sbase + 128 * c);
        const unsigned ld = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds + 1024u * (unsigned)c));
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sb), "s"(ld) : "memory");
    }
}
// Check of the early pass: slot q of lane = lane + 64 c from the LDS image; `own` (this workgroup's slot index) comes from `mine_bits`.
template <int LMAX>
__device__ __forceinline__ bool kr_poll_check_lds(const KrSweepArgs &a, const unsigned long long *spoll, const unsigned phase, const int lane,
                                                  const int own, const unsigned long long (&mine_bits)[LMAX], double (&tot)[LMAX]) {
    const int L = a.L, P = a.P;
    bool ok = true;
#pragma unroll
    for (int l = 0; l < LMAX; ++l) tot[l] = 0.0;
    for (int q = lane; q < P; q += 64) {   // at most KR_POLL_CHUNKS trips (the caller checks P)
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            if (l < L) {
                const unsigned long long g = (q == own) ? mine_bits[l] : spoll[(size_t)q * LMAX + l];
                ok = ok & (((unsigned)g & 1u) == phase);
                tot[l] += __longlong_as_double((long long)g);
            }
        }
    }
    return ok;
}

// What kr_exchange hands to the window work: the window calls it ONCE, on every wave, about half-way through; on the polling wave it
// issues the early poll pass, on the others nothing.
template <int LMAX>
struct KrMidPoll {
    const unsigned long long *sbase;
    unsigned long long *spoll;
    int pieces, lane;
    bool armed, issued;
    __device__ __forceinline__ void operator()() {
        if (armed) { kr_poll_issue_lds<LMAX>(sbase, pieces, spoll, lane); issued = true; }
    }
};

// One wave gathers all P slots of buffer `par` until every slot carries `phase`; tot = sum over the slots, same order as in
// kr_exchange.  Bounded; false = gave up (the caller's wave 0 flags it).
template <int LMAX>
__device__ __forceinline__ bool kr_gather_slots(const KrSweepArgs &a, const int par, const unsigned phase, const int lane, double (&tot)[LMAX],
                                                const unsigned sgran = KR_SLOT_GRANULES) {
    const unsigned long long t0 = kr_realtime();
    for (unsigned spins = 1;; ++spins) {
        if (__all(kr_poll_pass<LMAX>(a, par, phase, lane, tot, sgran))) break;
        if ((spins & 63u) == 0u) {
            const unsigned st = __hip_atomic_load((kr_gu32 *)a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (st != 0u || kr_realtime() - t0 > a.spin_ticks) return false;
        }
    }
    kr_wave_sum_all<LMAX>(tot, a.L);
    return true;
}

struct KrNoWork { template <class M> __device__ __forceinline__ void operator()(M &) const {} };
#ifndef KR_EARLY_POLL
#define KR_EARLY_POLL 0
#endif
template <class F> struct KrIsNoWork { static constexpr bool value = false; };
template <> struct KrIsNoWork<KrNoWork> { static constexpr bool value = true; };

// `work` is called exactly once by every wave while the partial sums travel (after the publishing store, before the
// polling wave starts to poll): barrier-free work that does not depend on the total (kr_sweep_update, HERM path).
template <int LMAX, int ROWS, int SCHED = 0, class F = KrNoWork>
__device__ __forceinline__ bool kr_exchange(const KrSweepArgs &a, const int n, const int wg, const bool writer,
                                            const int rowid, const double (&part)[LMAX], double (&total)[LMAX],
                                            double *spart /*[LMAX][ROWS]*/, double *stot /*[LMAX+1]*/, F work = F(),
                                            const bool xlocal = false, const bool deposited = false) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = a.L, P = a.P;
    // Tags count exchange events over the life of the handle (32-bit, wrapping); the buffer parity is taken from
    // the TAG, not from the step index, so that consecutive events alternate buffers across launches as well:
    // with peers on other GPUs nothing separates one launch's last step from the next launch's first one.
    const unsigned tag = a.epoch_base + (unsigned)n + 1u;
    const int par = (int)(tag & 1u);                       // buffer of the mailbox between GPUs (xmail)
    // mailbox inside the GPU: buffer and phase from the step index of THIS launch (the mailbox is zeroed in front of it)
    const int lpar = kr_mail_buffer((unsigned)n);
    const unsigned phase = kr_mail_phase((unsigned)n);
    const unsigned sgran = xlocal ? (unsigned)KR_SLOT_GRANULES_LOCAL : (unsigned)KR_SLOT_GRANULES;   // words between slots

    constexpr bool EARLY = !KrIsNoWork<F>::value;   // a window exists: overlap the first poll pass with it
    __shared__ __attribute__((aligned(16))) unsigned long long spoll[EARLY ? 64 * KR_POLL_CHUNKS * LMAX : 2];

    // 1. the ROWS row owners of the workgroup deposit their partial sums (`deposited`, workgroup-uniform: the caller has done that
    //    already, in front of a barrier of its own -- kr_sweep_update's window variant deposits with the step's last barrier)
    if (!deposited) {
        if (writer) {
#pragma unroll
            for (int l = 0; l < LMAX; ++l) if (l < L || l < 2) spart[l * ROWS + rowid] = (l < L) ? part[l] : 0.0;   // (the pair sums read l = 1 at L = 1)
        }
        __syncthreads();
    }

    // workgroup partial: rows in a fixed order
    auto own_partial = [&](double (&t)[LMAX]) {
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            double v = 0.0;
            if (l < L || l < 2) {
#pragma unroll
                for (int r = 0; r < ROWS; r += 64) if (r + lane < ROWS) v += spart[l * ROWS + r + lane];
            }
            t[l] = v;
        }
        kr_wave_sum_all<LMAX>(t, L);
    };
    if (SCHED == 1 && wave == 1 && !(P == 1 && a.world == 1)) {
        double tot1[LMAX];
        own_partial(tot1);
        if (lane < 4 * KR_MAIL_REPLICAS && (lane & 3) < L && (!xlocal || (unsigned)(lane >> 2) == kr_xcc_id())) {   // lane = 4 * replica + l
            double v = 0.0;
#pragma unroll
            for (int l = 0; l < LMAX; ++l) if (l == (lane & 3)) v = tot1[l];
            const unsigned long long *dst = a.mail + ((size_t)lpar * KR_MAIL_REPLICAS + (lane >> 2)) * KR_MAIL_RSTRIDE + (size_t)wg * sgran + (lane & 3);
#ifdef KR_GATHER_SEQ
            const unsigned long long val = kr_slot_bits_step(v, phase, (a.rs_fault && wg == 1 && (n & 15) == 5) ? -n : n);
#else
            const unsigned long long val = kr_slot_bits_step(v, phase, n);
#endif
            if (xlocal) asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(dst), "v"(val) : "memory");   // stays in this XCD's L2
            else asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(dst), "v"(val) : "memory");
        }
    }
    if (wave != 0) { KrMidPoll<LMAX> nomid{nullptr, nullptr, 0, 0, false, false}; work(nomid); }
    if (wave == 2 && a.world > 1 && wg == 0) {
        // 4a. inter-GPU hop, sending side: wave 2 of workgroup 0 gathers the slots as well and stores the GPU total into
        //     every rank's xmail (slot = this rank).  Its stores cross xGMI; a wave that polls after storing waits for
        //     their acknowledgements with its first pass (vmcnt counts loads and stores together), so the polling wave 0
        //     never stores to a peer and this wave never waits (asm stores, not counted by the compiler either).
        if (SCHED == 1) for (int w = 0; w < a.poll_delay; ++w) __builtin_amdgcn_s_sleep(2);
        double tf[LMAX];
        const int W = a.world;
        if (kr_gather_slots<LMAX>(a, lpar, phase, lane, tf, sgran) && lane < 2 * L * W) {
            const int r = lane / (2 * L), g = lane % (2 * L), l = g >> 1;
            double v = 0.0;
#pragma unroll
            for (int ll = 0; ll < LMAX; ++ll) if (ll == l) v = tf[ll];
            const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
            const unsigned half = (g & 1) ? (unsigned)(bits >> 32) : (unsigned)bits;
            const unsigned long long *dst = a.peer_xmail[r] + ((size_t)par * W + a.rank) * (2 * LMAX) + g;
            const unsigned long long val = ((unsigned long long)tag << 32) | half;
            asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(dst), "v"(val) : "memory");
        }
    }
    if (wave == 0) {
        bool ok_all = true;
        double tot[LMAX];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) tot[l] = 0.0;
        // (SCHED 1: wave 1 publishes, and this wave needs the workgroup's own partial only when it IS the total -- two wave sums
        //  less in front of its window work, i.e. the poll starts ~0.2 us earlier after the publish)
        if (SCHED == 0 || KR_EARLY_POLL || (P == 1 && a.world == 1)) own_partial(tot);
        if (!(P == 1 && a.world == 1)) {
            // 2. publish this workgroup's partial
            if (SCHED == 0 && lane < 4 * KR_MAIL_REPLICAS && (lane & 3) < L) {   // lane = 4 * replica + l
                kr_gu64 *slot = (kr_gu64 *)(a.mail + ((size_t)lpar * KR_MAIL_REPLICAS + (lane >> 2)) * KR_MAIL_RSTRIDE + (size_t)wg * sgran);
                double v = 0.0;
#pragma unroll
                for (int l = 0; l < LMAX; ++l) if (l == (lane & 3)) v = tot[l];
                if (xlocal) {   // every workgroup of this launch sits on this XCD: one replica, and the store stays in its L2
                    if ((unsigned)(lane >> 2) == kr_xcc_id()) {
                        const unsigned long long *dst = (const unsigned long long *)slot + (lane & 3);
                        const unsigned long long val = kr_slot_bits_step(v, phase, n);
                        asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(dst), "v"(val) : "memory");
                    }
                } else {
                    __hip_atomic_store(slot + (lane & 3), kr_slot_bits_step(v, phase, n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#ifdef KR_ARRIVALS  // diagnostic build: when did every workgroup publish / complete two exchanges in the middle of the sweep
            // (stamps stay in registers until the exchange is over: a store in front of the poll pass would be waited for by its vmcnt)
            const bool arr = a.debug != nullptr && (n == a.NT / 2 || n == a.NT / 2 + 1);
            unsigned long long arr_pub = 0, arr_wend = 0, arr_m0 = 0, arr_m1 = 0, arr_m2 = 0;
            if (arr) arr_pub = kr_realtime();
#endif
            const int pieces = (P * 8 * LMAX + 1023) >> 10;
            const bool early = EARLY && SCHED == 0 && a.world == 1 && P <= 64 * KR_POLL_CHUNKS && P > 1;
            unsigned long long mine_bits[LMAX];
#pragma unroll
            for (int l = 0; l < LMAX; ++l) mine_bits[l] = kr_slot_bits_step(tot[l], phase, n);
            KrMidPoll<LMAX> mid{a.mail + ((size_t)lpar * KR_MAIL_REPLICAS + kr_xcc_id()) * KR_MAIL_RSTRIDE, spoll, pieces, lane, early, false};
            work(mid);
#ifdef KR_ARRIVALS
            if (arr) { arr_wend = kr_realtime(); arr_m0 = __builtin_amdgcn_s_memtime(); }
#endif
            bool done_early = false;
            if (mid.issued) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                done_early = __all(kr_poll_check_lds<LMAX>(a, spoll, phase, lane, wg, mine_bits, tot));
            }
            if (SCHED == 1) for (int w = 0; w < a.poll_delay; ++w) __builtin_amdgcn_s_sleep(2);
            // 3. gather all P slots (lane handles slots lane, lane+64, ...), bounded spin.  With peers on other GPUs only
            //    the forwarding wave gathers them (one poller instead of P: the stores become visible sooner); everybody
            //    else waits for the grand total in xmail, which includes this GPU's own contribution.
            const unsigned long long t0 = kr_realtime();
            unsigned spins = 0;
            for (; a.world == 1 && !done_early;) {
#ifdef KR_ARRIVALS
                if (arr) arr_m1 = __builtin_amdgcn_s_memtime();
#endif
                const bool ok = kr_poll_pass<LMAX>(a, lpar, phase, lane, tot, sgran, n);
#ifdef KR_ARRIVALS
                if (arr) arr_m2 = __builtin_amdgcn_s_memtime();
#endif
                if (__all(ok)) break;
                if ((++spins & 63u) == 0u) {
                    const unsigned st = __hip_atomic_load((kr_gu32 *)a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (st != 0u || kr_realtime() - t0 > a.spin_ticks) {
                        if (lane == 0 && st == 0u) {
                            __hip_atomic_store((kr_gu32 *)a.status + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store((kr_gu32 *)a.status + 2, (unsigned)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store((kr_gu32 *)a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        ok_all = false;
                        break;
                    }
                }
                __builtin_amdgcn_s_sleep(KR_POLL_SLEEP);
            }
            if (a.world == 1) kr_wave_sum_all<LMAX>(tot, L);
#ifdef KR_ARRIVALS
            if (arr && lane == 0) {   // word 3: passes | cycles of the last pass << 8 | cycles window end -> sums done << 32 (s_memtime)
                const unsigned long long m3 = __builtin_amdgcn_s_memtime();
                unsigned long long *d = a.debug + 8 + (size_t)wg * 8 + 4 * (n - a.NT / 2);
                d[0] = arr_pub; d[1] = arr_wend; d[2] = kr_realtime();
                d[3] = (unsigned long long)((spins + (done_early ? 100u : 0u)) & 255u) | (((arr_m2 - arr_m1) & 0xffffffull) << 8) | (((m3 - arr_m0) & 0xffffffull) << 32);
            }
#endif

            // 4. inter-GPU hop, receiving side (the sending side is wave 2 of workgroup 0, above): every workgroup polls
            //    its own GPU's xmail and sums the ranks.
            if (a.world > 1 && ok_all) {
                const int W = a.world;
                for (int w = 0; w < a.xpoll_delay; ++w) __builtin_amdgcn_s_sleep(2);
                const unsigned long long t1 = kr_realtime();
                spins = 0;
                double gt[LMAX];
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int l = 0; l < LMAX; ++l) gt[l] = 0.0;
                    {  // lane r < W reads rank r's slot (all loads in flight); fixed reduction tree below
                        const bool mine = lane < W;
                        kr_gu64 *sq = (kr_gu64 *)(a.xmail + ((size_t)par * W + min(lane, W - 1)) * (2 * LMAX));
                        unsigned long long g[2 * LMAX];
#pragma unroll
                        for (int w = 0; w < 2 * LMAX; ++w)
                            g[w] = (w < 2 * L) ? __hip_atomic_load(sq + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0ull;
#pragma unroll
                        for (int l = 0; l < LMAX; ++l) {
                            if (l < L) {
                                const bool good = ((unsigned)(g[2 * l] >> 32) == tag) & ((unsigned)(g[2 * l + 1] >> 32) == tag);
                                ok = ok & (good | !mine);
                                const unsigned long long bits = (g[2 * l + 1] << 32) | (g[2 * l] & 0xffffffffull);
                                gt[l] = mine ? __longlong_as_double((long long)bits) : 0.0;
                            }
                        }
                    }
                    if (__all(ok)) break;
                    if ((++spins & 63u) == 0u) {
                        const unsigned st = __hip_atomic_load((kr_gu32 *)a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (st != 0u || kr_realtime() - t1 > a.spin_ticks) {
                            if (lane == 0 && st == 0u) {
                                __hip_atomic_store((kr_gu32 *)a.status + 1, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                __hip_atomic_store((kr_gu32 *)a.status + 2, (unsigned)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                __hip_atomic_store((kr_gu32 *)a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                            ok_all = false;
                            break;
                        }
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                kr_wave_sum_all<LMAX>(gt, L);   // same tree on every GPU: bitwise identical totals
#pragma unroll
                for (int l = 0; l < LMAX; ++l) tot[l] = gt[l];
            }
        } else {
            KrMidPoll<LMAX> nomid{nullptr, nullptr, 0, 0, false, false};
            work(nomid);
        }
        if (lane == 0) {
#pragma unroll
            for (int l = 0; l < LMAX; ++l) stot[l] = tot[l];
            stot[LMAX] = ok_all ? 1.0 : 0.0;
        }
    }
    __syncthreads();
    // totals and flag in ONE LDS round trip: the caller returns on a bad flag, and left alone the compiler reads the flag, branches,
    // and only then asks for the totals (a second LDS latency on the critical path of every step)
    const double flag = stot[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) total[l] = stot[l];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) asm volatile("" :: "v"(total[l]));
    return flag != 0.0;
}

// =============================================================================================
// Update sweep (optimize.jl:321-370): strictly sequential in n, one global sum per step.
// Wave roles: wave 0 runs the exchange; the LAST wave moves states between HBM and LDS
// (chi(t_{n+1}) prefetch, forward-storage writes); every wave takes part in the matvecs.
// =============================================================================================
// CAL = 1: the same kernel under another name for the short sweeps of calibrate_exchange (krotov_hip.hip), so that
// profiler statistics of kr_sweep_update<..., 0, HERM> hold the caller's launches only.
// HERM = 1 (every control operator H_l Hermitian, checked by kr_create): Im<chi|H_l|psi> = Im<H_l chi|psi>, and
// c_l(n) = H_l chi(t_n) depends on stored data only.  The step then STARTS with a plain dot product of c_l(n) with psi(t_n)
// and the exchange; forming H(eps_guess), the products b = H(eps_guess) psi and a_l = H_l psi (needed for the free first
// application) and c_l(n+1) are computed by all waves WHILE the partial sums travel (kr_exchange's `work`), i.e. phase A
// leaves the critical path of the step.  chi(t_{n+1}) must then be in LDS one step earlier (prefetch distance 2).
// XLOC = 1: the launch uses the one-XCD placement (KrSweepArgs::wg_stride > 1, P <= 32 workgroups).  A template parameter, not a
// run-time flag: as a flag it cost the big grids 2 % (C3: 5.55 -> 5.65 us per step, same-box A/B) for a path they never take.
// POLY = 1 (round 5): non-linear controls with polynomial amplitudes f_l of degree <= 3 (KrSweepArgs::poly): the matrices are combined
// with f_l(eps_l) instead of eps_l, the totals are scaled by f_l'(eps_guess) before the pulse update.  A separate instantiation: the
// linear kernel's code is untouched.
// The amplitude's change f_l(eg + d) - f_l(eg) for a cubic f_l, exactly, as its Taylor polynomial at the guess value: d (f' + d (f''/2 + d c3)).
// f' = f_l'(eg) comes with the step's prefetched scalars, f''/2 = c2 + 3 c3 eg is one FMA that needs no total, c2 and c3 are loaded once:
// nothing but three dependent FMAs joins the critical path behind the exchange (evaluating f_l(eps_new) from its coefficients there --
// scalar loads and a wait that also covers every LDS operation in flight -- cost C3 0.6 us per step).
__device__ __forceinline__ double kr_poly3_delta(const double d, const double fp, const double f2, const double c3) { return d * fma(d, fma(d, c3, f2), fp); }
template <int NP, int LMAX, int KT, int SCHED, int CAL, int HERM, int XLOC = 0, int POLY = 0>
__global__ __launch_bounds__((KrGeom<NP, LMAX, KT>::THREADS), (KrGeom<NP, LMAX, KT>::THREADS / 256)) void kr_sweep_update(const KrSweepArgs a) {
    using G = KrGeom<NP, LMAX, KT>;
    constexpr int JS = G::JS, JQ = G::JQ, RT = G::RT, SG = G::SG, KTILE = G::KTILE, NPP = G::NPP, NW = G::NW;
    constexpr int CH = (KTILE * NP + 63) / 64;  // 16-B elements per lane when one wave moves a whole tile
    __shared__ __attribute__((aligned(16))) double2 sx[2][KTILE][NPP];
    __shared__ __attribute__((aligned(16))) double2 sy[2][KTILE][NPP];
    __shared__ __attribute__((aligned(16))) double2 sH0[G::H0LDS ? JS * RT : 1];
    __shared__ __attribute__((aligned(16))) double2 sc[2][KTILE][NPP];  // chi(t_k) in slot k & 1, padded like sx (it is a matvec operand when HERM)
    __shared__ double spart[LMAX * 64];
    __shared__ double stot[LMAX + 1];
    // one-XCD placement of a small grid (KrSweepArgs::wg_stride): only every wg_stride-th block works
    const int wstride = (XLOC && a.wg_stride > 1) ? a.wg_stride : 1;
    if (XLOC && wstride > 1 && (blockIdx.x % wstride) != 0) return;
    const int tid = threadIdx.x, wg = blockIdx.x / wstride, lane = tid & 63, wave = tid >> 6;
    const int team = tid / RT, tt = tid % RT, i = tt / JQ, jq = tt % JQ;
    const bool writer = (jq == 0), row_ok = (i < a.N);
    const bool memwave = (wave == NW - 1);
    const double my_rinv = kr_rinv[lane <= KR_MAX_ORDER ? lane : 0];
    const double my_thr = kr_load_my_thr(a);
    const KrGen<LMAX> gx = kr_gen_ctx<LMAX>(a, a.wg_gen ? a.wg_gen[wg] : 0);
    // Did the placement work?  Every workgroup posts its XCC id behind the mailbox (zeroed in front of the launch), wave 0 reads
    // all P entries: local iff they all carry this workgroup's id.  Every workgroup reads the SAME complete table, so all of them
    // reach the same conclusion (they must: the local path uses one replica and one slot per 128-byte line); a workgroup that never
    // sees the table complete aborts the sweep like any other bounded spin (the launch's workgroups are not co-resident).
    bool xlocal = false;
    if (XLOC && wstride > 1 && a.P <= 32) {
        __shared__ int sxl;
        if (wave == 0) {
            kr_gu64 *xtab = (kr_gu64 *)(a.mail + KR_MAIL_XTAB_OFF);
            const unsigned xcc = kr_xcc_id();
            if (lane == 0) __hip_atomic_store(xtab + wg, (1ull << 32) | xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long t0 = kr_realtime();
            bool same = false, dead = false;
            for (;;) {
                const unsigned long long v = __hip_atomic_load(xtab + min(lane, a.P - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all((v >> 32) != 0ull)) { same = __all((unsigned)(v & 15ull) == xcc); break; }
                if (kr_realtime() - t0 > a.spin_ticks) {
                    if (lane == 0) {
                        __hip_atomic_store((kr_gu32 *)a.status + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store((kr_gu32 *)a.status + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store((kr_gu32 *)a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    dead = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if (lane == 0) sxl = dead ? -1 : (same ? 1 : 0);
        }
        __syncthreads();
        if (sxl < 0) return;
        xlocal = (sxl != 0);
    }
    const int rowid = tid / JQ;  // 0..63
    const int pi_ = i + i / JS;
    const int xoff = jq * (JS + 1);
    const int N = a.N, NT = a.NT, L = a.L;
    const size_t slab = (size_t)a.Kp * N;
    const size_t tile_base = (size_t)wg * KTILE * N;  // the tile's trajectories are contiguous in every [Kp][N] slab
    const int tile_elems = KTILE * N;

    // generator slices: H_1..H_L always in registers; H0 in registers or (H0LDS) in LDS, thread-major
    double Hr[LMAX + 1][JS], Hi[LMAX + 1][JS];
#pragma unroll
    for (int m = 0; m <= LMAX; ++m) {
#pragma unroll
        for (int jj = 0; jj < JS; ++jj) {
            double2 h = make_double2(0.0, 0.0);
            if (m <= L) h = gx.Hpack[((size_t)m * JS + jj) * RT + tt];
            if (G::H0LDS && m == 0) {
                sH0[jj * RT + tt] = h;
                Hr[m][jj] = 0.0; Hi[m][jj] = 0.0;  // dead: never read when H0LDS
            } else {
                Hr[m][jj] = h.x;
                Hi[m][jj] = h.y;
            }
        }
    }

    // memory wave: element e = lane + 64 c of the tile <-> (trajectory tq, row i)
    int m_tq[CH], m_i[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int e = lane + 64 * c;
        m_tq[c] = (e < tile_elems) ? e / N : -1;
        m_i[c] = (e < tile_elems) ? e % N : 0;
    }
    double2 cpf[CH];  // chi prefetch registers (memory wave)

    double pr[KT], pim[KT];
#pragma unroll
    for (int q = 0; q < KT; ++q) {
        const int tq = q * SG + team;
        double2 v = make_double2(0.0, 0.0);
        if (row_ok) v = a.init[tile_base + (size_t)tq * N + i];
        pr[q] = v.x; pim[q] = v.y;
        if (writer) {
            sx[0][tq][pi_] = v;
            double2 c = make_double2(0.0, 0.0);
            if (row_ok) c = a.X[tile_base + (size_t)tq * N + i];  // chi_k(t_0)
            sc[0][tq][pi_] = c;
            double2 c1 = make_double2(0.0, 0.0);
            if (HERM && row_ok && NT > 1) c1 = a.X[slab + tile_base + (size_t)tq * N + i];  // chi_k(t_1): read in step 0's window
            sc[1][tq][pi_] = c1;
        }
    }
    if (memwave && a.store) {
#pragma unroll
        for (int c = 0; c < CH; ++c)
            if (m_tq[c] >= 0) a.store[tile_base + lane + 64 * c] = a.init[tile_base + lane + 64 * c];
    }
    double gint[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) gint[l] = 0.0;
    __syncthreads();
    // HERM: c_l(n) = (H_l chi(t_n))[row i], carried from the window of step n-1; c_l(0) here
    double clr[KT][LMAX], cli[KT][LMAX];
#pragma unroll
    for (int q = 0; q < KT; ++q) {
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            clr[q][l] = 0.0; cli[q][l] = 0.0;
            if (HERM && l < L) kr_row_dot<JS, JQ>(Hr[1 + l], Hi[1 + l], &sc[0][q * SG + team][xoff], clr[q][l], cli[q][l]);
        }
    }

    // optional phase stamps of workgroup 0 (diagnostic runs only; a.debug == nullptr otherwise)
    const bool stamp = (a.debug != nullptr) && (wg == 0);
    unsigned long long tA = 0, tX = 0, tB = 0, tS = 0, tB1 = 0, tB2 = 0, t_prev = 0, rt0 = 0;
    if (stamp) { t_prev = __builtin_amdgcn_s_memtime(); rt0 = kr_realtime(); }
    int cur = 0;
    // guess pulse, step size alpha = S/lambda and dt of the NEXT step are fetched one step ahead
    double eg_nx[LMAX], al_nx[LMAX], dt_nx = a.dt[0];
    // POLY: the coefficients of f_l (uniform: scalar registers); f_l(eps_guess) and f_l'(eps_guess) are formed at the top of the step,
    // in the shadow of the exchange (arrays of them, prefetched like eps_guess, cost eight more VGPRs: the HERM variant then spilled)
    double pc0[POLY ? LMAX : 1], pc1[POLY ? LMAX : 1], pc2[POLY ? LMAX : 1], pc3[POLY ? LMAX : 1];
    if (POLY) {
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            // (v_readfirstlane: the compiler kept them in vector registers otherwise -- 16 VGPRs the HERM variant does not have)
            pc0[l] = kr_readlane_f64((l < L) ? a.poly[4 * l] : 0.0, 0); pc1[l] = kr_readlane_f64((l < L) ? a.poly[4 * l + 1] : 0.0, 0);
            pc2[l] = kr_readlane_f64((l < L) ? a.poly[4 * l + 2] : 0.0, 0); pc3[l] = kr_readlane_f64((l < L) ? a.poly[4 * l + 3] : 0.0, 0);
        }
    }
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
        eg_nx[l] = (l < L) ? a.eps[(size_t)l * NT] : 0.0;
        al_nx[l] = (l < L) ? a.alpha[(size_t)l * NT] : 0.0;
    }
    const double *econ_rows = a.econ ? a.econ + ((size_t)gx.gen * NT) * KR_ECON_W + (lane & (KR_ECON_W - 1)) : nullptr;
    double ec_nx = econ_rows ? econ_rows[0] : 0.0;
    // This workgroup's log of the totals it gathers (KrSweepArgs::dulog; kr_pulse_finish forms eps_new from workgroup 0's and compares
    // all of them).  Stored behind the exchange by lane 0 of wave 2, which issues no other vector-memory operation.  Measured, C3 / C2
    // update sweep against the kernel without any of it (5.21 / 3.23 us per step, same box, profiles/r05_ab_guard.log): this form 5.29 /
    // 3.19; the totals parked in LDS and stored at the top of the next step 5.35 / 3.27, by the memory wave next to its storage writes
    // 5.33 / 3.29, through a 64-step ring in LDS with one coalesced store per 64 steps 5.24 / 3.35; a fold of the pulse values kept in
    // two VGPRs 5.35 / 3.27, in scalar registers 5.33, in LDS 5.37 / 3.32; every workgroup's memory wave writing its own copy of eps_new
    // 5.44 / 3.28.  The kernel's speed hangs on its register allocation and on which wave reaches a barrier last, not on the work added.
    double *const dulog = a.dulog + (size_t)wg * NT * LMAX;
    for (int n = 0; n < NT; ++n) {
        const double dt = dt_nx;
        const double ec = ec_nx;
        double eg[LMAX], al[LMAX];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) { eg[l] = eg_nx[l]; al[l] = al_nx[l]; }

        if (n + 1 < NT) {
            dt_nx = a.dt[n + 1];
            if (econ_rows) ec_nx = econ_rows[(size_t)(n + 1) * KR_ECON_W];
#pragma unroll
            for (int l = 0; l < LMAX; ++l) {
                if (l < L) { eg_nx[l] = a.eps[(size_t)l * NT + n + 1]; al_nx[l] = a.alpha[(size_t)l * NT + n + 1]; }
            }
        }
        double Er[JS], Ei[JS];
        double a0r[KT], a0i[KT], alr[KT][LMAX], ali[KT][LMAX];
        double part[LMAX];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) part[l] = 0.0;
        // H(eps_guess) of this step (eps_guess is known in advance; exact, no drift), b = H(eps_guess) psi, a_l = H_l psi
        auto form_E = [&]() {
            double agl[LMAX];   // the amplitudes the matrices are combined with: f_l(eps_guess) (POLY), or eps_guess itself
#pragma unroll
            for (int l = 0; l < LMAX; ++l) agl[l] = POLY ? fma(fma(fma(pc3[l], eg[l], pc2[l]), eg[l], pc1[l]), eg[l], pc0[l]) : eg[l];
#pragma unroll
            for (int jj = 0; jj < JS; ++jj) {
                double r, im;
                if (G::H0LDS) { const double2 h0 = sH0[jj * RT + tt]; r = h0.x; im = h0.y; }
                else { r = Hr[0][jj]; im = Hi[0][jj]; }
#pragma unroll
                for (int l = 0; l < LMAX; ++l) {
                    r = fma(agl[l], Hr[1 + l][jj], r);
                    im = fma(agl[l], Hi[1 + l][jj], im);
                }
                Er[jj] = r;
                Ei[jj] = im;
            }
        };
        auto apply_b = [&]() {
#pragma unroll
            for (int q = 0; q < KT; ++q) kr_row_dot<JS, JQ>(Er, Ei, &sx[cur][q * SG + team][xoff], a0r[q], a0i[q]);
        };
        auto apply_controls = [&]() {
#pragma unroll
            for (int q = 0; q < KT; ++q) {
                const double2 *src = &sx[cur][q * SG + team][xoff];
#pragma unroll
                for (int l = 0; l < LMAX; ++l) {
                    alr[q][l] = 0.0; ali[q][l] = 0.0;
                    if (l < L) kr_row_dot<JS, JQ>(Hr[1 + l], Hi[1 + l], src, alr[q][l], ali[q][l]);
                }
            }
        };
        if (!HERM) {
            // ---- phase A: a_l = H_l psi and the partial sums Im<chi|H_l|psi>; H(eps_guess) and b = H(eps_guess) psi, which only the
            //      free first application needs, wait for the window (round 3: they used to sit in front of the exchange) ----
            apply_controls();
#pragma unroll
            for (int q = 0; q < KT; ++q) {
                const double2 chi = sc[n & 1][q * SG + team][pi_];
                // Im(conj(chi) * a) = chi_r a_i - chi_i a_r ; one lane per row contributes
#pragma unroll
                for (int l = 0; l < LMAX; ++l)
                    if (l < L && ((gx.present >> l) & 1u)) part[l] += chi.x * ali[q][l] - chi.y * alr[q][l];
            }
        } else if (n == 0) {
            // ---- Im<H_l chi|psi> = c_r psi_i - c_i psi_r from the carried c_l(n); everything else moves into the window.
            //      (Steps n > 0: done and deposited at the end of step n - 1, deposit_next below.) ----
#pragma unroll
            for (int q = 0; q < KT; ++q) {
#pragma unroll
                for (int l = 0; l < LMAX; ++l)
                    if (l < L && ((gx.present >> l) & 1u)) part[l] += clr[q][l] * pim[q] - cli[q][l] * pr[q];
            }
        }
        // HERM: the partial sums of step n + 1 need c_l(n+1) (this step's window) and psi(t_{n+1}) (this step's result), both in
        // this thread's registers when the step ends: the row owners deposit them together with psi(t_{n+1}), in front of the
        // step's LAST barrier -- the next exchange starts without a deposit and a barrier of its own.
        auto deposit_next = [&]() {
            if (HERM && n + 1 < NT) {
                double pn[LMAX];
#pragma unroll
                for (int l = 0; l < LMAX; ++l) pn[l] = 0.0;
#pragma unroll
                for (int q = 0; q < KT; ++q) {
#pragma unroll
                    for (int l = 0; l < LMAX; ++l)
                        if (l < L && ((gx.present >> l) & 1u)) pn[l] += clr[q][l] * pim[q] - cli[q][l] * pr[q];
                }
                if (writer) {
#pragma unroll
                    for (int l = 0; l < LMAX; ++l) if (l < L || l < 2) spart[l * 64 + rowid] = (l < L) ? pn[l] : 0.0;
                }
            }
        };
        auto window = [&](auto &mid) {
            form_E();
            apply_b();
            if (HERM) {
                apply_controls();
                if (KR_EARLY_POLL) mid();   // (polling wave: early poll pass, kr_exchange)
#pragma unroll
                for (int q = 0; q < KT; ++q) {
#pragma unroll
                    for (int l = 0; l < LMAX; ++l) {
                        // c_l(n) has been consumed by the dot products above: c_l(n+1) takes its place
                        if (l < L && n + 1 < NT)
                            kr_row_dot<JS, JQ>(Hr[1 + l], Hi[1 + l], &sc[(n + 1) & 1][q * SG + team][xoff], clr[q][l], cli[q][l]);
                    }
                }
            }
        };
        if (stamp) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tA += t - t_prev; t_prev = t; }

        // ---- global sum over all trajectories of all workgroups (and GPUs) ------------------
        double du[LMAX];
        if (!kr_exchange<LMAX, 64, SCHED>(a, n, wg, writer, rowid, part, du, spart, stot, window, XLOC ? xlocal : false, HERM && n > 0)) return;
        if (tid == 128) {
#pragma unroll
            for (int l = 0; l < LMAX; ++l) dulog[(size_t)n * LMAX + l] = (l == 0 && a.rs_fault == 2 && wg == 1 && n == 3) ? -du[l] : du[l];   // (rs_fault 2: the planted fault of the tests)
        }

        if (stamp) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tX += t - t_prev; t_prev = t; }

        // ---- pulse update (optimize.jl:351-358), identical on every workgroup ---------------
        double e[LMAX], de[LMAX];
        // H(eps_new) = H(eps_guess) + sum_l (eps_new - eps_guess)_l H_l   (H(eps_guess) was formed before phase A)
        auto update_E = [&]() {
#pragma unroll
            for (int jj = 0; jj < JS; ++jj) {
#pragma unroll
                for (int l = 0; l < LMAX; ++l) {
                    Er[jj] = fma(de[l], Hr[1 + l][jj], Er[jj]);
                    Ei[jj] = fma(de[l], Hi[1 + l][jj], Ei[jj]);
                }
            }
        };
        double vr[KT], vi[KT];
        auto form_v = [&](const double fsv, const bool store) {   // v = -i fs (H(eps_new) psi) = -i fs (b + sum de_l a_l): no matvec
#pragma unroll
            for (int q = 0; q < KT; ++q) {
                double wr = a0r[q], wi = a0i[q];
#pragma unroll
                for (int l = 0; l < LMAX; ++l) { wr = fma(de[l], alr[q][l], wr); wi = fma(de[l], ali[q][l], wi); }
                vr[q] = fsv * wi;   // -i (wr + i wi) = wi - i wr
                vi[q] = -fsv * wr;
                if (store && writer) sy[0][q * SG + team][pi_] = make_double2(vr[q], vi[q]);
            }
        };
        double theta = gx.nrm[0];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            e[l] = 0.0; de[l] = 0.0;
            if (l < L) {
                const double alpha = al[l];  // = S[l][n] / lambda[l], IEEE division done once on the host
                // POLY: mu_l = f_l'(eps_guess) H_l -- the total of Im<chi|H_l|psi> times the derivative at the guess value (optimize.jl:337,345)
                const double fpl = POLY ? fma(fma(3.0 * pc3[l], eg[l], 2.0 * pc2[l]), eg[l], pc1[l]) : 1.0;   // f_l'(eps_guess)
                const double dul = POLY ? __dmul_rn(fpl, du[l]) : du[l];
                const double deps = __dmul_rn(alpha, dul);
                e[l] = __dadd_rn(eg[l], deps);
                // what H(eps_guess) has to be corrected by: the change of the AMPLITUDE (== deps up to rounding for linear controls)
                de[l] = POLY ? kr_poly3_delta(e[l] - eg[l], fpl, fma(3.0 * pc3[l], eg[l], pc2[l]), pc3[l]) : (e[l] - eg[l]);
                const double an = POLY ? fma(fma(fma(pc3[l], e[l], pc2[l]), e[l], pc1[l]), e[l], pc0[l]) : e[l];   // (only the step's norm bound needs it)
                const double ad = fabs(dul);
                gint[l] = __dadd_rn(gint[l], __dmul_rn(__dmul_rn(alpha, __dmul_rn(ad, ad)), dt));
                theta += fabs(an) * gx.nrm[1 + l];
            }
        }
        theta *= fabs(dt);
        int s, m;
        kr_plan_fast(a, my_thr, theta, s, m);
        const KrEcon eco = kr_econ_select(ec, theta, s, m);
        const double my_g = eco.on ? ec : my_rinv;   // Horner ratios of this step, lane j holds g_j

        // ---- phase B: psi <- exp(-i H dt) psi.  The first application H(eps_new) psi = b + sum de_l a_l
        //      is already known, so sub-step 0 costs m-1 matvecs:
        //      psi' = psi + y1,  y_m = v,  y_{j-1} = v + (-i fs / j) H y_j,  v = -i fs (H psi)
        const double fs = (s == 1) ? dt : dt / (double)s;
        {
            form_v(fs, m > 1);
            // (round 4: H(eps_new) is formed BEHIND the LDS write of v, whose latency then runs under these 32 FMAs: C3 5.24 -> 5.19 us
            //  per step, same box -- left to itself the compiler kept the source order.  Moving MORE behind it -- the running integral,
            //  theta, the plan, the economized row: nothing waits for them -- measured slower at every N: 5.19 -> 5.30, C2 3.22 -> 3.33)
            __builtin_amdgcn_sched_barrier(0);
            update_E();
            if (m > 1) kr_team_sync<RT>();
            // memory wave: start fetching chi(t_{n+1}); it lands while the Taylor rounds run
            if (memwave && n + 1 + HERM < NT) {
                // unconditional loads from a clamped address: a per-lane conditional assignment would make the compiler
                // wait for the data right here (select against the old register value) -- a full HBM latency per step
#pragma unroll
                for (int c = 0; c < CH; ++c)
                        cpf[c] = a.X[(size_t)(n + 1 + HERM) * slab + tile_base + min(lane + 64 * c, tile_elems - 1)];
            }
            if (stamp) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tB1 += t - t_prev; }
            int w = 1;
            double yr[KT], yi[KT];
#pragma unroll
            for (int q = 0; q < KT; ++q) { yr[q] = vr[q]; yi[q] = vi[q]; }
            for (int j = m; j >= 2; --j) {
                const double f = fs * kr_rinv_lane(my_g, j);
#pragma unroll
                for (int q = 0; q < KT; ++q) {
                    const int tq = q * SG + team;
                    double sr, si;
                    kr_row_dot<JS, JQ>(Er, Ei, &sy[w ^ 1][tq][xoff], sr, si);
                    yr[q] = fma(f, si, vr[q]);
                    yi[q] = fma(-f, sr, vi[q]);
                    if (j > 2 && writer) sy[w][tq][pi_] = make_double2(yr[q], yi[q]);
                }
                if (j > 2) kr_team_sync<RT>();
                w ^= 1;
            }
            if (stamp) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tB2 += t - t_prev; }
#pragma unroll
            for (int q = 0; q < KT; ++q) {
                pr[q] = fma(eco.a1, yr[q], eco.a0 * pr[q]);     // (Taylor: a0 = a1 = 1, the plain sum)
                pim[q] = fma(eco.a1, yi[q], eco.a0 * pim[q]);
                if (writer) sx[cur ^ 1][q * SG + team][pi_] = make_double2(pr[q], pim[q]);
            }
            if (s == 1 && memwave && n + 1 + HERM < NT) {  // hand chi(t_{n+1}) (HERM: chi(t_{n+2})) to the next step
#pragma unroll
                for (int c = 0; c < CH; ++c) if (m_tq[c] >= 0) sc[(n + 1 + HERM) & 1][m_tq[c]][m_i[c] + m_i[c] / JS] = cpf[c];
            }
            if (s == 1) deposit_next();
            __syncthreads();
            cur ^= 1;
        }
        for (int sub = 1; sub < s; ++sub) {  // further sub-steps (only when ||H dt|| > KR_THETA_MAX)
            int w = 0;
            for (int j = m; j >= 1; --j) {
                const double f = fs * kr_rinv_lane(my_rinv, j);
#pragma unroll
                for (int q = 0; q < KT; ++q) {
                    const int tq = q * SG + team;
                    const double2 *src = (j == m) ? &sx[cur][tq][xoff] : &sy[w ^ 1][tq][xoff];
                    double sr, si;
                    kr_row_dot<JS, JQ>(Er, Ei, src, sr, si);
                    const double y_r = fma(f, si, pr[q]);
                    const double y_i = fma(-f, sr, pim[q]);
                    if (j > 1) {
                        if (writer) sy[w][tq][pi_] = make_double2(y_r, y_i);
                    } else {
                        pr[q] = y_r; pim[q] = y_i;
                        if (writer) sx[cur ^ 1][tq][pi_] = make_double2(y_r, y_i);
                    }
                }
                if (j == 1 && sub == s - 1 && memwave && n + 1 + HERM < NT) {
#pragma unroll
                    for (int c = 0; c < CH; ++c) if (m_tq[c] >= 0) sc[(n + 1 + HERM) & 1][m_tq[c]][m_i[c] + m_i[c] / JS] = cpf[c];
                }
                if (j == 1 && sub == s - 1) { deposit_next(); __syncthreads(); }   // end of the step: states and chi change hands between waves
                else kr_team_sync<RT>();
                w ^= 1;
            }
            cur ^= 1;
        }
        if (stamp) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tB += t - t_prev; t_prev = t; }

        // memory wave: forward storage slot n+1 <- psi(t_{n+1}) (coalesced 16-B stores from LDS).  (Issued at the top of the NEXT step
        // instead -- so that the loop head's vmcnt(0) does not hold the memory wave until the acknowledgement -- it measured slower:
        // C3 5.25 -> 5.37, C4 9.93 -> 10.00 us per step, same box.)
        if (memwave && a.store) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (m_tq[c] >= 0) {
                    const int ii = m_i[c];
                    a.store[(size_t)(n + 1) * slab + tile_base + lane + 64 * c] = sx[cur][m_tq[c]][ii + ii / JS];
                }
            }
        }
        if (stamp) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tS += t - t_prev; t_prev = t; }
    }
    if (stamp && tid == 0) {
        a.debug[0] = tA; a.debug[1] = tX; a.debug[2] = tB; a.debug[3] = tS;
        a.debug[4] = kr_realtime() - rt0; a.debug[5] = tB1; a.debug[6] = tB2;
    }
    if (a.fin && writer && row_ok) {
#pragma unroll
        for (int q = 0; q < KT; ++q) a.fin[tile_base + (size_t)(q * SG + team) * N + i] = make_double2(pr[q], pim[q]);
    }
    if (wg == 0 && tid == 0) {
#pragma unroll
        for (int l = 0; l < LMAX; ++l) if (l < L) a.g_a_int[l] = gint[l];
    }
}

// ---- launcher table for one NP ---------------------------------------------------------------
#include "kr_device_prop.h"
// propagator build: matrix cores for NP = 64 (KROTOV_HIP_EXPM_MFMA=0 at kr_create selects the VALU kernel)
template <int NP, int LM, int CB>
static hipError_t kr_launch_expm(int grid, hipStream_t s, const KrSweepArgs *a) {
    if constexpr (NP == 64) {
        if (a->use_mfma) {
            constexpr size_t lds = (size_t)4 * 64 * KR_YS * sizeof(double);
            // (per kernel AND device: kr_raise_dynamic_lds; on failure the VALU build kernel below carries the sweep)
            if (kr_raise_dynamic_lds((const void *)kr_expm_mfma64<LM>, (int)lds) == hipSuccess) {
                kr_launch_kernel((kr_expm_mfma64<LM>), grid, 256, lds, s, *a);
                return hipSuccess;
            }
        }
    }
    if (a->u_layout == 2) return hipErrorInvalidValue;   // the fragment order of the time-blocked sweep is written by the matrix-core build only
    kr_launch_kernel((kr_expm_batch<NP, LM, CB>), grid, KrGeom<NP, LM, 1>::THREADS, 0, s, *a);
    return hipSuccess;
}

template <int NP>
static hipError_t kr_launch_np(int lmax, int kt, int kind, int grid, hipStream_t s, const KrSweepArgs *a) {
    constexpr int CB = (NP >= 16) ? 4 : 1;
    if (kind == 8) {   // block propagators of the time-blocked sweeps, N <= 32 (a->tb_P: the output)
        if constexpr (NP <= 32) { kr_launch_kernel((kr_tb_products_small<NP>), grid, 256, 0, s, *a); return hipGetLastError(); }
        else return hipErrorInvalidValue;
    }
#define KR_CASE(LM, KTv)                                                                          \
    if (lmax == LM && kt == KTv) {                                                                \
        if (kind == 0) kr_launch_kernel((kr_sweep_known<NP, LM, KTv>), grid, KrGeom<NP, LM, KTv>::THREADS, 0, s, *a); \
        else if ((kind & 15) == 1 && (kind & 128)) {  /* one-XCD placement (P <= 32, no calibration): window variant (wave 1 publishes) or general variant (classic schedule) */ \
            if (kind & 64) kr_launch_kernel((kr_sweep_update<NP, LM, KTv, 1, 0, 1, 1>), grid, KrGeom<NP, LM, KTv>::THREADS, 0, s, *a); \
            else kr_launch_kernel((kr_sweep_update<NP, LM, KTv, 0, 0, 0, 1>), grid, KrGeom<NP, LM, KTv>::THREADS, 0, s, *a); \
        } \
        else if ((kind & 15) == 1 && (kind & 256)) {  /* non-linear controls (POLY): both schedules, both variants, no calibration build, no one-XCD placement */ \
            switch ((kind >> 4) & 5) { \
            case 0: kr_launch_kernel((kr_sweep_update<NP, LM, KTv, 0, 0, 0, 0, 1>), grid, KrGeom<NP, LM, KTv>::THREADS, 0, s, *a); break; \
            case 1: kr_launch_kernel((kr_sweep_update<NP, LM, KTv, 1, 0, 0, 0, 1>), grid, KrGeom<NP, LM, KTv>::THREADS, 0, s, *a); break; \
            case 4: kr_launch_kernel((kr_sweep_update<NP, LM, KTv, 0, 0, 1, 0, 1>), grid, KrGeom<NP, LM, KTv>::THREADS, 0, s, *a); break; \
            default: kr_launch_kernel((kr_sweep_update<NP, LM, KTv, 1, 0, 1, 0, 1>), grid, KrGeom<NP, LM, KTv>::THREADS, 0, s, *a); break; \
            } \
        } \
        else if ((kind & 15) == 1) {  /* update sweep: kind = 1 | SCHED << 4 | CAL << 5 | HERM << 6 */ \
            switch (kind >> 4) { \
            case 0: kr_launch_kernel((kr_sweep_update<NP, LM, KTv, 0, 0, 0>), grid, KrGeom<NP, LM, KTv>::THREADS, 0, s, *a); break; \
            case 1: kr_launch_kernel((kr_sweep_update<NP, LM, KTv, 1, 0, 0>), grid, KrGeom<NP, LM, KTv>::THREADS, 0, s, *a); break; \
            case 2: kr_launch_kernel((kr_sweep_update<NP, LM, KTv, 0, 1, 0>), grid, KrGeom<NP, LM, KTv>::THREADS, 0, s, *a); break; \
            case 3: kr_launch_kernel((kr_sweep_update<NP, LM, KTv, 1, 1, 0>), grid, KrGeom<NP, LM, KTv>::THREADS, 0, s, *a); break; \
            case 4: kr_launch_kernel((kr_sweep_update<NP, LM, KTv, 0, 0, 1>), grid, KrGeom<NP, LM, KTv>::THREADS, 0, s, *a); break; \
            case 5: kr_launch_kernel((kr_sweep_update<NP, LM, KTv, 1, 0, 1>), grid, KrGeom<NP, LM, KTv>::THREADS, 0, s, *a); break; \
            case 6: kr_launch_kernel((kr_sweep_update<NP, LM, KTv, 0, 1, 1>), grid, KrGeom<NP, LM, KTv>::THREADS, 0, s, *a); break; \
            default: kr_launch_kernel((kr_sweep_update<NP, LM, KTv, 1, 1, 1>), grid, KrGeom<NP, LM, KTv>::THREADS, 0, s, *a); break; \
            } \
        } \
        else if (kind == 2) { const hipError_t ee_ = kr_launch_expm<NP, LM, CB>(grid, s, a); if (ee_ != hipSuccess) return ee_; } \
        else if (kind == 7) {  /* known-pulse sweep in blocks of time, N <= 32: the parallel pass (kr_sweep_apply<.., TB>) */ \
            if constexpr (NP <= 32) kr_launch_kernel((kr_sweep_apply<NP, KTv, 1>), grid, KrApplyGeom<NP>::THREADS, 0, s, *a); \
            else return hipErrorInvalidValue; \
        } \
        else kr_launch_kernel((kr_sweep_apply<NP, KTv>), grid, KrApplyGeom<NP>::THREADS, 0, s, *a);      \
        return hipGetLastError();                                                                 \
    }
    KR_CASE(2, 1) KR_CASE(2, 2) KR_CASE(2, 4)
    KR_CASE(4, 1) KR_CASE(4, 2) KR_CASE(4, 4)
#undef KR_CASE
    return hipErrorInvalidValue;
}
