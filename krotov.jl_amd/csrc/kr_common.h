// Shared host/device declarations of the Krotov sweep kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define KR_MAX_THREADS 512      // NP=64 uses 512-thread workgroups (2 waves/SIMD), smaller NP 256
#define KR_LMAX_CAP 4           // largest L of the register-resident kernels
#define KR_THETA_MAX 1.0        // largest ||H dt|| handled by one Taylor sub-step
#define KR_MAX_ORDER 48
#define KR_MAIL_REPLICAS 8      // copies of the per-step mailbox, one per XCD (kr_exchange)
#define KR_MAIL_RSTRIDE (65536 / 8)   // unsigned long longs between replicas
#define KR_RS_MAX_TURNS 16     // update sweep in turns (kr_rowsplit_turns_update): trajectories a group of workgroups carries at most
#define KR_ECON_W 32           // doubles per row of the economized-polynomial table: [0] a0, [1] a1, [j] a_j / a_{j-1} (2 <= j <= m <= 29), [30] theta_c, [31] m
#ifndef KR_APPLY_JQ64
#define KR_APPLY_JQ64 8          // lanes per row of the apply sweep at NP = 64 (8: 512-thread workgroups)
#endif
#define KR_SPIN_DEFAULT_TICKS 400000000ull  // 4 s of s_memrealtime (100 MHz): default bound of every in-kernel spin
                                            // (kr_config.spin_timeout_ms overrides it per handle: KrSweepArgs::spin_ticks)

// status words (device): [0] abort/timeout flag, [1] where, [2] step, [3] spare
// (a device clock stamp of the sending wave at the first exchange was tried in round 4 for the timeout report of device lists: it cost the
//  C3 update sweep 3 % -- 5.17 -> 5.32 us per step, same-box A/B -- although the branch is never taken on one GPU; removed)
#define KR_STATUS_WORDS 4

struct KrSweepArgs {
    // generator slices, thread-major: Hpack[g][m][jj][tid_in_team] (double2), g = generator, m = 0..L
    const double2 *Hpack;
    size_t pack_stride;      // elements between two generators' packs
    // the same matrices dense, row-major, zero-padded to 64 x 64: Hdense[g][m][row][col] (kr_device_mfma.h, NP = 64 only)
    const double2 *Hdense;
    size_t dense_stride;     // elements between two generators
    int u_layout;            // 0: propagators packed for kr_sweep_apply (VALU); 1: as MFMA A operands for kr_mf_apply_sweep (kr_device_mfma.h)
    int gen_div;             // kr_sweep_apply: wg_gen is indexed by blockIdx.x / gen_div (apply tiles may be narrower than update tiles)
    const int *wg_gen;       // [P] generator of each workgroup's tile (nullptr: all 0)
    const double *nrm_dev;   // [G][1 + KR_LMAX_CAP] 2-norm bounds of H0, H_1..H_L per generator
    const unsigned *present_dev; // [G] bit l set: control l has a derivative in generator g (optimize.jl:344)
    const double *eps;       // [L][NT] pulses that drive this sweep (known-pulse sweeps) / guess (update)
    const double *dt;        // [NT]
    const double2 *init;     // [Kp][N] initial states (psi0, or chi(T) for the backward sweep)
    double2 *store;          // [(NT+1)][Kp][N] storage to write (or nullptr)
    double2 *fin;            // [Kp][N] final states (or nullptr)
    // update sweep only
    const double2 *X;        // [(NT+1)][Kp][N] co-states from the backward sweep
    const double *alpha;     // [L][NT] = S[l][n] / lambda[l] (optimize.jl:354), divided once on the host
    double *eps_new;         // [L][NT]
    double *g_a_int;         // [L]
    int xcd_group;             // row-split: the R workgroups of a trajectory have equal blockIdx % 8 (one XCD under round-robin placement)
    int xpoll_delay;           // world > 1: extra wait (x 128 cycles) before a workgroup starts polling xmail
    int gather_delay_local;    // the same when the all-gather stays inside one XCD's L2
    int gather_delay;          // row-split all-gather: the polling wave starts after gather_delay x 128 cycles
    // Economized step polynomial of the update sweep (Hermitian generators; kr_device.h: KrEcon): rows [G][NT][KR_ECON_W] written by
    // kr_econ_rows_kernel (krotov_hip.hip) from the guess pulses in front of every update sweep; nullptr = Taylor everywhere.
    const double *econ;
    int poll_delay;            // exchange schedule 1 (kr_exchange): wave 0 starts polling after poll_delay x 128 cycles
    unsigned long long *mail;  // [2][P][2*LMAX] granules {tag32 | half of a double}
    unsigned long long *xmail; // [2][world][2*LMAX] inter-GPU granules (local memory, peers write)
    unsigned long long *peer_xmail[8]; // every rank's xmail (peer_xmail[rank] == xmail)
    double2 *U;              // [NT][NP*NP] precomputed step propagators, thread-major (kr_device_prop.h), or nullptr
    unsigned long long *gmail; // row-split kernels: per-trajectory vector mailbox [tiles][2][NP][4] granules
    int rs_fault;              // diagnostics (option RS_FAULT=w): workgroup w of a row-split launch flips one bit of its copy of the final state vector, so that
                               // the coherence check of the group (kr_rowsplit_check) has something to find (tests; w = 1: tile 0, w = R + 1: tile 1, ...);
                               // option PULSE_FAULT sets it to 2 for the N <= 64 kernels: workgroup 1 logs a wrong total at step 3 (kr_pulse_finish)
    int rs_sets;               // row-split PAIR mode: pairs a group of workgroups carries in turn (1: one pair per group)
    double2 *rs_scratch;       // ... and the per-workgroup scratch of their state vectors [P][KR_PAIR_MAX_SETS][2][NP] (rs_sets > 1)
    unsigned epoch_base2;    // tag base of the vector mailbox for this launch
    unsigned *status;        // [KR_STATUS_WORDS]
    unsigned long long *debug; // nullptr, or [8]: workgroup 0's cycle counts per phase (diagnostics only)
    double tol;
    double thr[KR_MAX_ORDER];    // thr[m] = largest theta <= KR_THETA_MAX for which Taylor order m meets tol
    int N, NT, Kp, L;
    int dir;                 // +1 forward (store slot n+1), -1 backward (store slot n)
    int P;                   // workgroups
    int rank, world;
    unsigned epoch_base;     // tags = epoch_base + n + 1
    unsigned long long spin_ticks; // bound of every in-kernel spin in s_memrealtime ticks (100 MHz); kr_config.spin_timeout_ms
    int use_mfma;            // propagator build on the matrix cores where available (NP = 64)
    // One-XCD placement of a small update grid (kr_sweep_update, P <= 32): the launch has wg_stride * P blocks, block b works as
    // workgroup b / wg_stride iff b % wg_stride == 0 (the others exit at once), so that -- workgroups being dealt round-robin
    // over the 8 XCDs -- all P working groups land on ONE XCD and their per-step exchange stays in that XCD's L2 (sc0 stores,
    // sc1 loads: 295 ns per hop against 626 ns across XCDs, profiles/r02_ubench_pingpong.txt).  Confirmed in the kernel from
    // HW_REG_XCC_ID; any other placement keeps the replicated write-through mailbox.  1 = ordinary launch.
    int wg_stride;
    // Log of the per-step totals (round 5; kr_sweep_update, kr_mf_update, kr_stream_update): every workgroup forms eps_new for ITSELF from
    // the totals du_l it gathered; one lane of an otherwise idle wave stores them, dulog[(wg NT + n) LMAX + l].  After the sweep
    // kr_pulse_finish forms the caller's eps_new from workgroup 0's log -- the kernels themselves no longer write it -- and compares all P
    // logs bit for bit (status code 8): a workgroup that summed a stale or lost partial sum has propagated with another pulse than its
    // neighbours.  optimize.jl:351-358: ONE pulse value per step, whoever consumes it.
    double *dulog;
    // Non-linear controls, static part (round 5 on the persistent update kernels; kr_config.control_poly): H(eps) = H0 + sum_l f_l(eps_l) H_l
    // with polynomial f_l of degree <= 3, mu_l = f_l'(eps_guess) H_l (_eval_mu, optimize.jl:268-272,337,345).  The host hands the update
    // sweep what depends on the GUESS only -- amp_g[l][n] = f_l(eps_guess), fprime_g[l][n] = f_l'(eps_guess), both [L][NT] -- and the
    // coefficients poly[l][0..3] for the one evaluation per step that cannot be made in advance, f_l(eps_new).  (Known-pulse sweeps need
    // nothing: they are launched with the amplitudes f_l(eps) in the place of the pulses.)  All three nullptr: linear controls.
    // The row-split update kernels (kr_device_rowsplit.h; run-time branch, no instantiation of their own) take the roles the other way
    // round: `eps` = f_l(eps_guess) -- what their resident H(eps) blocks are built from -- and `amp_g` = the guess PULSES.
    const double *amp_g, *fprime_g, *poly;
    // Known-pulse sweeps in blocks of time (round 5; kr_device_tblock.h, NP = 64): steps [b S, (b + 1) S) form block b; tb_bound holds
    // the state at every block boundary ([NB + 1][Kp][N], written by the sequential apply sweep over the block propagators),
    // tb_items the trajectory ranges one workgroup carries ({first slot, count <= 64, generator, 0} per item).
    const double2 *tb_bound;
    const int *tb_items;
    int tb_S, tb_nitems, tb_NT;
    // N <= 32 (kr_device_prop.h: kr_tb_products_small, kr_sweep_apply<.., TB>): number of blocks, output of the block products;
    // tb_nitems = the apply kernel's tiles per block there
    int tb_NB;
    double2 *tb_P;
};
#define KR_MAIL_XTAB_OFF ((size_t)2 * KR_MAIL_REPLICAS * KR_MAIL_RSTRIDE + 64 * 4)   // words: XCC-id table behind the mailbox + tail
#define KR_MAIL_XTAB_WORDS 64
// buffer and validity phase of the mailbox inside the GPU at step n of a launch (kr_exchange; the timeout report of krotov_hip.hip
// reads the mailbox with the same two expressions)
__host__ __device__ inline int kr_mail_buffer(const unsigned n) { return (int)(n & 1u); }
__host__ __device__ inline unsigned kr_mail_phase(const unsigned n) { return ((n >> 1) & 1u) ^ 1u; }

// Taylor plan: sub-steps s and order m so that the truncation bound of exp(A), ||A|| <= theta,
// is <= tol per step.  Same code on host and device -> identical plans everywhere.
__host__ __device__ inline void kr_taylor_plan(double theta, double tol, int &s, int &m) {
    s = 1;
    if (theta > KR_THETA_MAX) s = (int)ceil(theta / KR_THETA_MAX);
    const double th = theta / (double)s;
    const double budget = tol / (double)s;
    double term = th;  // th^1/1!
    m = 1;
    for (; m < KR_MAX_ORDER; ++m) {
        // remainder after m terms <= th^(m+1)/(m+1)! / (1 - th/(m+2))
        const double next = term * th / (double)(m + 1);
        const double rem = next / (1.0 - th / (double)(m + 2));
        if (rem <= budget) break;
        term = next;
    }
}

// Per-workgroup generator context (per-trajectory generators: control_derivs[k][l], workspace.jl:74)
template <int LMAX>
struct KrGen {
    const double2 *Hpack;
    double nrm[1 + LMAX];
    unsigned present;
    int gen;
};
template <int LMAX>
__device__ __forceinline__ KrGen<LMAX> kr_gen_ctx(const KrSweepArgs &a, const int gen) {
    KrGen<LMAX> g;
    g.gen = gen;
    g.Hpack = a.Hpack + (size_t)gen * a.pack_stride;
#pragma unroll
    for (int m = 0; m <= LMAX; ++m) g.nrm[m] = a.nrm_dev[(size_t)gen * (1 + KR_LMAX_CAP) + m];
    g.present = a.present_dev[gen];
    return g;
}

// 1/j, j = 0..KR_MAX_ORDER (index 0 unused): the Horner coefficients without an fp64 division
__device__ __constant__ static const double kr_rinv[KR_MAX_ORDER + 1] = {
    0.0, 1.0, 1.0 / 2, 1.0 / 3, 1.0 / 4, 1.0 / 5, 1.0 / 6, 1.0 / 7, 1.0 / 8, 1.0 / 9, 1.0 / 10, 1.0 / 11, 1.0 / 12,
    1.0 / 13, 1.0 / 14, 1.0 / 15, 1.0 / 16, 1.0 / 17, 1.0 / 18, 1.0 / 19, 1.0 / 20, 1.0 / 21, 1.0 / 22, 1.0 / 23,
    1.0 / 24, 1.0 / 25, 1.0 / 26, 1.0 / 27, 1.0 / 28, 1.0 / 29, 1.0 / 30, 1.0 / 31, 1.0 / 32, 1.0 / 33, 1.0 / 34,
    1.0 / 35, 1.0 / 36, 1.0 / 37, 1.0 / 38, 1.0 / 39, 1.0 / 40, 1.0 / 41, 1.0 / 42, 1.0 / 43, 1.0 / 44, 1.0 / 45,
    1.0 / 46, 1.0 / 47, 1.0 / 48};

// Device-side plan without memory accesses on the critical path: lane k of every wave keeps thr[k] in a
// register (`my_thr`, loaded once at kernel start); for a wave-uniform theta the order is one compare, one
// ballot and a popcount:  m = 1 + #{1 <= k : theta > thr[k]}  (thr is increasing in k).
__device__ __forceinline__ double kr_load_my_thr(const KrSweepArgs &a) {
    const int lane = threadIdx.x & 63;
    return a.thr[lane < KR_MAX_ORDER ? lane : KR_MAX_ORDER - 1];
}
__device__ __forceinline__ void kr_plan_fast(const KrSweepArgs &a, const double my_thr, const double theta, int &s, int &m) {
    if (theta <= KR_THETA_MAX) {
        const int lane = threadIdx.x & 63;
        const unsigned long long above = __ballot(lane >= 1 && lane < KR_MAX_ORDER - 1 && theta > my_thr);
        s = 1;
        m = 1 + __popcll(above);
    } else {
        kr_taylor_plan(theta, a.tol, s, m);
    }
}

// Launch, or -- in query mode -- answer how many workgroups of this kernel one CU can hold (kr_create's co-residency check:
// the persistent kernels wait for each other, so the whole grid must be resident at once).
struct KrOccQuery { int on, blocks_per_cu; };
KrOccQuery &kr_occ_query();   // thread-local (krotov_hip.hip)
template <class Kern>
static inline void kr_launch_kernel(Kern k, int grid, int threads, size_t lds, hipStream_t s, const KrSweepArgs &a) {
    KrOccQuery &q = kr_occ_query();
    if (q.on) {
        int nb = 0;
        q.blocks_per_cu = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k, threads, lds) == hipSuccess) ? nb : -1;
        return;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, s, a);
}

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the code object loaded on ONE device: raise it once per (kernel, device
// that is current), under a lock -- a handle with a device list launches from several threads on several GPUs (krotov_hip.hip).
hipError_t kr_raise_dynamic_lds(const void *func, int bytes);

typedef hipError_t (*kr_launch_fn)(int variant_lmax, int variant_kt, int kind, int grid,
                                   hipStream_t stream, const KrSweepArgs *args);
// kind: 0 = known-pulse sweep (Taylor), 1 = update sweep, 2 = batched propagator build (grid = NT),
//       3 = known-pulse sweep with precomputed propagators, 4 = update sweep with the delayed-poll exchange schedule
//       (N <= 64 only), 5 / 6 = kinds 1 / 4 as kr_sweep_update<..., CAL = 1> (calibration launches),
//       7 / 8 = time-blocked known-pulse sweeps at N <= 32: parallel apply pass / block propagators (kr_device_prop.h)
hipError_t kr_launch_np8(int lmax, int kt, int kind, int grid, hipStream_t s, const KrSweepArgs *a);
hipError_t kr_launch_np16(int lmax, int kt, int kind, int grid, hipStream_t s, const KrSweepArgs *a);
hipError_t kr_launch_np32(int lmax, int kt, int kind, int grid, hipStream_t s, const KrSweepArgs *a);
hipError_t kr_launch_np64(int lmax, int kt, int kind, int grid, hipStream_t s, const KrSweepArgs *a);
hipError_t kr_launch_np128(int lmax, int kt, int kind, int grid, hipStream_t s, const KrSweepArgs *a);
hipError_t kr_launch_np256(int lmax, int kt, int kind, int grid, hipStream_t s, const KrSweepArgs *a);
// matrix-core update sweep for NP = 64 (kr_device_mfma.h): second argument = 4-column tiles per workgroup (2 trajectories each)
hipError_t kr_launch_mf64_l2(int tiles, int kind, int grid, hipStream_t s, const KrSweepArgs *a);
hipError_t kr_launch_mf64_l4(int tiles, int kind, int grid, hipStream_t s, const KrSweepArgs *a);
static inline hipError_t kr_launch_mf64(int lmax, int tiles, int kind, int grid, hipStream_t s, const KrSweepArgs *a) {
    return lmax <= 2 ? kr_launch_mf64_l2(tiles, kind, grid, s, a) : kr_launch_mf64_l4(tiles, kind, grid, s, a);
}
// row-split kernels (kr_device_rowsplit.h): second argument = rows per workgroup
hipError_t kr_launch_rs128(int lmax, int nr, int kind, int grid, hipStream_t s, const KrSweepArgs *a);
hipError_t kr_launch_rs256(int lmax, int nr, int kind, int grid, hipStream_t s, const KrSweepArgs *a);
hipError_t kr_launch_rs512(int lmax, int nr, int kind, int grid, hipStream_t s, const KrSweepArgs *a);
hipError_t kr_launch_rs1024(int lmax, int nr, int kind, int grid, hipStream_t s, const KrSweepArgs *a);

// known-pulse sweeps in blocks of time (kr_device_tblock.h, NP = 64): kind 0 = block propagators into Pout (grid = generators x blocks),
// kind 1 = all blocks in parallel (grid = blocks x items)
hipError_t kr_launch_tblock(int kind, int grid, hipStream_t s, const KrSweepArgs *a, double2 *Pout);
// ... and the block propagators as a tree of pairwise products, one level per launch (blocks of 2^p steps)
hipError_t kr_launch_tbpair(int generators, int n_in, int level0, int final, hipStream_t s, const KrSweepArgs *a, const double2 *in, double2 *out);

// generic (multi-pass) kernels (kr_device_generic.h)
struct KrGenArgs {
    const double2 *H;          // [G][1 + L][N * N] column-major (forward pack, or the adjoint pack for dir = -1)
    size_t gen_stride;         // elements between two generators
    double2 *Heps;             // [G][N * N] scratch: H(eps) of the current step (update sweep)
    const int *tile_gen;       // [tiles] generator of each tile (nullptr: all 0)
    const double *econ;        // rows of the economized step polynomial [G][NT][KR_ECON_W] (Hermitian generators, Taylor method), or nullptr
    const double *nrm;         // [G][1 + L] 2-norm bounds
    const unsigned *present;   // [G] presence bits (L <= 32)
    const double *eps;         // [L][NT] pulses of this sweep (known) / guess pulses (update)
    const double *alpha;       // [L][NT]
    double *eps_new;           // [L][NT]
    double *g_a_int;           // [L] running integrals (device)
    const double *dt;          // [NT]
    const double2 *init;       // [Kp][N]
    double2 *state;            // [Kp][N] running state of the update sweep
    double2 *store;            // [(NT+1)][Kp][N] or nullptr
    double2 *fin;              // [Kp][N] or nullptr
    const double2 *X;          // [(NT+1)][Kp][N]
    double *partial;           // [tiles][L]
    double *du_local;          // [L]
    const double *du_total;    // [L]
    double tol;
    double thr[KR_MAX_ORDER];
    int N, NT, Kp, L, G, KTG, RP, JQ, tiles, dir, n;
    int fused;                 // stepped update sweep on one rank: kr_gen_form sums the partials itself, kr_gen_apply adds the next step's partial sums
    // step propagator (kr_prop_method): per-step tables of the polynomial methods (kr_gen_propagate)
    int method, ptab_stride;
    const double *ptab;
    const double *poly;        // non-linear controls: [L][poly_deg + 1] coefficients of f_l, or nullptr (f_l(x) = x)
    int poly_deg;
    const double *erange;      // [L] control range the polynomial methods were initialised for (update sweep), or nullptr
    unsigned *status;
};
// values of KrGenArgs::method = enum kr_prop_method of include/krotov_hip.h (device code does not see that header)
#define KR_METHOD_CHEBY 2
#define KR_METHOD_NEWTON 3

hipError_t kr_launch_generic(int kind, int ktg, int grid, size_t lds_bytes, hipStream_t s, const KrGenArgs *a);
