/* krotov_hip.h -- C ABI of libkrotov_hip.so, the MI355X (gfx950) engine for the hot path of
 * Krotov's method: forward propagation -> backward co-state propagation -> sequential
 * pulse-update sweep, batched over an ensemble of trajectories.
 *
 * Reference interfaces replaced (paths under /root/reference):
 *   src/optimize.jl:247-265  krotov_initial_fw_prop!     -> kr_forward
 *   src/optimize.jl:303-317  backward sweep              -> kr_backward
 *   src/optimize.jl:321-370  update + forward sweep      -> kr_update_forward
 *   src/optimize.jl:279-371  krotov_iteration            -> kr_iteration
 *   src/workspace.jl:128-161 storages + fw/bw propagators-> kr_create (device-resident)
 *   src/optimize.jl:332-334  get_from_storage[!]         -> kr_read_storage
 *   src/optimize.jl:297-302,381-386  chi(T) / tau / J_T for the standard functionals
 *                                                        -> kr_iteration_functional (device side)
 * The fine-grained per-trajectory/per-step API the reference consumes (prop_step!,
 * reinit_prop!, write_to_storage!, dot(chi, mu, psi): optimize.jl:251-263,306-315,346,361-367)
 * is NOT exported one-to-one: it would mean 2*K*N_T foreign calls per iteration with a host
 * reduction between steps.  The engine owns whole sweeps.
 *
 * Conventions
 *   - every function returns an int status (KR_OK = 0); nothing throws, aborts or longjmps
 *     across the boundary: every entry point (and every library-owned thread of a device-list
 *     handle) runs inside a catch-all that turns C++ exceptions into KR_ERR_INTERNAL + message.  kr_last_error() gives the message (the Julia shim turns it into an
 *     ErrorException, which optimize.jl:218-226 already captures).
 *   - operators: N*N complex128, COLUMN-major, interleaved (re,im) = Julia Matrix{ComplexF64}.
 *   - states: [K][N] complex128 interleaved, trajectory-major (one Vector{ComplexF64} after the
 *     other).  pulses / update shapes: [L][N_T] float64 (values on the N_T intervals,
 *     workspace.jl:37-43,118-125).
 *   - the caller owns every host array and keeps it alive for the duration of the call
 *     (GC.@preserve).  The library copies in/out, owns all device memory behind the handle and
 *     never returns a device pointer.  Pulses are re-read from the caller on every call
 *     (callbacks may mutate them between iterations: optimize.jl:113-119,
 *     test/test_iterations.jl:125-143); input arrays are never written
 *     (test/test_pulse_optimization.jl:33-35).
 *   - a handle is not re-entrant: one call at a time per handle (calls block until done).
 *   - the *_dev entry points take DEVICE pointers (same layouts) and enqueue on the handle's
 *     stream without synchronising; they exist so that whole iterations chain in HBM.
 */
#ifndef KROTOV_HIP_H
#define KROTOV_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KR_ABI_VERSION 3   /* v2: kr_config grew (exchange, prop_method, device list, max_workgroups); v3: two of v2's reserved words became
                            * spin_timeout_ms and the `options` pointer, which kr_create dereferences -- a v2 caller that never zeroed
                            * `reserved` must not reach that code, so v2 configs are refused (kr_create: KR_ERR_INVALID) */

enum kr_status {
    KR_OK = 0,
    KR_ERR_INVALID = 1,     /* bad argument / configuration (e.g. L = 0: workspace.jl:71-73)   */
    KR_ERR_NO_DEVICE = 2,   /* no usable HIP device -- there is NO CPU fallback                */
    KR_ERR_HIP = 3,         /* a HIP runtime call failed                                       */
    KR_ERR_UNSUPPORTED = 4, /* valid request outside what this build implements                */
    KR_ERR_TIMEOUT = 5,     /* an in-kernel bounded spin gave up (device status word set)      */
    KR_ERR_STATE = 6,       /* call order violated (e.g. update before backward)               */
    KR_ERR_INTERNAL = 7     /* a C++ exception (std::bad_alloc, std::system_error, ...) was caught at the
                               boundary (the message names it; nothing propagates into the caller), or an
                               integrity check of a sweep failed: the generator pack on the device no longer
                               matches what kr_create uploaded, or the workgroups that share a trajectory
                               ended a row-split sweep with different copies of its state vector.  The
                               results of that call are invalid; the message says which check.           */
};

enum kr_flags {
    KR_STORE_FW = 1u << 0, /* keep forward storage Phi (workspace.jl:128), slot n+1 = state after step n */
    KR_NO_STORE_FW = 1u << 1
};

enum kr_functional { /* standard final-time functionals, weights 1 [QuantumControl.Functionals] */
    KR_J_T_SM = 0,     /* 1 - |1/K sum tau|^2 ; chi_k = (sum tau)/K^2 * tgt_k */
    KR_J_T_SS = 1,     /* 1 - 1/K sum |tau|^2 ; chi_k = tau_k/K * tgt_k       */
    KR_J_T_RE = 2      /* 1 - Re(1/K sum tau) ; chi_k = tgt_k/(2K)            */
};

enum kr_which_storage { KR_STORAGE_FW = 0, KR_STORAGE_BW = 1 };

/* How the per-step sum of the L update integrands (optimize.jl:339-349) crosses GPUs / ranks.                */
enum kr_exchange {
    KR_EXCHANGE_AUTO = 0,     /* mailboxes where the persistent kernels can run, else RCCL                     */
    KR_EXCHANGE_MAILBOX = 1,  /* in-kernel peer mailboxes over xGMI (one launch per sweep)                     */
    KR_EXCHANGE_RCCL = 2,     /* ncclAllReduce per time step between per-step kernels (the baseline BASELINE.json
                                 names; also the fallback when mailboxes cannot be wired)                     */
    KR_EXCHANGE_CALLBACK = 3  /* a host function sums the L doubles per step (MPI, gloo, ...)                  */
};

/* Step propagator of the sweeps (the reference's prop_method, optimize.jl:108,137-153).                      */
enum kr_prop_method {
    KR_PROP_AUTO = 0,    /* Taylor in the update sweep, batched exact propagators in the known-pulse sweeps    */
    KR_PROP_TAYLOR = 1,
    KR_PROP_CHEBY = 2,   /* Chebyshev expansion; Hermitian generators only.  With a persistent plan: the same
                            truncation in powers of the generator, per-step spectral bound (persistent kernels);
                            else / option CHEBY_RECURRENCE=1: the three-term recurrence (generic kernels)         */
    KR_PROP_NEWTON = 3   /* Newton interpolation at Leja points on the spectral bounding box                   */
};

typedef struct kr_config {
    int32_t abi_version; /* = KR_ABI_VERSION */
    int32_t N;           /* state-vector length (Hilbert or Liouville dimension)              */
    int32_t NT;          /* number of time intervals = length(tlist)-1 (optimize.jl:255,282)   */
    int32_t K;           /* number of trajectories LOCAL to this handle (optimize.jl:283)      */
    int32_t L;           /* number of controls (optimize.jl:284); 0 is an error                */
    int32_t G;           /* number of distinct generators (1 = shared by all trajectories)    */
    const int32_t *traj_generator; /* [K] k -> g, or NULL when G == 1                          */
    const double *tlist;           /* [NT+1]                                                   */
    const double *H0;              /* [G][N*N] drift, column-major complex                     */
    const double *Hc;              /* [G][L][N*N] control operators = dH/d eps_l (workspace.jl:74) */
    const uint8_t *hc_present;     /* [G][L] 0 where the derivative is `nothing` (optimize.jl:344), or NULL */
    uint32_t flags;                /* kr_flags                                                 */
    double prop_tol;               /* truncation bound of the step propagator per step; <=0 -> 1e-16 */
    int32_t device;                /* HIP device ordinal, -1 = current                         */
    /* multi-GPU (one process per GPU; trajectories sharded, K above is the local count)      */
    int32_t rank;                  /* 0 .. world-1 */
    int32_t world;                 /* 1 = single GPU */
    int32_t K_global;              /* total trajectories over all ranks (0 -> K)               */
    /* --- ABI v2 --- */
    int32_t exchange;              /* kr_exchange                                              */
    int32_t prop_method;           /* kr_prop_method                                           */
    int32_t max_workgroups;        /* cap on co-resident persistent workgroups of this handle (0 = derive from the
                                      device's occupancy; several handles sharing one GPU must split its CUs -- XCD
                                      by XCD: a launch's workgroups are dealt out to the 8 XCDs in turn, so the caps
                                      of handles that run at the same time must satisfy sum ceil(cap / 8) <= 32, the
                                      CUs of one XCD; a device list that repeats an ordinal does that by itself)   */
    int32_t n_devices;             /* > 1: ONE handle drives n_devices GPUs from this process (trajectories block-
                                      partitioned over them; rank/world must be 0/1); 0 or 1: `device` alone     */
    int32_t devices[8];            /* HIP device ordinals of the n_devices GPUs (the same ordinal may repeat)     */
    /* non-linear controls, static part: H(eps) = H0 + sum_l f_l(eps_l) H_l with polynomials
     * f_l(x) = sum_p control_poly[l][p] x^p.  The control derivative is then mu_l = f_l'(eps_l) H_l, evaluated at the GUESS
     * value of the step as _eval_mu does (optimize.jl:268-272,337,345).  degree 0 / NULL = linear controls (f_l(x) = x).   */
    int32_t control_poly_degree;   /* 0 (linear) or 1..7                                         */
    int32_t pad_;
    const double *control_poly;    /* [L][control_poly_degree + 1], or NULL                      */
    /* Bound of every in-kernel spin (per-step exchange between workgroups / GPUs, row-split all-gather, cross-rank tau sum)
     * in milliseconds; 0 = the default 4000.  With one process per GPU the ranks run unsynchronised between iterations, so
     * a rank whose host-side chi / callback takes longer than this bound makes the OTHER ranks' sweeps give up with
     * KR_ERR_TIMEOUT: size it to the slowest host step.  (ABI v3; the first of v2's reserved words.)                       */
    int32_t spin_timeout_ms;
    int32_t reserved0;
    /* Schedule and diagnostics options, "NAME=integer;NAME=integer" (separators ; , or white space), or NULL.  The names are those
     * INTEGRATION.md lists as KROTOV_HIP_<NAME> (the prefix is optional here); an option given here takes precedence over the
     * environment variable of the same name, which stays the fallback for the lab.  kr_create fails with KR_ERR_INVALID on a name
     * it does not know or a value that is not an integer.  The string is parsed inside kr_create and not kept.  (ABI v3; occupies
     * words that v2 called reserved, hence the version bump: kr_create refuses a v2 config instead of reading a wild pointer.)      */
    const char *options;
    int32_t reserved[4];
} kr_config;

typedef struct kr_handle kr_handle;

/* Creation mirrors KrotovWrk(problem) (workspace.jl:65-200) for what lives on the device. */
int kr_create(const kr_config *cfg, kr_handle **out);
int kr_destroy(kr_handle *h);
const char *kr_last_error(const kr_handle *h); /* h may be NULL: error of the last failed kr_create */
int kr_abi_version(void);

/* Use an existing HIP stream (e.g. torch's current stream); NULL = the handle's own stream. */
int kr_set_stream(kr_handle *h, void *hip_stream);

/* trajectories[k].initial_state (optimize.jl:323); targets only for the device functionals.  */
int kr_set_initial_states(kr_handle *h, const double *psi0 /* [K][N] */);
int kr_set_target_states(kr_handle *h, const double *tgt /* [K][N] */);

/* a1: psi_k(t_{n+1}) = U_n(eps) psi_k(t_n); psiT_out may be NULL.                             */
int kr_forward(kr_handle *h, const double *eps /* [L][NT] */, double *psiT_out /* [K][N] */);
/* a3: X[NT] = chiT; X[n] = U_n(eps)^dagger X[n+1]; X stays on the device.                     */
int kr_backward(kr_handle *h, const double *eps_guess, const double *chiT /* [K][N] */);
/* a4-a7: needs a preceding kr_backward.  eps_new_out / g_a_int_out / psiT_out are outputs.     */
int kr_update_forward(kr_handle *h, const double *eps_guess, const double *S /* [L][NT] */,
                      const double *lambda /* [L] */, double *eps_new_out /* [L][NT] */,
                      double *g_a_int_out /* [L] */, double *psiT_out /* [K][N] or NULL */);
/* krotov_iteration given chi(T): backward + update.                                           */
int kr_iteration(kr_handle *h, const double *eps_guess, const double *chiT, const double *S,
                 const double *lambda, double *eps_new_out, double *g_a_int_out, double *psiT_out);

/* Device-resident chain for the standard functionals: tau, J_T and chi(T) are
 * computed on the device from the final states of the previous sweep.  eps_dev_in/out: DEVICE
 * pointers [L][NT]; S/lambda host.  out_host (may be NULL) receives {J_T_before, g_a_int[0..L-1]}
 * only when sync != 0.                                                                         */
int kr_set_update_params(kr_handle *h, const double *S /* [L][NT] */, const double *lambda /* [L] */);
int kr_forward_dev(kr_handle *h, const void *eps_dev);
int kr_iteration_functional_dev(kr_handle *h, int functional, const void *eps_guess_dev,
                                void *eps_new_dev);
/* tau_k = <tgt_k|psi_k(T)> and J_T of the states left by the last forward/update sweep.       */
int kr_final_states(kr_handle *h, double *psiT_out /* [K][N] */);
int kr_taus(kr_handle *h, double *tau_out /* [K] complex */);
int kr_g_a_int(kr_handle *h, double *g_a_int_out /* [L] */);
int kr_synchronize(kr_handle *h);

/* Callback inspection (workspace.jl:24; optimize.jl:258-262): state k at time index n (0..NT). */
int kr_read_storage(kr_handle *h, int which, int k, int n, double *out /* [N] */);

/* Introspection for bench / tests.  info_out (16 ints): {NP, n_workgroups, traj_per_wg, threads,
 * KT, LMAX variant, K padded, store_fw, JQ, precomputed propagators, row-split R, exchange poll delay
 * (units of 128 cycles, -1 = classic schedule; measured once by kr_create),
 * Hermitian-controls kernel variant in use, matrix-core update sweep, generic (multi-pass) known sweeps,
 * [15] = generic / stepped update sweep (bit 0) | exchange in use (kr_exchange, bits 4-5) | economized step polynomial in the
 * update sweep (bit 8: all generators Hermitian)}.
 * kr_enable_timing(h, 1) starts a window in which every sweep-kernel launch is bracketed by HIP
 * events on the handle's stream; kr_get_timings syncs the stream and returns, per kind
 * {forward, backward, update}: {sum of kernel durations in ms, number of launches} (6 doubles). */
int kr_get_info(kr_handle *h, int32_t *info_out /* [16] */);
/* One human-readable line: which kernels carry this handle's sweeps, grid, exchange, spin bound -- and, when the problem left the
 * persistent kernels (capacity, co-residency, RCCL / callback exchange) or something could not be verified (occupancy query),
 * why.  Valid until kr_destroy; "" for NULL.                                                                              */
const char *kr_plan_description(const kr_handle *h);
int kr_get_timings(kr_handle *h, double *timings_ms /* [6] */);
int kr_enable_timing(kr_handle *h, int on);
/* Diagnostics (env KROTOV_HIP_STAMPS=1 at kr_create): workgroup 0's accumulated shader cycles of the last
 * update sweep per phase {matvec+dot, exchange, Taylor, store} and the sweep's length in 100 MHz
 * ticks; out: 8 x uint64.  Zero when stamps are off.                                              */
int kr_debug_stamps(kr_handle *h, unsigned long long *out);
/* Diagnostic builds of the library only (-DKR_ARRIVALS): per workgroup {publish, end of window work, completion, poll passes}
 * of two exchanges in the middle of the last update sweep, 100 MHz ticks; zeros otherwise.  words <= 8192. */
int kr_debug_arrivals(kr_handle *h, unsigned long long *out, int words);
/* The bounds of the operators' spectral norms that set the degree of every step polynomial (Taylor remainder, interval of the
 * economized polynomial): out[g][0] >= ||H0_g||_2, out[g][1 + l] >= ||H_l,g||_2, [G][1 + L] doubles.  Rigorous upper bounds, at
 * most 0.3 % above the norms (kr_norm.hip).  A handle with a device list answers for its first shard (all shards hold the same
 * operators).                                                                                                               */
int kr_get_norm_bounds(kr_handle *h, double *out /* [G][1 + L] */);

/* Per-step exchange through RCCL (KR_EXCHANGE_RCCL): rank 0 makes a unique id, the host side broadcasts it, every
 * rank calls kr_rccl_init (collective).  librccl is loaded at run time (dlopen): no link-time dependency.
 * A handle with a device list (n_devices > 1) sets its communicators up by itself.                              */
int kr_rccl_unique_id_size(void);
int kr_rccl_get_unique_id(void *id_out /* kr_rccl_unique_id_size() bytes */);
int kr_rccl_init(kr_handle *h, const void *unique_id);
/* Per-step exchange through a host function (KR_EXCHANGE_CALLBACK): fn must sum buf[0..count-1] over all ranks in place
 * (the same result on every rank) and return 0; it is called once per time step from the calling thread.        */
typedef int (*kr_allreduce_fn)(void *ctx, double *buf, int count);
int kr_set_exchange_callback(kr_handle *h, kr_allreduce_fn fn, void *ctx);

/* Multi-GPU mailbox wiring (in-kernel per-step exchange of the L partial sums).
 * Each rank exports an IPC handle of its mailbox, the host side all-gathers them
 * (torch.distributed), each rank then connects to all peers.                                   */
int kr_ipc_handle_size(void);
int kr_ipc_export(kr_handle *h, void *handle_out /* kr_ipc_handle_size() bytes */);
int kr_ipc_connect(kr_handle *h, const void *all_handles /* [world][kr_ipc_handle_size()] */);
#ifdef __cplusplus
}
#endif
#endif /* KROTOV_HIP_H */
