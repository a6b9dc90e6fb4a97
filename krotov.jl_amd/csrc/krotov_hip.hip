// libkrotov_hip.so -- host side of the C ABI declared in include/krotov_hip.h.
//
// Owns: device memory (generator packs, pulses, storages X / Phi, states, mailboxes), the
// launch plan (which kernel variant, how many workgroups, trajectories per workgroup), the
// stream, and error reporting.  There is NO CPU fallback: without a HIP device kr_create fails
// with KR_ERR_NO_DEVICE.
//
// Reference anchors (under /root/reference): what KrotovWrk builds per trajectory
// (workspace.jl:128-161: three storages, fw and bw propagators from the trajectory and its
// adjoint) is built here once per handle for all K trajectories.
#include "../../include/krotov_hip.h"
#include "kr_common.h"
#include <rccl/rccl.h>   // types only: librccl is loaded with dlopen (kr_rccl_*), never linked
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <new>
#include <thread>
#include <chrono>
#include <string>
#include <tuple>
#include <vector>

typedef std::complex<double> zc;

static thread_local std::string g_create_error;
KrOccQuery &kr_occ_query() { static thread_local KrOccQuery q{0, 0}; return q; }

struct kr_handle {
    // a handle with a device list (kr_config.n_devices > 1) is a GROUP: one shard handle per GPU (rank i of world P), driven
    // from library-owned threads; all other fields then belong to the shards
    std::vector<kr_handle *> subs;
    std::vector<int> k_off;   // [P + 1] first trajectory of every shard
    kr_config cfg{};
    std::vector<std::pair<std::string, int>> options;   // kr_config.options, parsed
    std::string err;
    std::mutex mtx;
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // plan
    int NP = 0, LMAXv = 0, KT = 0, SG = 0, KTILE = 0, P = 0, Kp = 0, threads = 0, JQ = 4, JS = 0, RT = 0;
    kr_launch_fn launch = nullptr;
    // row-split mode (64 < N <= 256): R workgroups of 512 threads share one trajectory (kr_device_rowsplit.h)
    bool rowsplit = false;
    int rs_sets = 1;           // pair mode: pairs per group of workgroups, in turn (kr_rowsplit_pair_sweep<.., MULTI>)
    double2 *rs_scratch_d = nullptr;
    int rs_batch = 0;          // > 0: N > 256 with more trajectories than one row-split launch carries -- the KNOWN-pulse sweeps run the
                               // row-split kernel in batches of rs_batch trajectories (independent launches), the update sweep is the stepped
                               // generic one (its per-step sum needs every trajectory)
    int rs_turns = 0;          // > 0 (with rs_batch): the update sweep runs kr_rowsplit_turns_update, every group of R workgroups carrying
                               // rs_turns trajectories in turn (kr_device_rowsplit.h); 0 with rs_batch > 0: stepped generic update
    bool rs_pair = false;      // row-split PAIR mode (kr_device_rowsplit2.h): R workgroups share two trajectories and interleave them
    int R = 1, NR = 0;
    kr_launch_fn launch_rs = nullptr;
    unsigned long long *gmail_d = nullptr;
    size_t gmail_words = 0;
    bool xcd_group = false;  // row-split: workgroup ids laid out so that a trajectory's R workgroups share blockIdx % 8
    unsigned launch_id = 0;
    bool store_fw = true;
    double tol = 1e-16;
    double thr[KR_MAX_ORDER]{};
    // per-trajectory generators: trajectories are grouped by generator, every workgroup tile is homogeneous
    int G = 1;
    std::vector<int> perm;        // user trajectory k -> internal slot in the [Kp][N] buffers
    bool identity_perm = true;
    size_t pack_stride = 0;
    int *wg_gen_d = nullptr;
    double *nrm_d = nullptr;
    unsigned *present_d = nullptr;
    // device buffers
    double2 *Hpack_fw = nullptr, *Hpack_bw = nullptr;
    // the operator copies every sweep reads and nothing ever writes after kr_create (kr_pack_check): 0 forward pack, 1 adjoint pack,
    // 2 dense copies of the matrix-core sweep, 3 / 4 column-major copies (forward / adjoint) of the generic kernels
    unsigned long long pack_sum[5] = {0ull, 0ull, 0ull, 0ull, 0ull};   // checksums of the host copies
    size_t const_elems[5] = {0, 0, 0, 0, 0};
    unsigned long long *pack_words_d = nullptr;      // [2] running checksum + block counter of kr_pack_check
    size_t pack_elems = 0;
    double *dt_d = nullptr, *eps_a = nullptr, *eps_b = nullptr, *S_d = nullptr, *lambda_d = nullptr, *gint_d = nullptr;
    double2 *psi0_d = nullptr, *tgt_d = nullptr, *psiT_d = nullptr, *chiT_d = nullptr, *X_d = nullptr, *Phi_d = nullptr;
    double2 *tau_d = nullptr, *tausum_d = nullptr;
    double2 *U_d = nullptr;   // precomputed step propagators of the known-pulse sweeps (N <= 64), or nullptr
    // known-pulse sweeps in blocks of time (kr_device_tblock.h, NP = 64): tb_S steps per block (0: sequential apply sweep), block
    // propagators [G][NB][4096], states at the block boundaries [NB + 1][Kp][N], the trajectory ranges a workgroup carries
    int tb_S = 0, tb_NB = 0, tb_nitems = 0;
    bool tb_small = false;     // N <= 32: kr_tb_products_small + kr_sweep_apply<.., TB> (kr_device_prop.h) instead of the matrix-core kernels
    bool tb_tree = false;      // block propagators as a tree of pairwise products (tb_S a power of two), two ping-pong buffers of N_T / 2 matrices
    double2 *tb_P_d = nullptr, *tb_bound_d = nullptr, *tb_T_d[2] = {nullptr, nullptr};
    int *tb_items_d = nullptr;
    bool precomp = false;
    bool use_mfma = true;
    // update sweep on the matrix cores (kr_device_mfma.h, NP = 64): KT = 2 * tiles trajectories per 256-thread workgroup
    bool mfma_sweep = false;
    double2 *Hdense_fw = nullptr;
    size_t dense_stride = 0;
    unsigned long long *mail_d = nullptr, *xmail_d = nullptr;
    size_t mail_words = 0;
    bool herm = false;    // every control operator Hermitian: update kernel variant HERM (kr_device.h)
    bool econ = false;    // every generator Hermitian (drift and control operators): economized step polynomial in the update sweep
    double *econ_d = nullptr;   // [G][NT][KR_ECON_W] rows, rewritten from the guess pulses in front of every update sweep
    bool econ_gen = false;      // generic (multi-pass) kernels, Taylor method, every generator Hermitian: economized polynomial there too
    double *skw_d = nullptr;    // [G][1 + KR_LMAX_CAP] 2-norm bounds of the anti-Hermitian parts (pair mode with a generator that is not Hermitian), or nullptr
    int poll_delay = -1;  // exchange schedule of the update sweep (KrSweepArgs::poll_delay), set by calibrate_exchange
    unsigned long long *peer_xmail[8]{};
    // host timeline of the last update sweep (ms since the library was loaded):
    // a timeout between shards of one handle is reported with every shard's line (group_run)
    double tl_launch0 = 0, tl_launch1 = 0, tl_wait0 = 0, tl_wait1 = 0;
    bool peer_cached[8] = {false, false, false, false, false, false, false, false};   // peer mailbox r comes from the process-wide cache of IPC mappings: never closed by this handle
    int mail_kind = 0, xmail_kind = 0;   // 0 ordinary hipMalloc, 1 uncached, 2 fine-grained (special_alloc: returned to the pool, not freed)
    size_t xmail_bytes = 0;
    unsigned last_epoch_base = 0;   // tag base of the last update launch (the timeout report prints the tags it expected)
    unsigned long long xmail_gen = 0;   // how often this process has handed out the xmail block (special_alloc): seeds the tag base of the handle
    bool pulse_check = true;        // the workgroups' logs of the per-step totals compared after every update sweep of the N <= 64 / streamed kernels (kr_pulse_finish)
    double *dulog_d = nullptr;      // [P][NT][LMAX] (KrSweepArgs::dulog)
    int quarantined = 0;            // mailbox blocks that failed the self-test of kr_create and were set aside
    bool peers_connected = false;
    bool peers_in_process = false;   // peer mailboxes are plain pointers of sibling shards (device list), not IPC mappings
    unsigned *status_d = nullptr;
    unsigned long long *debug_d = nullptr;
    bool debug_on = false;
    unsigned long long spin_ticks = KR_SPIN_DEFAULT_TICKS;   // kr_config.spin_timeout_ms
    std::string plan_note;     // why a problem left the persistent kernels / what could not be verified
    std::string plan_text;     // kr_plan_description(): one line, written at the end of kr_create
    int wg_stride = 1;         // one-XCD placement of a small update grid (KrSweepArgs::wg_stride)
    int sets = 1;              // tile sets a persistent workgroup time-multiplexes within a step (K beyond one set per CU)
    unsigned epoch_base = 0;
    unsigned func_epoch = 0;   // tag of the cross-rank tau sums (J_T_sm on the device, world > 1)
    bool have_psi0 = false, have_tgt = false, have_X = false, have_final = false, have_params = false;
    // generic (multi-pass) path, kr_device_generic.h: known-pulse sweeps / update sweep run there when the persistent
    // kernels cannot (capacity) or must not (per-step RCCL / callback exchange)
    bool fast_ok = true;       // the persistent kernels have a plan for this problem
    bool gen_known = false, gen_update = false;
    int KTG = 0, RP = 0, JQg = 0, gen_tiles = 0;
    size_t gen_lds = 0, col_stride = 0;
    double2 *Hcol_fw = nullptr, *Hcol_bw = nullptr, *Heps_d = nullptr;
    int *tile_gen_d = nullptr;
    std::vector<double> nrm_host;   // [G][1 + L] bounds of the operators' 2-norms (kr_get_norm_bounds)
    double *nrm_gen_d = nullptr, *partial_d = nullptr, *du_local_d = nullptr, *du_total_d = nullptr;
    double *du_host = nullptr;   // pinned, callback exchange
    int exchange = KR_EXCHANGE_AUTO;   // resolved at kr_create
    int method = KR_PROP_AUTO;         // kr_prop_method; Chebyshev / Newton run on the generic kernels
    bool cheby_served = false;         // prop_method = CHEBY carried by the persistent kernels (make_plan); `method` is AUTO then
    double *ptab_d = nullptr, *erange_d = nullptr;
    double *poly_d = nullptr;          // non-linear controls: [L][poly_deg + 1] coefficients of f_l (generic kernels), or nullptr
    // ... on the persistent kernels (round 5): the known-pulse sweeps of every family are launched with the amplitudes f_l(eps) in the
    // place of the pulses (amp_d); the N <= 64 update kernels have instantiations for degree <= 3 (poly_fast: coefficients padded to
    // four per control in poly3_d, f_l(eps_guess) in amp_d and f_l'(eps_guess) in fprime_d, rewritten in front of every sweep)
    bool poly_fast = false;
    double *poly3_d = nullptr, *amp_d = nullptr, *fprime_d = nullptr;
    int poly_deg = 0;
    int ptab_stride = 0;
    double lam0 = 0.0;                 // max over generators of the norm bound of H0
    std::vector<double> lam_l;         // [L] the same for H_l
    std::vector<double> dt_host, ptab_host, eps_stage;
    kr_allreduce_fn cb = nullptr;
    void *cb_ctx = nullptr;
    ncclComm_t rccl_comm = nullptr;
    bool rccl_bad = false;             // a collective on rccl_comm failed: the communicator does not go back into the cache
    // timing: per kind (0 forward, 1 backward, 2 update) a pool of event pairs, one per launch since
    // the last reset; durations are read back (after a stream sync) by kr_get_timings
    bool timing = false;
    std::vector<hipEvent_t> ev[3];
    size_t ev_n[3]{0, 0, 0};
    // host staging
    std::vector<double> stage;
};

static int opt_int(const kr_handle *h, const char *name, int dflt) {
    if (h)
        for (const auto &o : h->options) if (o.first == name) return o.second;
    char env[96];
    snprintf(env, sizeof env, "KROTOV_HIP_%s", name);
    const char *s = getenv(env);
    return (s && *s) ? atoi(s) : dflt;
}

static double now_ms() {
    static const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
static int fail(kr_handle *h, int code, const char *fmt, ...) {
    char buf[3072];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}

// ---- nothing throws across the C boundary --------------------------------------------------------
// Every extern "C" entry point and every library-owned thread runs its body through kr_guard: std::bad_alloc from the staging
// vectors / new kr_handle, std::system_error from std::thread, anything else -> KR_ERR_INTERNAL + message, never an exception
// (or std::terminate) inside a ccall (/root/reference/src/optimize.jl:218-226 only knows Julia exceptions).
static int fail_nothrow(kr_handle *h, int code, const char *where, const char *what) noexcept {
    try { return fail(h, code, "%s: C++ exception caught at the library boundary: %s", where, what); } catch (...) { return code; }
}
template <class F>
static int kr_guard(kr_handle *h, const char *where, F &&f) noexcept {
    try { return f(); }
    catch (const std::bad_alloc &) { return fail_nothrow(h, KR_ERR_INTERNAL, where, "out of host memory (std::bad_alloc)"); }
    catch (const std::exception &e) { return fail_nothrow(h, KR_ERR_INTERNAL, where, e.what()); }
    catch (...) { return fail_nothrow(h, KR_ERR_INTERNAL, where, "unknown exception"); }
}

hipError_t kr_raise_dynamic_lds(const void *func, int bytes) {
    static std::mutex mtx;
    static std::vector<std::tuple<const void *, int, int>> done;   // (kernel, device, bytes)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> g(mtx);
    for (auto &d : done) if (std::get<0>(d) == func && std::get<1>(d) == dev && std::get<2>(d) >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.emplace_back(func, dev, bytes);
    return e;
}

#define HIPCHK(h, call)                                                                     \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail((h), KR_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ---- small host linear algebra ---------------------------------------------------------------
// Spectral norms of the operators: rigorous UPPER bounds (kr_norm.hip: ||A||_2 <= ||(A^H A)^p||_F^(1/(2p)) by repeated squaring on
// the device, at most 0.3 % above the norm).  Until round 4 a host power iteration ESTIMATED them from below (+0.1 % when its
// convergence test fired -- it can fire on a plateau --, +5 % at its iteration cap); the Taylor remainder forgives an
// underestimate (a factor (1 + d)^m), the economized step polynomial does not: it is accurate on [-theta_c, theta_c] only.
hipError_t kr_norm2_upper_bounds(int n, int count, const zc *const *mats, double *bounds);

// Pack generator m (column-major N x N) into the thread-major slice layout of the kernels:
// pack[(jj)*RT + i*JQ + jq] = M[i][jq*JS + jj]  (zero outside N).  adj: use M^dagger.
static void pack_generator(int N, int NP, int JQ, const zc *M, bool adj, double2 *out) {
    const int JS = NP / JQ, RT = NP * JQ;
    for (int jj = 0; jj < JS; ++jj)
        for (int i = 0; i < NP; ++i)
            for (int jq = 0; jq < JQ; ++jq) {
                const int j = jq * JS + jj;
                zc v = 0;
                if (i < N && j < N) v = adj ? std::conj(M[j + (size_t)i * N]) : M[i + (size_t)j * N];
                out[(size_t)jj * RT + i * JQ + jq] = make_double2(v.real(), v.imag());
            }
}

// Row-split pack: out[((jj)*R + rb)*T + il*JQ + jq] = M[rb*NR + il][column of (jq, jj)]  (one matrix; zero outside N).
// Column map: contiguous segments jq*JS + jj (single mode, padded LDS vectors) or strided jj*JQ + jq (pair mode, plain vectors)
static void pack_generator_rowsplit(int N, int NP, int NR, const zc *M, bool adj, double2 *out, bool strided) {
    const int T = 512;
    const int R = NP / NR, JQ = T / NR, JS = NP / JQ;
    for (int jj = 0; jj < JS; ++jj)
        for (int rb = 0; rb < R; ++rb)
            for (int il = 0; il < NR; ++il)
                for (int jq = 0; jq < JQ; ++jq) {
                    const int i = rb * NR + il, j = strided ? (jj * JQ + jq) : (jq * JS + jj);
                    zc v = 0;
                    if (i < N && j < N) v = adj ? std::conj(M[j + (size_t)i * N]) : M[i + (size_t)j * N];
                    out[((size_t)jj * R + rb) * T + il * JQ + jq] = make_double2(v.real(), v.imag());
                }
}

// Schedule / diagnostics options (INTEGRATION.md lists them).  A caller behind a C ABI sets them in kr_config.options
// ("NAME=value;NAME=value", names without the KROTOV_HIP_ prefix; kept in the handle); the environment variable KROTOV_HIP_<NAME>
// is the fallback for the lab.  kr_create rejects a name that is not in this list (a typo would otherwise be silently ignored).
static const char *const kr_option_names[] = {
    "APPLY_KT", "CALIBRATE", "CALIBRATE_CACHE", "CHEBY_RECURRENCE", "ECON", "ECON_PAIR", "EXPM_MFMA", "EXPM_PS", "GATHER_DELAY", "GATHER_DELAY_LOCAL", "GENERIC",
    "GENERIC_FUSED", "HERM", "KT", "MAIL_MEM", "MAX_WG", "MFMA_APPLY", "MFMA_APPLY_KT", "MFMA_SWEEP", "MAIL_SELFTEST", "MAIL_FAULT", "PACK_CHECK", "PACK_FAULT", "POLL_DELAY", "PRECOMP", "PULSE_CHECK", "PULSE_FAULT", "RCCL_CACHE", "ROWSPLIT",
    "RS512", "RS_BATCH", "RS_FAULT", "RS_LMAX1", "RS_TURNS", "RS_PAIR", "RS_POLL_DELAY", "RS_R1", "RS_SETS", "SETS", "SHARD_SKEW_MS", "STAMPS", "TBLOCK", "TBLOCK_MIN_NT", "TBLOCK_TREE", "UC_POOL", "XCD_GROUP", "XCD_LOCAL", "XLOC_HERM", "XPOLL_DELAY", "ZERO_KERNEL", "ZERO_SYNC"};
static bool parse_options(const char *text, std::vector<std::pair<std::string, int>> &out, std::string &bad) {
    out.clear();
    if (!text) return true;
    std::string t(text);
    size_t i = 0;
    while (i < t.size()) {
        size_t j = t.find_first_of(";, \t\n", i);
        if (j == std::string::npos) j = t.size();
        std::string tok = t.substr(i, j - i);
        i = j + 1;
        if (tok.empty()) continue;
        const size_t eq = tok.find('=');
        std::string name = tok.substr(0, eq);
        if (name.rfind("KROTOV_HIP_", 0) == 0) name = name.substr(11);
        bool known = false;
        for (const char *k : kr_option_names) known = known || name == k;
        char *end = nullptr;
        const std::string val = (eq == std::string::npos) ? std::string() : tok.substr(eq + 1);
        const long v = val.empty() ? 0 : strtol(val.c_str(), &end, 10);
        if (!known || val.empty() || (end && *end)) { bad = tok; return false; }
        out.emplace_back(name, (int)v);
    }
    return true;
}
static int opt_int(const kr_handle *h, const char *name, int dflt);

// ---- small device kernels for the standard functionals ---------------------
// tau_k = <tgt_k|psi_k(T)>  (taus!, optimize.jl:381); one wave per trajectory.
// Rows of the economized step polynomial (kr_device.h: KrEcon), one wave per (generator, interval): theta_c = margin x the
// bound of ||H(eps_guess_n) dt_n||; Chebyshev coefficients (2 - delta_k0) J_k(theta_c) by the ascending series (theta_c <= 1),
// smallest degree m whose dropped tail is below tol, then the same polynomial in powers of z = -i H dt:
//   a_j = theta_c^-j  sum_p (-1)^p (2 - delta) J_{j+2p}(theta_c) t_{j+2p, j},   T_k(x) = sum_j t_kj x^j.
// Row: [0] a_0, [1] a_1, [j] a_j / a_{j-1}, [30] theta_c, [31] m; m = 0: no row (the step takes the Taylor plan).
// One wave per row: lane k evaluates J_k, lane j the coefficient a_j (T_k's coefficients by the three-term recurrence, lane j holding
// t_kj and taking t_k,j-1 from its neighbour), ~10 us for the N_T rows of a sweep.
#define KR_ECON_KMAX 31
//
// NEARLY Hermitian generators (round 3; `skw` != nullptr: bounds of the 2-norms of the anti-Hermitian parts (M - M^dagger) / 2 of the
// drift and the control operators, laid out as `nrm`): a Liouvillian with weak dissipation.  The numerical range of x = H dt / theta_c
// lies in the rectangle [-1, 1] x i [-b, b], b = beta / theta_c, beta = the bound of the anti-Hermitian part of H(eps) dt for every
// pulse value the row is used with (the kernels use a row only while theta <= theta_c, which bounds sum |eps_l| ||H_l|| and with it
// beta <= s_0 + max_l (s_l / ||H_l||) (margin sum |eps_g,l| ||H_l|| + (margin - 1) ||H_0||)).  On the ellipse with foci +-1 through
// the rectangle's corner, |T_k| <= (rho^k + rho^-k) / 2, and ||p(A) - f(A)|| <= (1 + sqrt 2) max over the numerical range of |p - f|
// (Crouzeix-Palencia), so the degree is the smallest m with (1 + sqrt 2) sum_{k>m} |c_k| (rho^k + rho^-k) / 2 <= tol.  C5's
// Liouvillian (theta 0.09, beta 0.006, rho 1.3): degree 8 against Taylor's 9.  A row that saves nothing is not written.
// `nstride`: doubles per generator in nrm / skw (1 + KR_LMAX_CAP for the persistent kernels' table, 1 + L for the generic kernels');
// L: controls to sum over (the caller caps it at what the table holds).
__global__ __launch_bounds__(64) void kr_econ_rows_kernel(const double *eps, const double *dt, const double *nrm, const double *skw, int G, int NT, int L,
                                                           int nstride, double tol, double margin, double *rows) {
    __shared__ double c[64];
    const int idx = blockIdx.x, lane = threadIdx.x;
    if (idx >= G * NT) return;
    const int g = idx / NT, n = idx % NT;
    double *row = rows + (size_t)idx * KR_ECON_W;
    double th = nrm[(size_t)g * nstride];
    for (int l = 0; l < L; ++l) th += fabs(eps[(size_t)l * NT + n]) * nrm[(size_t)g * nstride + 1 + l];
    double beta = 0.0;
    if (skw) {
        const double n0 = nrm[(size_t)g * nstride];
        double ratio = 0.0, ctl = 0.0;
        bool unbounded = false;
        for (int l = 0; l < L; ++l) {
            const double nl = nrm[(size_t)g * nstride + 1 + l], sl = skw[(size_t)g * nstride + 1 + l];
            ctl += fabs(eps[(size_t)l * NT + n]) * nl;
            if (sl > 0.0) { if (nl > 0.0) ratio = fmax(ratio, sl / nl); else unbounded = true; }
        }
        beta = skw[(size_t)g * nstride] + ratio * (margin * ctl + (margin - 1.0) * n0);
        beta *= fabs(dt[n]);
        if (unbounded) th = 0.0;   // (cannot happen: the 2-norm bound of an operator is not below that of its anti-Hermitian part)
    }
    th *= fabs(dt[n]) * margin;
    if (!(th > 0.0) || th > KR_THETA_MAX) {   // no row: the step takes the Taylor plan
        if (lane < KR_ECON_W) row[lane] = 0.0;
        return;
    }
    // weight of |c_k| in the tail: 1 on the segment; (1 + sqrt 2) (rho^k + rho^-k) / 2 on the ellipse (see above)
    double wgt = 1.0;
    if (beta > 0.0) {
        const double b = beta / th;
        const double ae = 0.5 * (sqrt(4.0 + b * b) + b);           // semi-major axis of the ellipse with foci +-1 through (1, b)
        const double rho = ae + sqrt(fmax(ae * ae - 1.0, 0.0));
        if (rho * th > 2.0) {   // far from Hermitian: the weighted coefficients beyond KR_ECON_KMAX would no longer be negligible -- no row
            if (lane < KR_ECON_W) row[lane] = 0.0;
            return;
        }
        double rk = 1.0;
        for (int k = 0; k < lane && k < KR_ECON_KMAX + 1; ++k) rk *= rho;
        wgt = 2.414213562373096 * 0.5 * (rk + 1.0 / rk);
    }
    {   // c_k = (2 - delta_k0) J_k(th): (th/2)^k / k!  x  sum_i (-th^2/4)^i / (i! (k+1)...(k+i))
        const double half = 0.5 * th, q = -half * half;
        double pref = 1.0;
        for (int k = 1; k <= lane && k <= KR_ECON_KMAX; ++k) pref *= half / (double)k;
        double sum = 1.0, term = 1.0;
        for (int i = 1; i <= 14; ++i) { term *= q / ((double)i * (double)(lane + i)); sum += term; }
        c[lane] = (lane <= KR_ECON_KMAX) ? (lane ? 2.0 : 1.0) * pref * sum : 0.0;
    }
    __syncthreads();
    double tail = 0.0;
    if (beta > 0.0) {   // (the weights differ per k: weigh first, in place of c -- the coefficients themselves are read from `cw` below)
        __shared__ double cw[64];
        cw[lane] = fabs(c[lane]) * wgt;
        __syncthreads();
        for (int k = KR_ECON_KMAX; k >= lane && k >= 0; --k) tail += cw[k];
    } else
    for (int k = KR_ECON_KMAX; k >= lane && k >= 0; --k) tail += fabs(c[k]);   // sum_{k >= lane} |c_k|
    const unsigned long long over = __ballot(lane >= 2 && lane <= KR_ECON_KMAX && tail > tol);
    const int m = over ? (63 - __builtin_clzll(over)) : 1;   // smallest m >= 1 with sum_{k > m} |c_k| <= tol
    if (m > 29) {
        if (lane < KR_ECON_W) row[lane] = 0.0;
        return;
    }
    double acc = 0.0, tp = 0.0, tc = (lane == 0) ? 1.0 : 0.0;   // lane j: a_j (before theta^-j), t_{k-1,j}, t_{k,j}
    for (int k = 0; k <= m; ++k) {
        if (lane <= k && (((k - lane) & 1) == 0)) acc += ((((k - lane) >> 1) & 1) ? -1.0 : 1.0) * c[k] * tc;
        const double up = __shfl_up(tc, 1, 64);
        double tn = ((lane > 0) ? 2.0 * up : 0.0) - tp;   // T_{k+1} = 2 x T_k - T_{k-1}
        if (k == 0) tn = (lane == 1) ? 1.0 : 0.0;         // T_1 = x
        tp = tc; tc = tn;
    }
    double ti = 1.0;
    for (int j = 0; j < lane && j < 30; ++j) ti /= th;
    const double aj = acc * ti, ajm1 = __shfl_up(aj, 1, 64);
    if (lane < KR_ECON_W) {
        double v = 0.0;
        if (lane <= 1) v = aj;
        else if (lane <= m) v = aj / ajm1;
        else if (lane == 30) v = th;
        else if (lane == 31) v = (double)m;
        row[lane] = v;
    }
}

// Integrity of the generator packs (round 4).  The packs are uploaded once at kr_create and read by every sweep of the handle's
// life; a corrupted pack does not crash anything -- it propagates with a slightly wrong generator, smoothly and silently (the one
// wrong result the driver ever recorded, GPUTEST_r03, had exactly that signature: a backward sweep -- the only reader of the adjoint
// pack -- off by ~1e-3 per step, never reproduced).  So every sweep is preceded by one small launch that folds the pack it is about
// to read into a position-sensitive 64-bit checksum (XOR of rotated words: order-independent, so plain atomics do) and compares
// it with the checksum of the host copy taken at kr_create; a difference becomes status code 7 (KR_ERR_INTERNAL).
__host__ __device__ static inline unsigned long long kr_pack_word(unsigned long long a, unsigned long long b, size_t i) {
    const int r1 = (int)(i & 63), r2 = (int)((i * 7 + 13) & 63);
    return ((a << r1) | (r1 ? a >> (64 - r1) : 0ull)) ^ ((b << r2) | (r2 ? b >> (64 - r2) : 0ull)) ^ ((unsigned long long)i * 0x9E3779B97F4A7C15ull);
}
static unsigned long long pack_checksum_host(const double2 *p, size_t n) {
    unsigned long long cs = 0ull;
    for (size_t i = 0; i < n; ++i) {
        unsigned long long a, b;
        memcpy(&a, &p[i].x, 8); memcpy(&b, &p[i].y, 8);
        cs ^= kr_pack_word(a, b, i);
    }
    return cs;
}
// words: [0] running checksum, [1] blocks that have added theirs; both zeroed by the last block for the next use
__global__ __launch_bounds__(256) void kr_pack_check(const double2 *p, size_t n, unsigned long long expect, unsigned long long *words, int which, unsigned *status) {
    __shared__ unsigned long long sw[4];
    unsigned long long cs = 0ull;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double2 v = p[i];
        cs ^= kr_pack_word((unsigned long long)__double_as_longlong(v.x), (unsigned long long)__double_as_longlong(v.y), i);
    }
    for (int o = 1; o < 64; o <<= 1) cs ^= __shfl_xor(cs, o, 64);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = cs;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicXor(words, sw[0] ^ sw[1] ^ sw[2] ^ sw[3]);
        __threadfence();
        if (atomicAdd(words + 1, 1ull) == (unsigned long long)gridDim.x - 1ull) {   // last block: every contribution is in
            const unsigned long long got = atomicXor(words, 0ull);
            if (got != expect && atomicCAS(status, 0u, 1u) == 0u) { status[1] = 7u; status[2] = (unsigned)which; }
            atomicExch(words, 0ull);
            atomicExch(words + 1, 0ull);
        }
    }
}

// Zeroing in front of a launch (mailboxes with one-bit validity, the running integrals) as an ordinary kernel on the handle's stream:
// KROTOV_HIP_ZERO_KERNEL=1 (default: hipMemsetAsync), KROTOV_HIP_ZERO_SYNC=1 waits for it on the host -- diagnostics that ruled the
// zeroing out as the source of the partial sums lost in round 4 ("device lists").
__global__ __launch_bounds__(256) void kr_zero_words(unsigned long long *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0ull;
}

// After a single-mode row-split sweep (kr_rowsplit_sweep): the R workgroups of every trajectory posted a checksum of the state
// vector they ended with; all R must agree (every element of every copy went through the all-gather).  A difference -- a stale or
// torn element accepted by one member of the group -- becomes status code 6 instead of a silently wrong sweep.
__global__ void kr_rowsplit_check(const unsigned long long *words, int tiles, int R, unsigned *status) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tiles) return;
    const unsigned long long ref = words[(size_t)t * R];
    bool bad = (ref == 0ull);
    for (int r = 1; r < R; ++r) bad = bad || (words[(size_t)t * R + r] != ref);
    if (bad && atomicCAS(status, 0u, 1u) == 0u) { status[1] = 6u; status[2] = (unsigned)t; }
}

// Behind an update sweep of the register-resident / matrix-core / streamed kernels (N <= 64 and the streamed fallback).  Every
// workgroup formed eps_new for itself from the totals du_l IT gathered and logged them (KrSweepArgs::dulog, [P][NT][LMAX]); the
// kernels no longer write eps_new.  This kernel forms it from workgroup 0's log with the sweep's own expression (optimize.jl:354:
// eps + (S / lambda) du, one rounded product, one rounded sum) and compares all P logs bit for bit: a workgroup that summed a
// stale or lost partial sum once has propagated with another pulse than its neighbours (status code 8, with workgroup, control and
// step).  optimize.jl:351-358 has ONE pulse value per step.
// Non-linear controls on the persistent kernels: amp[l][n] = f_l(eps[l][n]) and (optionally) fprime[l][n] = f_l'(eps[l][n]) for
// polynomial f_l (coefficients poly[l][0..deg], as kr_gen_f / kr_gen_fprime of the generic kernels evaluate them).
__global__ __launch_bounds__(256) void kr_poly_eval(const double *eps, const double *poly, int deg, int L, int NT, double *amp, double *fprime) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)L * NT) return;
    const double *cf = poly + (size_t)(i / NT) * (deg + 1);
    const double x = eps[i];
    double v = cf[deg];
    for (int p = deg - 1; p >= 0; --p) v = fma(v, x, cf[p]);
    amp[i] = v;
    if (fprime) {
        double d = deg * cf[deg];
        for (int p = deg - 1; p >= 1; --p) d = fma(d, x, p * cf[p]);
        fprime[i] = d;
    }
}

__global__ __launch_bounds__(256) void kr_pulse_finish(const double *eps_guess, const double *alpha, const double *fprime, const double *dulog, double *eps_new, int P, int L,
                                                       int LMAX, int NT, int check, unsigned *status, int ref_wg) {
    // blockIdx.y: a slice of the workgroups' logs (a first version walked all P logs from every thread: 256 dependent-free but
    // serial loads per thread on eight workgroups -- 112 us at C3, 2 % of the iteration)
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)L * NT) return;
    const int l = (int)(i / NT), n = (int)(i % NT);
    // (ref_wg: the workgroup whose log is the reference -- 0, or 1 where workgroup 0 spends its one store on the caller's eps_new: pair mode)
    const double du = dulog[((size_t)ref_wg * NT + n) * LMAX + l];
    if (blockIdx.y == 0 && eps_new)   // (row-split kernels write eps_new themselves and pass nullptr: comparison only)
        eps_new[i] = __dadd_rn(eps_guess[i], __dmul_rn(alpha[i], fprime ? __dmul_rn(fprime[i], du) : du));   // (fprime: mu_l = f_l'(eps_guess) H_l, non-linear controls)
    if (!check) return;
    const long long ref = __double_as_longlong(du);
    const int per = (P + (int)gridDim.y - 1) / (int)gridDim.y, w0 = max(ref_wg + 1, (int)blockIdx.y * per), w1 = min(P, ((int)blockIdx.y + 1) * per);
    int who = 0;
    for (int w = w0; w < w1; ++w) if (__double_as_longlong(dulog[((size_t)w * NT + n) * LMAX + l]) != ref && who == 0) who = w;
    if (who && atomicCAS(status, 0u, 1u) == 0u) { status[1] = 8u; status[2] = (unsigned)who; status[3] = (unsigned)i; }
}

// Self-test of a freshly allocated mailbox (kr_create; round 5): the exchange pattern of the sweeps on the mailbox's own words -- every
// one of P workgroups stores one word per slot it owns into every replica (write-through stores), then polls ALL slots of the replica of
// the XCD it runs on (loads that bypass the vector L1) until every word carries this round's value; two rounds with different values.
// What it catches ("blocks that follow a freed uncached block"): an address range on which single stores never reach memory,
// or reach it and stay invisible to some XCDs.  Bounded: a workgroup that does not see every word within `ticks` sets status code 9.
// Layout under test: `replicas` copies `rstride` words apart, slot q at q * sgran (+ w for the `width` words of a slot).
__global__ __launch_bounds__(64) void kr_mail_selftest(unsigned long long *mail, int P, int replicas, size_t rstride, int sgran, int width,
                                                        unsigned long long ticks, int fault, unsigned *status) {
    const int wg = blockIdx.x, lane = threadIdx.x;
    unsigned xcc;
    asm("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int rep = (int)(xcc & 7u) % replicas;
    for (unsigned round = 1; round <= 2; ++round) {
        const unsigned long long val = 0x5a17000000000000ull | ((unsigned long long)round << 40);
        for (int e = lane; e < replicas * width; e += 64) {
            const int r = e / width, w = e % width;
            if (fault && round == 2 && wg == P / 2 && r == replicas - 1) continue;   // (tests: one word never published)
            __hip_atomic_store(mail + (size_t)r * rstride + (size_t)wg * sgran + w, val | (unsigned long long)(wg * 8 + w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            bool ok = true;
            for (int e = lane; e < P * width; e += 64) {
                const int q = e / width, w = e % width;
                const unsigned long long got = __hip_atomic_load(mail + (size_t)rep * rstride + (size_t)q * sgran + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // (a faster workgroup may already have published its word of round 2)
                ok = ok && ((got & ~(3ull << 40)) == (0x5a17000000000000ull | (unsigned long long)(q * 8 + w))) && (((got >> 40) & 3ull) >= round);
            }
            if (__all(ok)) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > ticks) {
                if (lane == 0 && atomicCAS(status, 0u, 1u) == 0u) { status[1] = 9u; status[2] = (unsigned)wg | (round << 16) | ((unsigned)rep << 24); }
                return;
            }
            __builtin_amdgcn_s_sleep(4);
        }
    }
}

__global__ void kr_tau_kernel(const double2 *tgt, const double2 *psi, double2 *tau, int N, int K) {
    const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= K) return;
    double sr = 0, si = 0;
    for (int i = lane; i < N; i += 64) {
        const double2 t = tgt[(size_t)k * N + i], p = psi[(size_t)k * N + i];
        sr += t.x * p.x + t.y * p.y;  // conj(t) * p
        si += t.x * p.y - t.y * p.x;
    }
    for (int o = 1; o < 64; o <<= 1) { sr += __shfl_xor(sr, o, 64); si += __shfl_xor(si, o, 64); }
    if (lane == 0) tau[k] = make_double2(sr, si);
}
// Fixed-order sum of the K local taus (256 threads: strided partial sums, then an LDS tree); with world > 1 the local sum
// goes into every rank's xmail (tau slots behind the update-sweep slots: [2][8][4] granules {tag32 | half of a double},
// system-scope stores over xGMI / IPC) and kr_taugather_kernel adds the ranks' sums in rank order -- J_T_sm couples ALL
// trajectories (chi_k = (sum_j tau_j) / K^2 tgt_k, /root/reference/src/optimize.jl:297-302 with make_chi for J_T_sm).
#define KR_XMAIL_TAU_OFF (2 * 8 * 2 * KR_LMAX_CAP)
struct KrPeers { unsigned long long *p[8]; };
__global__ __launch_bounds__(256) void kr_tausum_kernel(const double2 *tau, double2 *out, int K, int world, int rank, unsigned tag, KrPeers peers) {
    __shared__ double sr[256], si[256];
    double ar = 0, ai = 0;
    for (int k = threadIdx.x; k < K; k += 256) { ar += tau[k].x; ai += tau[k].y; }
    sr[threadIdx.x] = ar; si[threadIdx.x] = ai;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { sr[threadIdx.x] += sr[threadIdx.x + o]; si[threadIdx.x] += si[threadIdx.x + o]; }
        __syncthreads();
    }
    if (world <= 1) { if (threadIdx.x == 0) out[0] = make_double2(sr[0], si[0]); return; }
    if ((int)threadIdx.x < 4 * world) {
        const int r = threadIdx.x >> 2, g = threadIdx.x & 3, par = (int)(tag & 1u);
        const unsigned long long bits = (unsigned long long)__double_as_longlong(g < 2 ? sr[0] : si[0]);
        const unsigned half = (g & 1) ? (unsigned)(bits >> 32) : (unsigned)bits;
        __hip_atomic_store(peers.p[r] + KR_XMAIL_TAU_OFF + ((size_t)par * 8 + rank) * 4 + g, ((unsigned long long)tag << 32) | half,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void kr_taugather_kernel(unsigned long long *xmail, double2 *out, int world, unsigned tag, unsigned *status, unsigned long long spin_ticks) {
    const int lane = threadIdx.x, par = (int)(tag & 1u);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    double re = 0, im = 0;
    for (;;) {
        bool ok = true;
        if (lane < world) {
            unsigned long long g[4];
            for (int q = 0; q < 4; ++q) g[q] = __hip_atomic_load(xmail + KR_XMAIL_TAU_OFF + ((size_t)par * 8 + lane) * 4 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            for (int q = 0; q < 4; ++q) ok = ok && ((unsigned)(g[q] >> 32) == tag);
            re = __longlong_as_double((long long)((g[1] << 32) | (g[0] & 0xffffffffull)));
            im = __longlong_as_double((long long)((g[3] << 32) | (g[2] & 0xffffffffull)));
        }
        if (__all(ok)) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > spin_ticks) {
            if (lane == 0) { status[1] = 4u; status[2] = tag; __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            break;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    double tr = 0, ti = 0;
    for (int r = 0; r < world; ++r) { tr += __shfl(re, r, 64); ti += __shfl(im, r, 64); }   // rank order: identical on every rank
    if (lane == 0) out[0] = make_double2(tr, ti);
}
// chi_k(T) for J_T_sm / J_T_ss / J_T_re [QuantumControl.Functionals make_chi], weights 1.
__global__ void kr_chi_kernel(int functional, const double2 *tgt, const double2 *tau, const double2 *tausum,
                              double2 *chi, int N, int K, int Kp, double Kglobal) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)Kp * N) return;
    const int k = (int)(idx / N);
    double2 c = make_double2(0.0, 0.0);
    if (k < K) {
        double fr, fi;
        if (functional == KR_J_T_SM) { fr = tausum[0].x / (Kglobal * Kglobal); fi = tausum[0].y / (Kglobal * Kglobal); }
        else if (functional == KR_J_T_SS) { fr = tau[k].x / Kglobal; fi = tau[k].y / Kglobal; }
        else { fr = 1.0 / (2.0 * Kglobal); fi = 0.0; }
        const double2 t = tgt[idx];
        c = make_double2(fr * t.x - fi * t.y, fr * t.y + fi * t.x);
    }
    chi[idx] = c;
}

// ---- RCCL, loaded at run time ---------------------------------------------------------------------
// The library has no link-time dependency on librccl: KR_EXCHANGE_RCCL resolves the five entry points it needs with
// dlopen/dlsym, preferring a librccl that is already mapped into the process (PyTorch-ROCm brings its own).
struct KrRccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};
static KrRccl g_rccl;
static KrRccl *kr_rccl() {
    KrRccl &r = g_rccl;
    static std::once_flag once;
    std::call_once(once, [&r] {
        // One RCCL per process: a second copy (PyTorch-ROCm bundles its own librccl.so, which its libtorch_hip.so requests
        // under that name) ends in a double free at exit.  KROTOV_HIP_RCCL_LIB names the copy to use (the Python binding
        // points it at PyTorch's file); otherwise whatever is mapped already, then the system library.
        const char *envp = getenv("KROTOV_HIP_RCCL_LIB");
        const char *names[] = {(envp && *envp) ? envp : "librccl.so", "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char *n : names) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!r.lib) { const char *de = dlerror(); r.why = std::string("librccl could not be loaded: ") + (de ? de : "not found"); return; }
#define KR_SYM(field, name) r.field = (decltype(r.field))dlsym(r.lib, name); if (!r.field && r.why.empty()) r.why = std::string("librccl lacks the symbol ") + name;
        KR_SYM(GetUniqueId, "ncclGetUniqueId") KR_SYM(CommInitRank, "ncclCommInitRank") KR_SYM(CommInitAll, "ncclCommInitAll")
        KR_SYM(CommDestroy, "ncclCommDestroy") KR_SYM(AllReduce, "ncclAllReduce") KR_SYM(GroupStart, "ncclGroupStart")
        KR_SYM(GroupEnd, "ncclGroupEnd") KR_SYM(GetErrorString, "ncclGetErrorString")
#undef KR_SYM
    });
    return r.why.empty() ? &r : nullptr;
}
static const char *kr_rccl_why() { (void)kr_rccl(); return g_rccl.why.empty() ? "librccl unavailable" : g_rccl.why.c_str(); }

// ---- plan -------------------------------------------------------------------------------------
static int make_plan_fast(kr_handle *h, const int32_t *tg, std::vector<int> &wg_gen, int maxwg_cfg) {
    const kr_config &c = h->cfg;
    if (c.N > 1024) return fail(h, KR_ERR_UNSUPPORTED, "N = %d > 1024 is not supported by the persistent kernels", c.N);
    if (c.N > 256 && (c.L > ((c.N <= 512) ? 4 : 2) || opt_int(h, "RS512", 1) == 0)) return fail(h, KR_ERR_UNSUPPORTED, "N = %d > 256 with L = %d: no persistent kernel", c.N, c.L);
    if (c.L > KR_LMAX_CAP) return fail(h, KR_ERR_UNSUPPORTED, "L = %d > %d controls not supported by this build", c.L, KR_LMAX_CAP);
    int NP = 8;
    while (NP < c.N) NP <<= 1;
    h->NP = NP;
    h->LMAXv = (c.L <= 2) ? 2 : 4;
    if (NP > 64) {
        // streamed-generator kernels (kr_device_stream.h): one team of 4*NP threads per workgroup
        h->threads = 4 * NP;
        h->JQ = 4;
    } else {
        // mirrors KrGeom (kr_device.h): NP = 64 uses 512-thread teams (JQ = 8)
        h->threads = (NP >= 64) ? 512 : 256;
        h->JQ = (h->threads == 512) ? 8 : 4;
    }
    h->JS = NP / h->JQ;
    h->RT = NP * h->JQ;
    h->SG = h->threads / h->RT;
    switch (NP) {
        case 8: h->launch = kr_launch_np8; break;
        case 16: h->launch = kr_launch_np16; break;
        case 32: h->launch = kr_launch_np32; break;
        case 64: h->launch = kr_launch_np64; break;
        case 128: h->launch = kr_launch_np128; break;
        default: h->launch = kr_launch_np256; break;
    }
    hipDeviceProp_t prop;
    HIPCHK(h, hipGetDeviceProperties(&prop, h->device));
    int cus = prop.multiProcessorCount;
    // one persistent workgroup per CU; a handle that shares its GPU (several ranks or shards per device) gets a share:
    // kr_config.max_workgroups (a device list with repeated ordinals sets it by itself), or the environment for tests
    int maxwg = opt_int(h, "MAX_WG", maxwg_cfg > 0 ? std::min(maxwg_cfg, cus) : cus);
    if (maxwg < 1) maxwg = 1;
    if (NP > 64 && opt_int(h, "ROWSPLIT", 1) != 0) {
        // R workgroups per trajectory, NR = NP / R rows each (8 <= NR <= 64: the H(eps) block must fit the registers)
        int R = (NP == 1024) ? 128 : (NP == 512) ? ((c.L > 2) ? 64 : 32) : NP / 64;   // (three or four controls at NP = 512: 8 rows per workgroup only)   // (NP = 512 / 1024, round 3: at most 16 / 8 rows per workgroup -- the block of H(eps) stays in 64 registers)
        // N = 256, one control, shared generator, more pairs than groups of 8 workgroups: the groups carry up to 4 pairs each in turn
        // (round 3: kr_rowsplit_pair_sweep<.., MULTI>; before: the streamed-generator kernels, 8 x the time per trajectory)
        int rs_sets = 1;
        if (NP == 256 && c.L == 1 && h->G == 1 && opt_int(h, "RS_LMAX1", 1) != 0 && opt_int(h, "RS_PAIR", 1) != 0 &&
            opt_int(h, "RS_SETS", 1) != 0 && maxwg >= 8 && (long long)c.K * R > maxwg) {
            const int groups = maxwg / 8, pairs = (c.K + 1) / 2;
            const int need = (pairs + groups - 1) / groups;
            if (need >= 2 && need <= 4) rs_sets = need;
        }
        // N <= 128 with more trajectories than half the CUs: R = 1, one workgroup per trajectory with the resident H(eps) block
        // (round 3; before: the streamed-generator kernels, (1 + L) matrices per ROUND -- 39 us per step against ~10)
        // (L <= 2: the four-control instantiation spills 2 KB per thread and loses to the streamed-generator kernels)
        if (NP == 128 && c.L <= 2 && (long long)c.K * R > maxwg && c.K <= maxwg && opt_int(h, "RS_R1", 1) != 0) R = 1;
        int rs_batch = 0;
        if (NP >= 512 && (long long)c.K * R > maxwg && R <= maxwg && opt_int(h, "RS_BATCH", 1) != 0) rs_batch = maxwg / R;
        // 128 < N <= 256 with two controls and more trajectories than 4 workgroups each allow (round 3, end): groups of 8 workgroups
        // (32 rows each: the H(eps) block in registers) with up to 8 trajectories per group in turn, known-pulse sweeps in batches --
        // instead of the streamed-generator kernels.  (One control: pair mode with sets.  More turns: streamed kernels as before.)
        if (NP == 256 && c.L == 2 && h->G == 1 && (long long)c.K * R > maxwg && maxwg >= 8 && (c.K + maxwg / 8 - 1) / (maxwg / 8) <= 8 /* (K = 512, 16 turns: 193 / 222 us per step against 175 / 240 streamed) */ &&
            opt_int(h, "RS_BATCH", 1) != 0 && opt_int(h, "RS_TURNS", 1) != 0) { R = 8; rs_batch = maxwg / 8; }
        if (((long long)c.K * R <= maxwg || rs_sets > 1 || rs_batch > 0) && (long long)c.NT * (KR_MAX_ORDER + 2) * 4 < (1ll << 24)) {
            while (rs_batch == 0 && NP / (2 * R) >= 8 && (long long)c.K * 2 * R <= maxwg) R *= 2;
            h->rs_batch = rs_batch;
            // the update sweep of such a handle: turns inside one persistent kernel (shared generator, one or two controls, at most
            // KR_RS_MAX_TURNS trajectories per group); otherwise the stepped generic update
            if (rs_batch > 0 && h->G == 1 && (c.L <= 2 || NP == 512) && opt_int(h, "RS_TURNS", 1) != 0) {   // (three or four controls: NP = 512 only)
                const int turns = (c.K + rs_batch - 1) / rs_batch;
                if (turns <= KR_RS_MAX_TURNS) h->rs_turns = turns;
            }
            h->rowsplit = true;
            if (c.L == 1 && opt_int(h, "RS_LMAX1", 1) != 0) h->LMAXv = 1;   // row-split kernels have an instantiation for ONE control (C5): fewer
                                                                                 // accumulators and chunk registers in the streaming passes
            // PAIR mode: when one trajectory per group would leave 64 rows per workgroup (the H(eps) block then fills the
            // registers, half of it in LDS, the control block is re-streamed every step and the group idles during every
            // all-gather), the same workgroups in groups of 2R carry TWO trajectories with 32 rows each: both blocks resident,
            // one trajectory's all-gather behind the other's row-dot.  One control, shared generator (C5's shape); an odd K is
            // padded with one zero trajectory (it contributes nothing to any sum).  KROTOV_HIP_RS_PAIR=0 keeps the single mode.
            if (NP / R == 64 && h->LMAXv == 1 && h->G == 1 && c.K >= 2 && ((long long)((c.K + 1) / 2) * (2 * R) <= maxwg || rs_sets > 1) &&
                opt_int(h, "RS_PAIR", 1) != 0) {
                h->rs_pair = true;
                R *= 2;
            }
            h->rs_sets = h->rs_pair ? rs_sets : 1;
            h->R = R;
            h->NR = NP / R;
            h->threads = 512;  // mirrors KrGeomR
            h->JQ = h->threads / h->NR;
            h->JS = NP / h->JQ;
            h->RT = h->threads;
            h->SG = 1;
            h->launch_rs = (NP == 128) ? kr_launch_rs128 : (NP == 512) ? kr_launch_rs512 : (NP == 1024) ? kr_launch_rs1024 : kr_launch_rs256;
            h->KT = h->rs_pair ? 2 : 1;
            h->KTILE = h->KT;
            h->Kp = ((c.K + h->KT - 1) / h->KT) * h->KT;   // pair mode: padded to an even count
            h->P = (h->Kp / h->KT) * R;
            if (h->rs_batch > 0) h->P = std::min(c.K, h->rs_batch) * R;   // (one batch: the grid of a known-pulse launch)
            if (h->rs_turns > 0) { h->Kp = h->rs_batch * h->rs_turns; h->P = h->rs_batch * R; }   // groups x turns, zero trajectories behind the last real one
            if (h->rs_sets > 1) {   // groups x sets pairs, zero trajectories behind the last real one
                const int groups = maxwg / R;
                h->Kp = 2 * groups * h->rs_sets;
                h->P = groups * R;
            }
            h->perm.assign(c.K, 0);
            wg_gen.assign(h->Kp / h->KT, 0);
            for (int k = 0; k < c.K; ++k) { h->perm[k] = k; wg_gen[k / h->KT] = tg ? tg[k] : 0; }
            h->identity_perm = true;
            return KR_OK;
        }
    }
    if (NP > 256) return fail(h, KR_ERR_UNSUPPORTED, "N = %d > 256 with %d trajectories: more than the row-split kernels carry (K x %d workgroups <= %d)", c.N, c.K, NP == 1024 ? 128 : 32, maxwg);
    int kt = opt_int(h, "KT", 0);
    if (kt != 0 && kt != 1 && kt != 2 && kt != 4) return fail(h, KR_ERR_INVALID, "KROTOV_HIP_KT must be 1, 2 or 4");
    // trajectories per generator (tiles never mix generators)
    std::vector<int> count(h->G, 0);
    for (int k = 0; k < c.K; ++k) count[tg ? tg[k] : 0]++;
    auto tiles_for = [&](int ktv) {
        int t = 0;
        for (int g = 0; g < h->G; ++g) t += (count[g] + h->SG * ktv - 1) / (h->SG * ktv);
        return t;
    };
    const int kt_forced = kt;
    if (kt == 0) {
        kt = 1;
        while (kt < 4 && tiles_for(kt) > maxwg) kt <<= 1;
    }
    // NP = 64: the update sweep runs on the matrix cores (v_mfma_f64_4x4x4_4b_f64: 2 trajectories fill a tile) whenever
    // the vector kernel would carry several trajectories per team or more than two control slices (both spill there).
    // KROTOV_HIP_MFMA_SWEEP = 0 never / 1 always / unset: that rule.
    if (NP == 64) {
        const int mf = opt_int(h, "MFMA_SWEEP", -1);
        const bool want = (mf == 1) || (mf == -1 && (kt >= 2 || h->LMAXv == 4 || tiles_for(kt) > maxwg) && kt_forced != 1);
        if (want && kt_forced != 1) {
            int ktm = (kt_forced >= 2) ? kt_forced : 2;
            while (ktm < 4 && kt_forced == 0 && tiles_for(ktm) > maxwg) ktm <<= 1;
            h->mfma_sweep = true;
            kt = ktm;
            // more trajectories than 4 per CU: the workgroup time-multiplexes several sets of two tiles within a step
            // (kr_mf_update<.., SETS>): K = 2048 / 4096 per GPU stay on the persistent kernel.  KROTOV_HIP_SETS fixes the count.
            if (kt == 4) {
                const int want_sets = opt_int(h, "SETS", 0);
                if (want_sets == 1 || want_sets == 2 || want_sets == 4) h->sets = want_sets;
                else while (h->sets < 4 && tiles_for(kt * h->sets) > maxwg) h->sets <<= 1;
            }
        }
    }
    h->KT = kt;
    h->KTILE = h->SG * kt * h->sets;
    h->P = tiles_for(kt * h->sets);
    if (h->P > maxwg)
        return fail(h, KR_ERR_UNSUPPORTED, "K = %d trajectories (%d generators) need %d co-resident workgroups (> %d CUs) at NP = %d", c.K, h->G, h->P, maxwg, NP);
    h->Kp = h->P * h->KTILE;
    // internal slots: generator groups one after the other, each padded to whole tiles
    h->perm.assign(c.K, 0);
    wg_gen.assign(h->P, 0);
    std::vector<int> tile_base(h->G, 0), seen(h->G, 0);
    int t = 0;
    for (int g = 0; g < h->G; ++g) {
        tile_base[g] = t;
        const int nt = (count[g] + h->KTILE - 1) / h->KTILE;
        for (int q = 0; q < nt; ++q) wg_gen[t + q] = g;
        t += nt;
    }
    h->identity_perm = true;
    for (int k = 0; k < c.K; ++k) {
        const int g = tg ? tg[k] : 0;
        h->perm[k] = tile_base[g] * h->KTILE + seen[g]++;
        if (h->perm[k] != k) h->identity_perm = false;
    }
    // Few workgroups (C2: 8): put them all on ONE XCD so that the per-step exchange stays in its L2 (KrSweepArgs::wg_stride).
    // Only when this handle has the GPU to itself (the other XCDs' blocks exit at once, but they must be dispatched) and for the
    // vector update kernel.  KROTOV_HIP_XCD_LOCAL=0 switches it off; = 2 forces it for a handle that shares its GPU (tests: several
    // ranks on the one GPU of the build box, the shape a shard of 32 trajectories has on every GPU of an 8-GPU run).
    h->wg_stride = 1;
    {
        const int xl = opt_int(h, "XCD_LOCAL", 1);
        // (an explicitly forced kernel variant -- KROTOV_HIP_HERM = 2, KROTOV_HIP_POLL_DELAY >= 0 -- keeps the ordinary launch: the
        //  placement kernel exists for the classic schedule of the general variant only)
        const bool forced = opt_int(h, "HERM", 1) == 2 || opt_int(h, "POLL_DELAY", -2) >= 0;
        if (NP <= 64 && !h->mfma_sweep && h->P >= 2 && h->P <= 32 && (maxwg == cus || xl == 2) && xl != 0 && !forced) h->wg_stride = 8;
    }
    return KR_OK;
}

// Generic plan (kr_device_generic.h): tiles of KTG trajectories, as many workgroups as it takes.
static int gen_geometry(kr_handle *h, int ktile_limit) {
    const int N = h->cfg.N;
    int rp = 16;
    while (rp < N && rp < 256) rp <<= 1;
    h->RP = rp;
    h->JQg = 256 / rp;
    auto lds = [&](int ktg) { return ((size_t)4 * ktg * N + (size_t)h->JQg * ktg * rp) * sizeof(double2); };
    int ktg = 4;
    while (ktg > 1 && (lds(ktg) > 64 * 1024 || (ktile_limit > 0 && ktile_limit % ktg != 0))) ktg >>= 1;
    if (lds(ktg) > 150 * 1024) return fail(h, KR_ERR_UNSUPPORTED, "N = %d: four state vectors do not fit the CU's LDS (generic kernels)", N);
    h->KTG = ktg;
    h->gen_lds = lds(ktg);
    return KR_OK;
}

// HERM variant of the update kernel: every control operator that enters the update integrand is exactly Hermitian
// (KROTOV_HIP_HERM: 0 = never, 1 = where it was measured to pay: one trajectory per team and >= 64 workgroups -- C3
// 6.63 -> 5.91 us per step, N = 32 / 16 / 8 at P = 256: 5.44 -> 5.07, 4.59 -> 4.38, 4.38 -> 4.26; KT = 2: 9.9 -> 10.1, KT = 4
// at N = 64: 17.6 -> 29 (scratch), P = 8: 4.30 -> 4.45 --, 2 = whenever the operators allow it).  Decided inside make_plan:
// the co-residency query must ask about the variant that will really be launched.
static void decide_herm(kr_handle *h, const kr_config *cfg) {
    const int N = cfg->N, L = cfg->L, G = cfg->G;
    const size_t nn = (size_t)N * N;
    const zc *Hc = (const zc *)cfg->Hc;
    const int mode = opt_int(h, "HERM", 1);
    // (matrix-core sweep: with one tile per workgroup the window hides phase A -- K = 512: 8.0k against ~8.9k cycles for
    //  dot + exchange --, with two tiles its five products outlast the exchange: C4 11.74 vs 11.35 us per step)
    h->herm = (mode == 2) || (mode == 1 && (h->mfma_sweep ? h->KT == 2 : h->KT == 1) && h->P >= 64 && h->NP <= 64);
    if (h->sets > 1) h->herm = false;   // several sets per workgroup: general variant only
    // one-XCD placement (P <= 32): the window variant with wave 1 publishing (kr_sweep_update<.., SCHED 1, HERM 1, XLOC 1>; round 3:
    // with the next step's partial sums deposited by the step's last barrier the publish is the FIRST thing a step does, and
    // phase A hides behind the L2 round trips -- C2 3.50 -> see DESIGN.md) or the general variant with the classic schedule
    // (N = 64: five row-dots of window outlast the L2 round trips: 5.00 vs 5.16 us per step at K = 16 -- general variant there)
    if (h->wg_stride > 1) h->herm = (mode != 0) && h->KT == 1 && h->NP <= 32 && opt_int(h, "XLOC_HERM", 1) != 0;
    if (h->mfma_sweep && h->LMAXv == 4 && h->KT == 4) h->herm = false;   // that instantiation spills (212 B): general variant
    if (h->rowsplit || h->NP > 64) h->herm = false;
    for (int g = 0; g < G && h->herm; ++g) {
        for (int l = 0; l < L && h->herm; ++l) {
            if (cfg->hc_present && !cfg->hc_present[(size_t)g * L + l]) continue;
            const zc *M = Hc + ((size_t)g * L + l) * nn;
            for (int c = 0; c < N && h->herm; ++c)
                for (int r = 0; r <= c; ++r)
                    if (M[(size_t)c * N + r] != std::conj(M[(size_t)r * N + c])) { h->herm = false; break; }
        }
    }
}

// Plan: the persistent kernels where they have one; the generic kernels for what they cannot carry (N > 256, L > 4, more
// trajectories than co-resident workgroups) and for the update sweep whenever the per-step exchange runs outside the kernel
// (RCCL / callback).  KROTOV_HIP_GENERIC = 1 forces the generic kernels everywhere (tests), 0 forbids the fallback.
static int make_plan(kr_handle *h, const kr_config *full, std::vector<int> &wg_gen) {
    const kr_config &c = h->cfg;
    const int32_t *tg = full->traj_generator;
    const int genv = opt_int(h, "GENERIC", -1);
    if (c.L > 32) return fail(h, KR_ERR_UNSUPPORTED, "L = %d > 32 controls not supported", c.L);
    if (c.exchange < KR_EXCHANGE_AUTO || c.exchange > KR_EXCHANGE_CALLBACK) return fail(h, KR_ERR_INVALID, "kr_create: exchange = %d", c.exchange);
    if (c.prop_method < KR_PROP_AUTO || c.prop_method > KR_PROP_NEWTON) return fail(h, KR_ERR_INVALID, "kr_create: prop_method = %d", c.prop_method);
    h->method = c.prop_method;
    const bool nonlinear = c.control_poly_degree > 0 && h->poly_deg > 0;
    if (nonlinear && (h->method == KR_PROP_CHEBY || h->method == KR_PROP_NEWTON))
        return fail(h, KR_ERR_UNSUPPORTED, "non-linear controls with prop_method = CHEBY / NEWTON are not supported (use TAYLOR)");
    // Chebyshev, Hermitian generators (round 5): the persistent update kernels already evaluate the Chebyshev truncation of the step
    // (kr_device.h, "economized step polynomial": the expansion of exp(-i theta x) cut at the same bound, rewritten in powers of the
    // generator, spectral bound per step), and the known-pulse sweeps apply batched propagators built to the same tolerance -- so a
    // CHEBY request is served there whenever the problem has a persistent plan.  The three-term recurrence on ONE spectral range
    // (5 x the guess amplitude, optimize.jl:238-244) stays on the generic kernels: KROTOV_HIP_CHEBY_RECURRENCE=1, or no persistent plan.
    const bool cheb_try = h->method == KR_PROP_CHEBY && genv != 1 && opt_int(h, "CHEBY_RECURRENCE", 0) == 0;
    bool poly = (h->method == KR_PROP_CHEBY && !cheb_try) || h->method == KR_PROP_NEWTON;   // these live in the generic kernels
    // (non-linear controls, round 5: the known-pulse sweeps of every persistent family take the amplitudes f_l(eps) in the place of the
    //  pulses; the update sweep has persistent instantiations for N <= 64 and degree <= 3 -- beyond that the stepped generic update)
    int rc = (genv == 1 || poly) ? KR_ERR_UNSUPPORTED : make_plan_fast(h, tg, wg_gen, c.max_workgroups);
    h->fast_ok = (rc == KR_OK);
    if (rc != KR_OK && rc != KR_ERR_UNSUPPORTED) return rc;
    if (nonlinear && h->fast_ok) h->wg_stride = 1;   // (no one-XCD placement build of the non-linear update kernels)
    // (row-split single / turns kernels: the same scalar change at run time, KrSweepArgs::poly; pair mode: instantiations <.., POLY>;
    //  the streamed-generator kernels -- N > 64 without a row-split plan -- have no such path: stepped generic update)
    h->poly_fast = nonlinear && h->fast_ok && ((h->NP <= 64 && !h->rowsplit && h->sets == 1) || h->rowsplit) && h->poly_deg <= 3 &&
                   c.exchange != KR_EXCHANGE_RCCL && c.exchange != KR_EXCHANGE_CALLBACK;
    if (h->fast_ok) {
        // Co-residency: the P persistent workgroups of an update sweep spin on each other, so all of them must be resident
        // at once.  Ask the runtime how many workgroups of the update kernel one CU holds (registers, LDS) and fail fast --
        // or move to the generic kernels -- instead of launching a grid that would sit in its bounded spin.  (What the
        // runtime cannot know -- other processes on the same GPU -- is the caller's to declare: kr_config.max_workgroups.)
        hipDeviceProp_t prop;
        HIPCHK(h, hipGetDeviceProperties(&prop, h->device));
        decide_herm(h, full);
        KrOccQuery &q = kr_occ_query();
        KrSweepArgs dummy;
        memset(&dummy, 0, sizeof dummy);
        // the variants launch_update may really pick: this handle's HERM, either exchange schedule (calibrate_exchange decides)
        int worst = 1 << 30;
        bool unknown = false;
        for (int sched = 0; sched < 2; ++sched) {
            int kind = 1;
            if (h->NP <= 64 && !h->rowsplit) kind |= (sched ? 16 : 0) | (h->herm ? 64 : 0) | (h->poly_fast ? 256 : 0);
            else if (sched) break;
            q.on = 1; q.blocks_per_cu = -2;
            if (h->rowsplit && h->rs_turns > 0) { dummy.rs_sets = h->rs_turns; dummy.rs_scratch = (double2 *)16; (void)h->launch_rs(h->LMAXv, h->NR, 5, h->P, nullptr, &dummy); }
            else if (h->rowsplit) (void)h->launch_rs(h->LMAXv, h->NR, h->rs_pair ? 3 : 1, h->P, nullptr, &dummy);
            else if (h->mfma_sweep) (void)kr_launch_mf64(h->LMAXv, (h->KT / 2) | (h->sets > 1 ? h->sets << 8 : 0), kind, h->P, nullptr, &dummy);
            else (void)h->launch(h->LMAXv, h->KT, kind, h->P, nullptr, &dummy);
            q.on = 0;
            if (q.blocks_per_cu < 0) unknown = true;   // -2: this launcher has no query mode; -1: the runtime's query failed
            else worst = std::min(worst, q.blocks_per_cu);
        }
        // a failed query is "unknown", not "zero": keep the persistent plan (its spins are bounded) and say so
        if (unknown) h->plan_note = "occupancy query unavailable for the update kernel: co-residency of the persistent grid not verified";
        const long long room = (long long)(worst == (1 << 30) ? (1 << 20) : worst) * prop.multiProcessorCount;
        if (room < h->P) {
            fail(h, KR_ERR_UNSUPPORTED, "the update sweep needs %d co-resident workgroups but this device holds %lld of them (%d per CU x %d CUs)",
                 h->P, room, worst, prop.multiProcessorCount);
            h->fast_ok = false;
            rc = KR_ERR_UNSUPPORTED;
        }
    }
    if (!h->fast_ok) h->poly_fast = false;
    if (cheb_try) {
        if (h->fast_ok) { h->method = KR_PROP_AUTO; h->cheby_served = true; }
        else poly = true;
    }
    if (!h->fast_ok && genv == 0 && !poly) return rc;
    // exchange
    int ex = c.exchange;
    const bool update_fast = h->fast_ok && (h->rs_batch == 0 || h->rs_turns > 0) &&   // (batched row-split plan: only the known-pulse sweeps are persistent)
                             (!nonlinear || h->poly_fast);                              // (non-linear controls beyond the persistent instantiations: stepped generic update)
    if (ex == KR_EXCHANGE_AUTO) ex = update_fast ? KR_EXCHANGE_MAILBOX : (c.world > 1 ? KR_EXCHANGE_RCCL : KR_EXCHANGE_MAILBOX);
    if (ex == KR_EXCHANGE_MAILBOX && !update_fast && c.world > 1)
        return fail(h, KR_ERR_UNSUPPORTED, "this problem runs on the generic kernels (%s); with world > 1 they need exchange = RCCL or CALLBACK", h->err.c_str());
    h->exchange = ex;
    h->gen_known = !h->fast_ok;
    h->gen_update = !update_fast || ex == KR_EXCHANGE_RCCL || ex == KR_EXCHANGE_CALLBACK;
    if (!h->gen_known && !h->gen_update) return KR_OK;
    if (h->fast_ok) {
        // stepped update sweep on the fast plan's trajectory layout: generic tiles subdivide its tiles
        rc = gen_geometry(h, h->KTILE);
        if (rc) return rc;
        h->gen_tiles = h->Kp / h->KTG;
        return KR_OK;
    }
    // everything generic: own layout (generator groups one after the other, each padded to whole tiles of KTG)
    h->plan_note = "generic multi-pass kernels" + (h->err.empty() ? std::string() : " (" + h->err + ")");
    h->err.clear();
    h->mfma_sweep = false; h->rowsplit = false; h->herm = false;
    rc = gen_geometry(h, 0);
    if (rc) return rc;
    h->NP = 0; h->LMAXv = 0; h->threads = 256; h->JQ = h->JQg; h->JS = 0; h->RT = 256; h->SG = 1;
    h->KT = h->KTG; h->KTILE = h->KTG;
    std::vector<int> count(h->G, 0);
    for (int k = 0; k < c.K; ++k) count[tg ? tg[k] : 0]++;
    int tiles = 0;
    for (int g = 0; g < h->G; ++g) tiles += (count[g] + h->KTG - 1) / h->KTG;
    h->P = tiles; h->gen_tiles = tiles; h->Kp = tiles * h->KTG;
    h->perm.assign(c.K, 0);
    wg_gen.assign(tiles, 0);
    std::vector<int> tile_base(h->G, 0), seen(h->G, 0);
    int t = 0;
    for (int g = 0; g < h->G; ++g) {
        tile_base[g] = t;
        const int nt = (count[g] + h->KTG - 1) / h->KTG;
        for (int q = 0; q < nt; ++q) wg_gen[t + q] = g;
        t += nt;
    }
    h->identity_perm = true;
    for (int k = 0; k < c.K; ++k) {
        const int g = tg ? tg[k] : 0;
        h->perm[k] = tile_base[g] * h->KTG + seen[g]++;
        if (h->perm[k] != k) h->identity_perm = false;
    }
    return KR_OK;
}

extern "C" int kr_abi_version(void) { return KR_ABI_VERSION; }

extern "C" const char *kr_last_error(const kr_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

static void special_release(kr_handle *h, void *p, size_t bytes, int kind, unsigned long long gen);
static void rccl_release(kr_handle *h);
static void free_all(kr_handle *h) {
    void *ptrs[] = {h->Hpack_fw, h->Hpack_bw, h->pack_words_d, h->dt_d, h->eps_a, h->eps_b, h->S_d, h->lambda_d, h->gint_d, h->psi0_d,
                    h->tgt_d, h->psiT_d, h->chiT_d, h->X_d, h->Phi_d, h->tau_d, h->tausum_d, h->status_d, h->debug_d, h->U_d, h->wg_gen_d, h->nrm_d, h->present_d, h->gmail_d, h->Hdense_fw, h->Hcol_fw, h->Hcol_bw, h->Heps_d, h->econ_d, h->skw_d,
                    h->tile_gen_d, h->nrm_gen_d, h->partial_d, h->du_local_d, h->du_total_d, h->ptab_d, h->erange_d, h->poly_d, h->rs_scratch_d,
                    h->tb_P_d, h->tb_bound_d, h->tb_items_d, h->dulog_d, h->poly3_d, h->amp_d, h->fprime_d, h->tb_T_d[0], h->tb_T_d[1]};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (h->peers_connected && !h->peers_in_process)   // (cached mappings stay open: ipc_open)
        for (int r = 0; r < h->cfg.world; ++r)
            if (r != h->cfg.rank && h->peer_xmail[r] && !h->peer_cached[r]) (void)hipIpcCloseMemHandle(h->peer_xmail[r]);
    if (h->mail_d) { if (h->mail_kind) special_release(h, h->mail_d, h->mail_words * sizeof(unsigned long long), h->mail_kind, 0); else (void)hipFree(h->mail_d); }
    // the mailbox between GPUs goes back to the pool WHATEVER its kind: peers in other processes may still be storing into it
    if (h->xmail_d) special_release(h, h->xmail_d, h->xmail_bytes, h->xmail_kind, h->xmail_gen);
    if (h->du_host) (void)hipHostFree(h->du_host);
    rccl_release(h);
    for (auto &pool : h->ev) for (auto &e : pool) if (e) (void)hipEventDestroy(e);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
}

static int one_destroy(kr_handle *h) {
    if (!h) return KR_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    free_all(h);
    delete h;
    return KR_OK;
}

template <class T>
static hipError_t dalloc(T **p, size_t count) {
    return hipMalloc((void **)p, std::max<size_t>(count, 1) * sizeof(T));
}

// Mailbox memory for the in-kernel exchanges.  With many workgroups polling, an UNCACHED allocation (MTYPE UC: the L2s
// keep no copy, every access goes to memory) makes the partial sums visible sooner than ordinary hipMalloc memory accessed
// with sc1: same-box A/B of the update sweep, C3 (P = 256): 7.17-7.25 -> 6.85-6.89 us per step, C4: 18.4 -> 17.9;
// fine-grained memory: no gain (7.17-7.25); with few workgroups ordinary memory is slightly faster (C2, P = 8: 4.16
// against 4.22).  KROTOV_HIP_MAIL_MEM = 0 (hipMalloc) / 1 (uncached) / 2 (fine-grained) overrides the choice.
// Uncached / fine-grained blocks are never handed back to the runtime while the library is loaded: a handle returns its block to a
// process-wide pool (per device, kind and size) and the next handle that asks for the same takes it from there (KROTOV_HIP_UC_POOL=0:
// hipFree as before).  See DESIGN.md, "device lists": [round 4] with blocks of both kinds allocated and freed in turn by the handles
// of one process, partial sums published into an ORDINARY mailbox went missing.
struct KrSpecialBlock { void *p; int device, kind; size_t bytes; unsigned long long gen; };
static std::mutex g_special_mtx;
static std::vector<KrSpecialBlock> g_special_free;
// kind: 1 uncached, 2 fine-grained, 0 ordinary device memory (the mailbox between GPUs when neither special kind is available: a block
// peers store into must outlive every kernel of theirs that may still be spinning towards it -- other processes cannot be waited for,
// so such a block goes back to the pool whatever its kind).  *gen (optional): how often this process has handed the block out, counting
// from 1 -- ranks mix the generations of all their mailboxes into the tag base of a handle (kr_ipc_connect), so that a store that a late
// peer of an EARLIER handle still sends into a recycled block never carries a tag the new handle waits for.
static hipError_t special_alloc(kr_handle *h, void **p, size_t bytes, int kind, unsigned long long *gen = nullptr) {
    if (opt_int(h, "UC_POOL", 1) != 0) {
        std::lock_guard<std::mutex> g(g_special_mtx);
        for (size_t i = 0; i < g_special_free.size(); ++i)
            if (g_special_free[i].device == h->device && g_special_free[i].kind == kind && g_special_free[i].bytes == bytes) {
                *p = g_special_free[i].p;
                if (gen) *gen = g_special_free[i].gen + 1;
                g_special_free.erase(g_special_free.begin() + (long)i);
                return hipSuccess;
            }
    }
    if (gen) *gen = 1;
    if (kind == 0) return hipMalloc(p, bytes);
    return hipExtMallocWithFlags(p, bytes, kind == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
}
static void special_release(kr_handle *h, void *p, size_t bytes, int kind, unsigned long long gen = 0) {
    if (!p) return;
    if (opt_int(h, "UC_POOL", 1) != 0) {
        try {
            std::lock_guard<std::mutex> g(g_special_mtx);
            g_special_free.push_back({p, h->device, kind, bytes, gen});
            return;
        } catch (...) {}
    }
    (void)hipFree(p);
}
// A block that failed the self-test of kr_create is never used and never handed back (the runtime would serve the next request from
// the same address range): it stays allocated for the life of the process.
static std::mutex g_quarantine_mtx;
static std::vector<void *> g_quarantine;
static void quarantine_block(void *p) {
    try { std::lock_guard<std::mutex> g(g_quarantine_mtx); g_quarantine.push_back(p); } catch (...) {}
}
static hipError_t alloc_mailbox(kr_handle *h, unsigned long long **p, size_t words, int pollers) {
    const int mm = opt_int(h, "MAIL_MEM", pollers >= 64 ? 1 : 0);
    h->mail_kind = 0;
    if (mm != 0) {
        const hipError_t e = special_alloc(h, (void **)p, std::max<size_t>(words, 1) * sizeof(unsigned long long), mm == 1 ? 1 : 2);
        if (e == hipSuccess) { h->mail_kind = mm == 1 ? 1 : 2; return e; }
        (void)hipGetLastError();  // not available on this device/driver: fall back to ordinary memory
    }
    return dalloc(p, words);
}

// one run of kr_mail_selftest on a region; st[0] != 0 afterwards: the region failed (st[2]: workgroup | round << 16 | replica << 24)
static int mailbox_selftest(kr_handle *h, unsigned long long *base, int P, int replicas, size_t rstride, int sgran, int width, int fault, unsigned (&st)[KR_STATUS_WORDS]) {
    hipLaunchKernelGGL(kr_mail_selftest, dim3(P), dim3(64), 0, h->own_stream, base, P, replicas, rstride, sgran, width, (unsigned long long)10000000ull /* 100 ms */, fault, h->status_d);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(st, h->status_d, sizeof st, hipMemcpyDeviceToHost, h->own_stream));
    HIPCHK(h, hipStreamSynchronize(h->own_stream));
    if (st[0] != 0) {
        HIPCHK(h, hipMemsetAsync(h->status_d, 0, KR_STATUS_WORDS * sizeof(unsigned), h->own_stream));
        HIPCHK(h, hipStreamSynchronize(h->own_stream));
    }
    return KR_OK;
}

static int calibrate_exchange(kr_handle *h);  // defined below, next to launch_update
static int check_status(kr_handle *h);

static int one_create_impl(const kr_config *cfg, kr_handle **out, kr_handle *&live) {
    if (out) *out = nullptr;
    if (!cfg || !out) return fail(nullptr, KR_ERR_INVALID, "kr_create: null argument");
    // the version FIRST: v3 gave words a meaning that v2 called reserved (spin_timeout_ms, the options pointer) -- nothing below may
    // look at a config of another version
    if (cfg->abi_version != KR_ABI_VERSION) return fail(nullptr, KR_ERR_INVALID, "kr_create: abi_version %d != %d", cfg->abi_version, KR_ABI_VERSION);
    if (cfg->spin_timeout_ms < 0) return fail(nullptr, KR_ERR_INVALID, "kr_create: spin_timeout_ms = %d must be >= 0 (0 = default 4000)", cfg->spin_timeout_ms);
    if (cfg->L <= 0) return fail(nullptr, KR_ERR_INVALID, "no controls in trajectories: cannot optimize");  // workspace.jl:72
    if (cfg->N <= 0 || cfg->NT <= 0 || cfg->K <= 0) return fail(nullptr, KR_ERR_INVALID, "kr_create: N, NT, K must be positive (got %d, %d, %d)", cfg->N, cfg->NT, cfg->K);
    if (!cfg->tlist || !cfg->H0 || !cfg->Hc) return fail(nullptr, KR_ERR_INVALID, "kr_create: tlist, H0, Hc are required");
    if (cfg->G < 1) return fail(nullptr, KR_ERR_INVALID, "kr_create: G = %d must be >= 1", cfg->G);
    if (cfg->G > 1 && !cfg->traj_generator) return fail(nullptr, KR_ERR_INVALID, "kr_create: G = %d > 1 needs traj_generator", cfg->G);
    if (cfg->traj_generator)
        for (int k = 0; k < cfg->K; ++k)
            if (cfg->traj_generator[k] < 0 || cfg->traj_generator[k] >= cfg->G) return fail(nullptr, KR_ERR_INVALID, "kr_create: traj_generator[%d] = %d out of range", k, cfg->traj_generator[k]);
    if (cfg->world < 1 || cfg->rank < 0 || cfg->rank >= cfg->world || cfg->world > 8) return fail(nullptr, KR_ERR_INVALID, "kr_create: bad rank/world %d/%d", cfg->rank, cfg->world);
    for (int n = 0; n < cfg->NT; ++n)
        if (!(cfg->tlist[n + 1] > cfg->tlist[n])) return fail(nullptr, KR_ERR_INVALID, "kr_create: tlist must be strictly increasing (index %d)", n);

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, KR_ERR_NO_DEVICE, "no HIP device available (%s); libkrotov_hip has no CPU fallback", e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    int dev = cfg->device;
    // a device list of ONE entry names the device (kr_create only takes the group path for n_devices > 1)
    if (cfg->n_devices == 1) dev = cfg->devices[0];
    else if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev < 0 || dev >= ndev) return fail(nullptr, KR_ERR_INVALID, "kr_create: device %d out of range (%d devices)", dev, ndev);

    kr_handle *h = new kr_handle();
    live = h;
    if (cfg->spin_timeout_ms > 0) h->spin_ticks = (unsigned long long)cfg->spin_timeout_ms * 100000ull;   // s_memrealtime: 100 MHz
    h->cfg = *cfg;
    h->cfg.options = nullptr;
    {
        std::string bad;
        if (!parse_options(cfg->options, h->options, bad)) {
            live = nullptr;
            delete h;
            return fail(nullptr, KR_ERR_INVALID, "kr_create: kr_config.options: '%s' is not NAME=integer with a known NAME (INTEGRATION.md lists them)", bad.c_str());
        }
    }
    h->cfg.traj_generator = nullptr; h->cfg.tlist = nullptr; h->cfg.H0 = nullptr; h->cfg.Hc = nullptr; h->cfg.hc_present = nullptr;
    if (h->cfg.K_global <= 0) h->cfg.K_global = cfg->K;
    h->device = dev;
    h->store_fw = !(cfg->flags & KR_NO_STORE_FW);
    h->tol = (cfg->prop_tol > 0) ? cfg->prop_tol : 1e-16;
    {   // thr[m]: largest theta in [0, KR_THETA_MAX] whose order-m remainder bound meets tol (bisection)
        const char *et = getenv("KROTOV_HIP_TOL");
        if (et && atof(et) > 0) h->tol = atof(et);
        for (int m = 1; m < KR_MAX_ORDER; ++m) {
            auto rem = [&](double th) {
                double term = th;
                for (int q = 1; q <= m; ++q) term = term * th / (double)(q + 1);
                return term / (1.0 - th / (double)(m + 2));
            };
            double lo = 0.0, hi = KR_THETA_MAX;
            if (rem(hi) <= h->tol) lo = hi;
            else for (int it = 0; it < 100; ++it) { const double mid = 0.5 * (lo + hi); if (rem(mid) <= h->tol) lo = mid; else hi = mid; }
            h->thr[m] = lo;
        }
        h->thr[0] = 0.0;
    }
    int rc = KR_OK;
    auto bail = [&](int code) { g_create_error = h->err; (void)hipSetDevice(dev); free_all(h); live = nullptr; delete h; return code; };
#define CHK(call) do { hipError_t e2_ = (call); if (e2_ != hipSuccess) { fail(h, KR_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e2_)); return bail(KR_ERR_HIP); } } while (0)
    CHK(hipSetDevice(dev));
    h->G = cfg->G;
    h->cfg.control_poly = nullptr;
    if (cfg->control_poly_degree != 0 || cfg->control_poly) {
        if (cfg->control_poly_degree < 1 || cfg->control_poly_degree > 7 || !cfg->control_poly) {
            fail(h, KR_ERR_INVALID, "kr_create: control_poly needs a degree in 1..7 and the coefficient array (got degree %d)", cfg->control_poly_degree);
            return bail(KR_ERR_INVALID);
        }
        h->poly_deg = cfg->control_poly_degree;
        CHK(dalloc(&h->poly_d, (size_t)cfg->L * (h->poly_deg + 1)));
        CHK(hipMemcpy(h->poly_d, cfg->control_poly, (size_t)cfg->L * (h->poly_deg + 1) * sizeof(double), hipMemcpyHostToDevice));
    }
    std::vector<int> wg_gen;
    rc = make_plan(h, cfg, wg_gen);
    if (rc != KR_OK) return bail(rc);
    CHK(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    CHK(dalloc(&h->status_d, KR_STATUS_WORDS));
    CHK(hipMemsetAsync(h->status_d, 0, KR_STATUS_WORDS * sizeof(unsigned), h->own_stream));   // (own stream: the self-test kernels of kr_create run there)
    CHK(hipStreamSynchronize(h->own_stream));

    const int N = cfg->N, NT = cfg->NT, L = cfg->L, NP = h->NP;
    const size_t nn = (size_t)N * N;
    // norms, presence masks and packs, per generator
    const int G = h->G;
    const zc *H0 = (const zc *)cfg->H0, *Hc = (const zc *)cfg->Hc;
    std::vector<double> nrm((size_t)G * (1 + KR_LMAX_CAP), 0.0), nrm_gen((size_t)G * (1 + L), 0.0);
    std::vector<unsigned> present(G, 0u);
    h->pack_stride = (size_t)(1 + L) * NP * NP;  // = (1+L) * JS * RT in every geometry
    const size_t pack_elems = h->fast_ok ? h->pack_stride * G : 0;
    std::vector<double2> pack(pack_elems);
    if (h->fast_ok) {
        CHK(dalloc(&h->Hpack_fw, pack_elems));
        CHK(dalloc(&h->Hpack_bw, pack_elems));
    }
    {
        std::vector<const zc *> mats((size_t)G * (1 + L));
        for (int g = 0; g < G; ++g) {
            mats[(size_t)g * (1 + L)] = H0 + (size_t)g * nn;
            for (int l = 0; l < L; ++l) mats[(size_t)g * (1 + L) + 1 + l] = Hc + ((size_t)g * L + l) * nn;
        }
        CHK(kr_norm2_upper_bounds(N, (int)mats.size(), mats.data(), nrm_gen.data()));
        h->nrm_host = nrm_gen;
    }
    for (int g = 0; g < G; ++g) {
        for (int l = 0; l < L; ++l)
            if (!cfg->hc_present || cfg->hc_present[(size_t)g * L + l]) present[g] |= 1u << l;
        for (int m = 0; m <= L && m <= KR_LMAX_CAP; ++m) nrm[(size_t)g * (1 + KR_LMAX_CAP) + m] = nrm_gen[(size_t)g * (1 + L) + m];
    }
    // Economized step polynomial (KrEcon, kr_device.h): needs a real spectrum, i.e. EVERY operator of every generator exactly
    // Hermitian; register-resident update kernels only (N <= 64).  KROTOV_HIP_ECON = 0 switches it off.
    // Row-split PAIR mode (round 3): also for generators that are only NEARLY Hermitian (kr_econ_rows_kernel decides per step from
    // the bounds of the anti-Hermitian parts whether an economized row exists; KROTOV_HIP_ECON_PAIR = 0 switches that off).
    const bool econ_pair = opt_int(h, "ECON", 1) != 0 && opt_int(h, "ECON_PAIR", 1) != 0 && h->fast_ok && !h->gen_update && (h->rowsplit || h->NP > 64);   // row-split and streamed-generator kernels
    if (econ_pair) {
        std::vector<double> skw((size_t)G * (1 + KR_LMAX_CAP), 0.0);
        std::vector<zc> Ms(nn);
        bool any = false;
        for (int g = 0; g < G; ++g)
            for (int m = 0; m <= L && m <= KR_LMAX_CAP; ++m) {
                if (m > 0 && !((present[g] >> (m - 1)) & 1u)) continue;
                const zc *M = (m == 0) ? H0 + (size_t)g * nn : Hc + ((size_t)g * L + (m - 1)) * nn;
                bool herm = true;
                for (int c = 0; c < N; ++c)
                    for (int r = 0; r < N; ++r) {
                        Ms[(size_t)c * N + r] = 0.5 * (M[(size_t)c * N + r] - std::conj(M[(size_t)r * N + c]));
                        if (Ms[(size_t)c * N + r] != zc(0.0, 0.0)) herm = false;
                    }
                if (!herm) {
                    const zc *mp = Ms.data();
                    CHK(kr_norm2_upper_bounds(N, 1, &mp, &skw[(size_t)g * (1 + KR_LMAX_CAP) + m]));
                    any = true;
                }
            }
        if (any) {
            CHK(dalloc(&h->skw_d, skw.size()));
            CHK(hipMemcpy(h->skw_d, skw.data(), skw.size() * sizeof(double), hipMemcpyHostToDevice));
        }
    }
    h->econ = opt_int(h, "ECON", 1) != 0 && h->fast_ok && !h->gen_update && !h->rowsplit && h->NP <= 64;
    for (int g = 0; g < G && h->econ; ++g) {
        for (int m = 0; m <= L && h->econ; ++m) {
            if (m > 0 && !((present[g] >> (m - 1)) & 1u)) continue;
            const zc *M = (m == 0) ? H0 + (size_t)g * nn : Hc + ((size_t)g * L + (m - 1)) * nn;
            for (int c = 0; c < N && h->econ; ++c)
                for (int r = 0; r <= c; ++r)
                    if (M[(size_t)c * N + r] != std::conj(M[(size_t)r * N + c])) { h->econ = false; break; }
        }
    }
    if (econ_pair) h->econ = true;
    // generic (multi-pass) kernels: Hermitian generators only (entry by entry), Taylor method, linear controls
    if (opt_int(h, "ECON", 1) != 0 && (h->gen_known || h->gen_update) && (h->method == KR_PROP_AUTO || h->method == KR_PROP_TAYLOR) && h->poly_deg == 0) {
        bool herm_all = true;
        for (int g = 0; g < G && herm_all; ++g)
            for (int m = 0; m <= L && herm_all; ++m) {
                if (m > 0 && L <= 32 && !((present[g] >> (m - 1)) & 1u)) continue;
                const zc *M = (m == 0) ? H0 + (size_t)g * nn : Hc + ((size_t)g * L + (m - 1)) * nn;
                for (int c = 0; c < N && herm_all; ++c)
                    for (int r = 0; r <= c; ++r)
                        if (M[(size_t)c * N + r] != std::conj(M[(size_t)r * N + c])) { herm_all = false; break; }
            }
        h->econ_gen = herm_all;
    }
    if (h->econ || h->econ_gen) {
        CHK(dalloc(&h->econ_d, (size_t)G * NT * KR_ECON_W));
        CHK(hipMemset(h->econ_d, 0, (size_t)G * NT * KR_ECON_W * sizeof(double)));
    }
    for (int adj = 0; adj < 2 && h->fast_ok; ++adj) {
        for (int g = 0; g < G; ++g)
            for (int m = 0; m <= L; ++m)
            {
                const zc *M = (m == 0) ? H0 + (size_t)g * nn : Hc + ((size_t)g * L + (m - 1)) * nn;
                double2 *dst = pack.data() + (size_t)g * h->pack_stride + (size_t)m * NP * NP;
                if (h->rowsplit) pack_generator_rowsplit(N, NP, h->NR, M, adj != 0, dst, h->rs_pair);
                else pack_generator(N, NP, h->JQ, M, adj != 0, dst);
            }
        CHK(hipMemcpy(adj ? h->Hpack_bw : h->Hpack_fw, pack.data(), pack_elems * sizeof(double2), hipMemcpyHostToDevice));
        CHK(hipDeviceSynchronize());   // (the host vector is refilled for the adjoint pack: the copy must have left it)
        h->pack_sum[adj] = pack_checksum_host(pack.data(), pack_elems);
    }
    if (opt_int(h, "PACK_CHECK", 1) != 0) {
        CHK(dalloc(&h->pack_words_d, 2));
        CHK(hipMemset(h->pack_words_d, 0, 2 * sizeof(unsigned long long)));
    }
    if (h->fast_ok) {
        h->pack_elems = pack_elems;
        h->const_elems[0] = h->const_elems[1] = pack_elems;
        if (opt_int(h, "PACK_FAULT", 0) != 0) {   // diagnostics (tests): one mantissa bit of one element of the ADJOINT pack flipped after the upload
            double2 v;
            const size_t at = pack_elems / 3;
            CHK(hipMemcpy(&v, h->Hpack_bw + at, sizeof v, hipMemcpyDeviceToHost));
            unsigned long long b;
            memcpy(&b, &v.x, 8); b ^= 1ull << 20; memcpy(&v.x, &b, 8);
            CHK(hipMemcpy(h->Hpack_bw + at, &v, sizeof v, hipMemcpyHostToDevice));
        }
    }
    if (h->mfma_sweep) {   // dense row-major copies, zero-padded to 64 x 64, for the A operands of the matrix-core sweep
        h->dense_stride = (size_t)(1 + L) * 64 * 64;
        std::vector<double2> dense(h->dense_stride * G, make_double2(0.0, 0.0));
        for (int g = 0; g < G; ++g)
            for (int m = 0; m <= L; ++m) {
                const zc *M = (m == 0) ? H0 + (size_t)g * nn : Hc + ((size_t)g * L + (m - 1)) * nn;
                double2 *dst = dense.data() + (size_t)g * h->dense_stride + (size_t)m * 64 * 64;
                for (int i = 0; i < N; ++i)
                    for (int j = 0; j < N; ++j) { const zc v = M[i + (size_t)j * N]; dst[(size_t)i * 64 + j] = make_double2(v.real(), v.imag()); }
            }
        CHK(dalloc(&h->Hdense_fw, dense.size()));
        CHK(hipMemcpy(h->Hdense_fw, dense.data(), dense.size() * sizeof(double2), hipMemcpyHostToDevice));
        h->pack_sum[2] = pack_checksum_host(dense.data(), dense.size());
        h->const_elems[2] = dense.size();
    }
    h->lam0 = 0.0;
    h->lam_l.assign(L, 0.0);
    for (int g = 0; g < G; ++g) {
        h->lam0 = std::max(h->lam0, nrm_gen[(size_t)g * (1 + L)]);
        for (int l = 0; l < L; ++l) h->lam_l[l] = std::max(h->lam_l[l], nrm_gen[(size_t)g * (1 + L) + 1 + l]);
    }
    h->dt_host.assign(NT, 0.0);
    for (int n = 0; n < NT; ++n) h->dt_host[n] = cfg->tlist[n + 1] - cfg->tlist[n];
    if (cfg->prop_method == KR_PROP_CHEBY) {   // Chebyshev needs real spectra: every operator exactly Hermitian
        for (int g = 0; g < G; ++g)
            for (int m = 0; m <= L; ++m) {
                const zc *M = (m == 0) ? H0 + (size_t)g * nn : Hc + ((size_t)g * L + (m - 1)) * nn;
                for (int c2 = 0; c2 < N; ++c2)
                    for (int r = 0; r <= c2; ++r)
                        if (M[(size_t)c2 * N + r] != std::conj(M[(size_t)r * N + c2])) {
                            fail(h, KR_ERR_INVALID, "prop_method = CHEBY needs Hermitian generators (operator %d of generator %d is not); use NEWTON or TAYLOR", m, g);
                            return bail(KR_ERR_INVALID);
                        }
            }
    }
    if (h->method == KR_PROP_CHEBY || h->method == KR_PROP_NEWTON) {
        h->ptab_stride = 4 + 3 * (KR_MAX_ORDER + 2);
        CHK(dalloc(&h->ptab_d, (size_t)NT * h->ptab_stride));
        CHK(dalloc(&h->erange_d, (size_t)L));
    }
    if (h->gen_known || h->gen_update) {
        // generic kernels: the operators column-major as the caller hands them over (+ the adjoint, column-major), the
        // generator of every generic tile, norms for any L, scratch for the per-step pipeline
        h->col_stride = (size_t)(1 + L) * nn;
        std::vector<double2> col(h->col_stride * G);
        for (int adj = 0; adj < 2; ++adj) {
            if (adj == 0 && !h->gen_update && !h->gen_known) continue;
            if (adj == 1 && !h->gen_known) continue;
            for (int g = 0; g < G; ++g)
                for (int m = 0; m <= L; ++m) {
                    const zc *M = (m == 0) ? H0 + (size_t)g * nn : Hc + ((size_t)g * L + (m - 1)) * nn;
                    double2 *dst = col.data() + (size_t)g * h->col_stride + (size_t)m * nn;
                    for (int j = 0; j < N; ++j)
                        for (int i = 0; i < N; ++i) {
                            const zc v = adj ? std::conj(M[j + (size_t)i * N]) : M[i + (size_t)j * N];
                            dst[i + (size_t)j * N] = make_double2(v.real(), v.imag());
                        }
                }
            double2 **dstp = adj ? &h->Hcol_bw : &h->Hcol_fw;
            CHK(dalloc(dstp, col.size()));
            CHK(hipMemcpy(*dstp, col.data(), col.size() * sizeof(double2), hipMemcpyHostToDevice));
            CHK(hipDeviceSynchronize());   // (the host vector is refilled for the adjoint copy)
            h->pack_sum[3 + adj] = pack_checksum_host(col.data(), col.size());
            h->const_elems[3 + adj] = col.size();
        }
        std::vector<int> tile_gen(h->gen_tiles, 0);
        const int per = h->KTILE / h->KTG;   // generic tiles per plan tile
        for (int t = 0; t < h->gen_tiles; ++t) tile_gen[t] = wg_gen.empty() ? 0 : wg_gen[std::min((size_t)(t / std::max(per, 1)), wg_gen.size() - 1)];
        CHK(dalloc(&h->tile_gen_d, tile_gen.size()));
        CHK(hipMemcpy(h->tile_gen_d, tile_gen.data(), tile_gen.size() * sizeof(int), hipMemcpyHostToDevice));
        CHK(dalloc(&h->nrm_gen_d, nrm_gen.size()));
        CHK(hipMemcpy(h->nrm_gen_d, nrm_gen.data(), nrm_gen.size() * sizeof(double), hipMemcpyHostToDevice));
        if (h->gen_update) {
            CHK(dalloc(&h->Heps_d, (size_t)G * nn));
            CHK(dalloc(&h->partial_d, (size_t)h->gen_tiles * L));
            CHK(dalloc(&h->du_local_d, (size_t)L));
            CHK(dalloc(&h->du_total_d, (size_t)L));
            CHK(hipHostMalloc((void **)&h->du_host, (size_t)L * sizeof(double), hipHostMallocDefault));
        }
    }
    CHK(dalloc(&h->nrm_d, nrm.size()));
    CHK(hipMemcpy(h->nrm_d, nrm.data(), nrm.size() * sizeof(double), hipMemcpyHostToDevice));
    CHK(dalloc(&h->present_d, (size_t)G));
    CHK(hipMemcpy(h->present_d, present.data(), G * sizeof(unsigned), hipMemcpyHostToDevice));
    CHK(dalloc(&h->wg_gen_d, wg_gen.size()));
    CHK(hipMemcpy(h->wg_gen_d, wg_gen.data(), wg_gen.size() * sizeof(int), hipMemcpyHostToDevice));
    if (h->rowsplit) {
        // vector mailboxes [K][2][NP][4] + one word per workgroup (XCD ids, see kr_rowsplit_sweep); ordinary cached
        // memory: the XCD-local path of the all-gather lives on L2 hits
        h->gmail_words = (size_t)h->Kp * 2 * NP * 4 + (size_t)h->P;
        // (the kernels lay the workgroup ids out by their TILES = groups of R workgroups of a launch: trajectories, pairs in pair mode,
        //  groups with sets or turns -- not by the trajectory count, which differs from it with sets / turns / batches)
        h->xcd_group = ((h->P / h->R) % 8 == 0) && opt_int(h, "XCD_GROUP", 1) != 0;
        CHK(dalloc(&h->gmail_d, h->gmail_words));
        CHK(hipMemsetAsync(h->gmail_d, 0, h->gmail_words * sizeof(unsigned long long), h->own_stream)); CHK(hipStreamSynchronize(h->own_stream));
        if (opt_int(h, "MAIL_SELFTEST", 1) != 0) {   // the all-gather's mailbox: P words spread over the whole block
            unsigned st[KR_STATUS_WORDS] = {};
            rc = mailbox_selftest(h, h->gmail_d, h->P, 1, 0, (int)std::max<size_t>(1, h->gmail_words / (size_t)h->P), 1, 0, st);
            if (rc != KR_OK) return bail(rc);
            if (st[0] != 0) {
                quarantine_block(h->gmail_d); h->gmail_d = nullptr;
                fail(h, KR_ERR_INTERNAL, "kr_create: the freshly allocated all-gather mailbox failed its self-test (workgroup %u did not see every published word within 100 ms): "
                     "device memory of this process loses stores or hides them from some XCDs; the block is set aside -- call kr_create again", st[2] & 0xffffu);
                return bail(KR_ERR_INTERNAL);
            }
            CHK(hipMemsetAsync(h->gmail_d, 0, h->gmail_words * sizeof(unsigned long long), h->own_stream)); CHK(hipStreamSynchronize(h->own_stream));
        }
        if (h->rs_sets > 1) CHK(dalloc(&h->rs_scratch_d, (size_t)h->P * 4 /* KR_PAIR_MAX_SETS */ * 2 * NP));
        if (h->rs_turns > 0) CHK(dalloc(&h->rs_scratch_d, (size_t)h->P * KR_RS_MAX_TURNS * NP));
    }
    std::vector<double> dt(NT);
    for (int n = 0; n < NT; ++n) dt[n] = cfg->tlist[n + 1] - cfg->tlist[n];
    CHK(dalloc(&h->dt_d, NT));
    CHK(hipMemcpy(h->dt_d, dt.data(), sizeof(double) * NT, hipMemcpyHostToDevice));
    CHK(dalloc(&h->eps_a, (size_t)L * NT));
    CHK(dalloc(&h->eps_b, (size_t)L * NT));
    CHK(dalloc(&h->S_d, (size_t)L * NT));
    if (h->poly_deg > 0 && h->fast_ok) {   // non-linear controls on the persistent kernels: amplitudes (and derivatives at the guess) of the sweep at hand
        CHK(dalloc(&h->amp_d, (size_t)L * NT));
        CHK(dalloc(&h->fprime_d, (size_t)L * NT));
        if (h->poly_fast) {
            std::vector<double> p3((size_t)L * 4, 0.0);
            for (int l = 0; l < L; ++l)
                for (int q = 0; q <= h->poly_deg; ++q) p3[(size_t)l * 4 + q] = cfg->control_poly[(size_t)l * (h->poly_deg + 1) + q];
            CHK(dalloc(&h->poly3_d, p3.size()));
            CHK(hipMemcpy(h->poly3_d, p3.data(), p3.size() * sizeof(double), hipMemcpyHostToDevice));
        }
    }
    CHK(dalloc(&h->lambda_d, L));
    CHK(dalloc(&h->gint_d, std::max(L, KR_LMAX_CAP)));
    const size_t slab = (size_t)h->Kp * N;
    CHK(dalloc(&h->psi0_d, slab));
    CHK(dalloc(&h->tgt_d, slab));
    CHK(dalloc(&h->psiT_d, slab));
    CHK(dalloc(&h->chiT_d, slab));
    CHK(dalloc(&h->X_d, slab * (NT + 1)));
    if (h->store_fw) CHK(dalloc(&h->Phi_d, slab * (NT + 1)));
    // Known-pulse sweeps with precomputed propagators (kr_device_prop.h): N <= 64, and the U buffer
    // (NT * NP^2 * 16 B) within a budget; KROTOV_HIP_PRECOMP=0 keeps the Taylor sweeps.
    {
        const size_t ubytes = (size_t)G * NT * NP * NP * sizeof(double2);
        h->precomp = h->fast_ok && (NP <= 64) && opt_int(h, "PRECOMP", 1) != 0 && ubytes <= ((size_t)8 << 30);
        h->use_mfma = opt_int(h, "EXPM_MFMA", 1) != 0;
        if (h->precomp) CHK(dalloc(&h->U_d, (size_t)G * NT * NP * NP));
        // ... and in blocks of time (round 5): NP = 64 with the matrix-core propagator build, grids long enough for the two extra
        // passes to pay (TBLOCK_MIN_NT), the boundary buffer within a budget.  KROTOV_HIP_TBLOCK = steps per block, 0 = off.
        // Steps per block: KROTOV_HIP_TBLOCK (0 = off), or -- default -- by the shape: the parallel pass (iv) wants at least one workgroup
        // per CU (blocks x pieces of <= 64 trajectories), the sequential pass (iii) as few blocks as possible.  Measured, backward sweep
        // [ms] at S = 8 / 16 / 32 with the tree of products: C3 (4 pieces) 0.50 / 0.46 / 0.57, C4 (16 pieces) 1.01 / 0.94 / 0.92
        // (profiles/r05_tb_tree.log).
        if (h->precomp && NP == 64 && h->use_mfma && NT >= opt_int(h, "TBLOCK_MIN_NT", 64)) {
            // pieces: runs of internal slots with one generator (the slots are grouped by generator, every group padded to whole tiles
            // with zero trajectories), cut into pieces of at most 64
            std::vector<int> items;
            const int tiles = (int)wg_gen.size(), ktile = h->KTILE;
            int t0 = 0;
            while (t0 < tiles) {
                int t1 = t0;
                while (t1 < tiles && wg_gen[t1] == wg_gen[t0]) ++t1;
                for (int k0 = t0 * ktile; k0 < t1 * ktile; k0 += 64) { items.push_back(k0); items.push_back(std::min(64, t1 * ktile - k0)); items.push_back(wg_gen[t0]); items.push_back(0); }
                t0 = t1;
            }
            const int nitems = (int)items.size() / 4;
            int S = opt_int(h, "TBLOCK", -1);
            if (S < 0) {
                hipDeviceProp_t prop;
                CHK(hipGetDeviceProperties(&prop, h->device));
                const int cus = prop.multiProcessorCount;
                auto wgs = [&](int s_) { return (long long)((NT + s_ - 1) / s_) * nitems; };
                S = (wgs(32) >= cus) ? 32 : (wgs(16) * 2 >= cus) ? 16 : 8;
            }
            if (S >= 2 && S <= 64 && NT >= 2 * S) {
                const int NB = (NT + S - 1) / S;
                const size_t bbytes = (size_t)(NB + 1) * h->Kp * N * sizeof(double2);
                if (bbytes <= ((size_t)4 << 30)) {
                    h->tb_S = S; h->tb_NB = NB; h->tb_nitems = nitems;
                    CHK(dalloc(&h->tb_P_d, (size_t)G * NB * 4096));
                    h->tb_tree = (S & (S - 1)) == 0 && S >= 4 && opt_int(h, "TBLOCK_TREE", 1) != 0;
                    if (h->tb_tree) {
                        CHK(dalloc(&h->tb_T_d[0], (size_t)G * ((NT + 1) / 2) * 4096));
                        CHK(dalloc(&h->tb_T_d[1], (size_t)G * ((NT + 3) / 4) * 4096));
                    }
                    CHK(dalloc(&h->tb_bound_d, (size_t)(NB + 1) * h->Kp * N));
                    CHK(dalloc(&h->tb_items_d, items.size()));
                    CHK(hipMemcpy(h->tb_items_d, items.data(), items.size() * sizeof(int), hipMemcpyHostToDevice));
                }
            }
        }
        // N <= 32 (round 5, end): the same scheme on the vector kernels -- chain of products per block (kr_tb_products_small), the
        // sequential apply sweep over the block propagators, the apply kernel once more with one workgroup per (block, tile).  A step
        // of the sequential sweep is ~0.3 us of latency whatever N, so what pays is cutting N_T steps to N_T / S + S: S ~ sqrt(N_T).
        // C2: 139 + 15 us per sweep -> see DESIGN.md 3.4.
        if (h->precomp && NP <= 32 && NT >= opt_int(h, "TBLOCK_MIN_NT", 128)) {
            int S = opt_int(h, "TBLOCK", -1);
            // (measured, C2 per sweep [us] at S = 8 / 12 / 16 / 22 / 32 / 45 / 64: 52 / 49 / 48 / 51 / 57 / 65 / 79 -- the chain of S - 1 dependent
            //  products per block weighs more than a step of the sweeps; N = 32, N_T = 1000, K = 64: 235 / 244 / 303 at 8 / 16 / 32, 500 without)
            if (S < 0) { S = (int)std::lround(0.7 * std::sqrt((double)NT)); S = std::max(4, std::min(32, S)); }
            if (S >= 2 && S <= 64 && NT >= 2 * S) {
                const int NB = (NT + S - 1) / S;
                const size_t bbytes = (size_t)(NB + 1) * h->Kp * N * sizeof(double2);
                if (bbytes <= ((size_t)4 << 30)) {
                    h->tb_S = S; h->tb_NB = NB; h->tb_small = true; h->tb_tree = false;
                    CHK(dalloc(&h->tb_P_d, (size_t)G * NB * NP * NP));
                    CHK(dalloc(&h->tb_bound_d, (size_t)(NB + 1) * h->Kp * N));
                }
            }
        }
    }
    CHK(dalloc(&h->tau_d, h->Kp));
    CHK(dalloc(&h->tausum_d, 1));
    CHK(hipMemset(h->psi0_d, 0, slab * sizeof(double2)));
    CHK(hipMemset(h->tgt_d, 0, slab * sizeof(double2)));
    CHK(hipMemset(h->psiT_d, 0, slab * sizeof(double2)));
    CHK(hipMemset(h->chiT_d, 0, slab * sizeof(double2)));
    CHK(hipMemset(h->gint_d, 0, std::max(L, KR_LMAX_CAP) * sizeof(double)));
    h->mail_words = KR_MAIL_XTAB_OFF + KR_MAIL_XTAB_WORDS;  // 2 buffers x 8 replicas (one per XCD) x 64 KB (kr_exchange); tail: a poll pass
                                                            // reads whole 64-slot chunks (kr_poll_pass); then the XCC-id table of the
                                                            // one-XCD placement (kr_sweep_update)
    // (the generic update kernels do not use this mailbox: any number of tiles -- N = 8, K = 8193 used to fail here)
    if (!h->gen_update && (size_t)h->P * 4 > KR_MAIL_RSTRIDE) { fail(h, KR_ERR_UNSUPPORTED, "%d workgroups do not fit one mailbox replica", h->P); return bail(KR_ERR_UNSUPPORTED); }
    h->pulse_check = opt_int(h, "PULSE_CHECK", 1) != 0 && !h->gen_update && h->P > 1;
    if (!h->gen_update && !h->rowsplit) CHK(dalloc(&h->dulog_d, (size_t)h->P * NT * h->LMAXv));   // (eps_new is formed from workgroup 0's log whether the logs are compared or not)
    else if (!h->gen_update && h->pulse_check && !h->rs_pair) CHK(dalloc(&h->dulog_d, (size_t)h->P * NT * KR_LMAX_CAP));   // (row-split kernels: comparison only, stride KR_LMAX_CAP; not the pair mode: kr_device_rowsplit2.h)
    // The mailbox of the per-step exchange, self-tested (kr_mail_selftest: the exchange's own store / poll pattern on both buffers and
    // all replicas, ~50 us): a block on which stores get lost or stay invisible to some XCDs -- seen for blocks that reuse the address
    // range of a freed uncached block, which this library no longer frees but another library of the process may -- is set aside and
    // another one is asked for (three attempts; KROTOV_HIP_MAIL_SELFTEST=0 skips the test).
    // (every memset of a mailbox under test runs on the handle's OWN stream and is waited for: the stream is non-blocking, a memset on
    //  the null stream is not ordered with the self-test kernel and wiped words it had just published -- 18 false alarms in 243 tests)
    for (int attempt = 0;; ++attempt) {
        CHK(alloc_mailbox(h, &h->mail_d, h->mail_words, h->P));
        CHK(hipMemsetAsync(h->mail_d, 0, h->mail_words * sizeof(unsigned long long), h->own_stream)); CHK(hipStreamSynchronize(h->own_stream));
        if (h->gen_update || opt_int(h, "MAIL_SELFTEST", 1) == 0) break;
        const int mf = opt_int(h, "MAIL_FAULT", 0);   // (tests: 1 = the first block fails, 2 = every block fails)
        unsigned st[KR_STATUS_WORDS] = {};
        for (int buf = 0; buf < 2 && st[0] == 0; ++buf) {
            rc = mailbox_selftest(h, h->mail_d + (size_t)buf * KR_MAIL_REPLICAS * KR_MAIL_RSTRIDE, h->P, KR_MAIL_REPLICAS, (size_t)KR_MAIL_RSTRIDE, h->LMAXv > 0 ? h->LMAXv : 2,
                                  std::max(1, std::min(h->cfg.L, 4)), (mf == 2 || (mf == 1 && attempt == 0)) ? 1 : 0, st);
            if (rc != KR_OK) return bail(rc);
        }
        if (st[0] == 0) { CHK(hipMemsetAsync(h->mail_d, 0, h->mail_words * sizeof(unsigned long long), h->own_stream)); CHK(hipStreamSynchronize(h->own_stream)); break; }
        quarantine_block(h->mail_d);   // never used, never freed
        h->mail_d = nullptr;
        h->quarantined++;
        if (attempt == 2) {
            fail(h, KR_ERR_INTERNAL, "kr_create: three freshly allocated mailboxes failed their self-test (workgroup %u, round %u, replica %u of the last one did not see every "
                 "published word within 100 ms): device memory of this process loses stores or hides them from some XCDs; see INTEGRATION.md, \"freed uncached blocks\"",
                 st[2] & 0xffffu, (st[2] >> 16) & 0xffu, st[2] >> 24);
            return bail(KR_ERR_INTERNAL);
        }
    }
    CHK(dalloc(&h->debug_d, 8 + 8 * 1024));   // 8 phase stamps + per-workgroup arrival stamps of diagnostic builds (KR_ARRIVALS)
    CHK(hipMemset(h->debug_d, 0, (8 + 8 * 1024) * sizeof(unsigned long long)));
    h->debug_on = opt_int(h, "STAMPS", 0) != 0;
    if (cfg->world > 1) {
        // inter-GPU mailbox: written by peers over xGMI, polled by this GPU's kernels -> uncached/fine-grained
        const size_t xwords = (size_t)2 * 8 * 2 * KR_LMAX_CAP + 2 * 8 * 4;   // update-sweep slots + tau slots (KR_XMAIL_TAU_OFF)
        // preference: uncached, then fine-grained, then plain device memory (peers always use system-scope atomics)
        h->xmail_bytes = xwords * sizeof(unsigned long long);
        for (int kind : {1, 2, 0}) {
            h->xmail_kind = kind;
            if (special_alloc(h, (void **)&h->xmail_d, h->xmail_bytes, kind, &h->xmail_gen) == hipSuccess) break;
            (void)hipGetLastError();
            h->xmail_d = nullptr;
        }
        if (!h->xmail_d) { fail(h, KR_ERR_HIP, "no device memory for the mailbox between GPUs"); return bail(KR_ERR_HIP); }
        CHK(hipMemsetAsync(h->xmail_d, 0, xwords * sizeof(unsigned long long), h->own_stream)); CHK(hipStreamSynchronize(h->own_stream));
        if (opt_int(h, "MAIL_SELFTEST", 1) != 0) {   // the mailbox between GPUs, as this GPU's workgroups see it
            unsigned st[KR_STATUS_WORDS] = {};
            rc = mailbox_selftest(h, h->xmail_d, 8, 1, 0, (int)(xwords / 8), 4, 0, st);
            if (rc != KR_OK) return bail(rc);
            if (st[0] != 0) {
                quarantine_block(h->xmail_d); h->xmail_d = nullptr;
                fail(h, KR_ERR_INTERNAL, "kr_create: the mailbox between GPUs failed its self-test (workgroup %u did not see every published word within 100 ms); the block is set aside -- call kr_create again", st[2] & 0xffffu);
                return bail(KR_ERR_INTERNAL);
            }
            CHK(hipMemsetAsync(h->xmail_d, 0, xwords * sizeof(unsigned long long), h->own_stream)); CHK(hipStreamSynchronize(h->own_stream));
        }
        h->peer_xmail[cfg->rank] = h->xmail_d;
    }
    CHK(hipDeviceSynchronize());
#undef CHK
    rc = calibrate_exchange(h);
    if (rc != KR_OK) return bail(rc);
    {   // kr_plan_description(): which kernels carry this problem, and why if not the persistent ones
        char buf[2100];
        const char *upd = h->gen_update ? "generic per-step kernels (kr_gen_partial/reduce/form/apply)"
                        : h->rowsplit ? (h->rs_pair ? "kr_rowsplit_pair_sweep (two interleaved trajectories per group)"
                                         : h->rs_turns > 0 ? "kr_rowsplit_turns_update (the groups of workgroups carry several trajectories in turn)" : "kr_rowsplit_sweep")
                        : h->mfma_sweep ? "kr_mf_update (matrix cores)"
                        : h->NP > 64 ? "kr_stream_update" : "kr_sweep_update";
        const char *known = h->gen_known ? "kr_gen_known" : h->rowsplit ? (h->rs_batch > 0 ? "kr_rowsplit_sweep in batches of launches" : "kr_rowsplit_sweep") : h->precomp ? "kr_expm + kr_sweep_apply" : "kr_sweep_known";
        char known_tb[160];
        if (h->tb_S > 0) {
            if (h->tb_small)
                snprintf(known_tb, sizeof known_tb, "time blocks (kr_expm_batch + kr_tb_products_small + kr_sweep_apply over %d block propagators + kr_sweep_apply per block, block length %d)",
                         h->tb_NB, h->tb_S);
            else
            snprintf(known_tb, sizeof known_tb, "time blocks on the matrix cores (kr_expm_mfma64 + %s + %s over %d block propagators + kr_tb_apply, block length %d)",
                     h->tb_tree ? "kr_tb_pair tree" : "kr_tb_products", h->mfma_sweep ? "kr_mf_apply_sweep" : "kr_sweep_apply", h->tb_NB, h->tb_S);
            known = known_tb;
        }
        static const char *const exn[] = {"auto", "mailbox", "rccl", "callback"};
        snprintf(buf, sizeof buf, "N=%d (padded %d) NT=%d K=%d (padded %d) L=%d G=%d rank %d/%d on device %d: update sweep = %s, %d workgroups x %d "
                 "trajectories, sets per workgroup %d, one-XCD placement %d, R=%d, HERM=%d, economized=%d, poll_delay=%d; known-pulse sweeps = %s; exchange = %s; spin bound %.0f ms; "
                 "integrity checks: operator-copy checksums %s, group coherence words %s, per-step totals of all workgroups compared %s, mailbox self-test %s%s%s%s%s",
                 h->cfg.N, h->NP, h->cfg.NT, h->cfg.K, h->Kp, h->cfg.L, h->G, h->cfg.rank, h->cfg.world, h->device, upd, h->P, h->KTILE, h->rs_turns > 0 ? h->rs_turns : h->sets,
                 h->wg_stride > 1 ? 1 : 0, h->rowsplit ? h->R : 1, h->herm ? 1 : 0, (h->econ || h->econ_gen) ? 1 : 0, h->poll_delay, known, exn[h->exchange & 3], (double)h->spin_ticks / 1e5,
                 h->pack_words_d ? "on" : "off", (h->rowsplit && h->R > 1) ? "on" : "n/a", (h->pulse_check && h->dulog_d) ? "on" : "n/a",
                 (h->gen_update || opt_int(h, "MAIL_SELFTEST", 1) == 0) ? "off" : (h->quarantined ? "passed after setting blocks aside" : "passed"),
                 opt_int(h, "UC_POOL", 1) == 0 ? "; WARNING: UC_POOL=0 (diagnostics only): uncached mailboxes are freed again, and blocks that reuse their address range lose stores on this platform" : "",
                 !h->cheby_served ? ""
                 : h->gen_update ? "; prop_method = CHEBY is served by this plan's own step polynomials at the same tolerance (stepped update sweep: economized Taylor); "
                                   "KROTOV_HIP_CHEBY_RECURRENCE=1 selects the three-term recurrence on the generic kernels"
                 : "; prop_method = CHEBY is served by the persistent kernels: the update sweep evaluates the Chebyshev truncation of the step in powers of the "
                   "generator (spectral bound per step; a step whose updated pulse leaves it takes the Taylor polynomial), the known-pulse sweeps take "
                   "their usual plan at the same tolerance; KROTOV_HIP_CHEBY_RECURRENCE=1 selects the three-term recurrence on the generic kernels",
                 h->plan_note.empty() ? "" : "; note: ", h->plan_note.c_str());
        h->plan_text = buf;
    }
    live = nullptr;
    *out = h;
    return KR_OK;
}
// An exception in the middle of kr_create (std::bad_alloc in a staging vector, ...) must not leak the half-built handle and
// its device memory: free what exists, then let kr_guard turn the exception into a status.
static int one_create(const kr_config *cfg, kr_handle **out) {
    kr_handle *live = nullptr;
    try {
        return one_create_impl(cfg, out, live);
    } catch (...) {
        if (live) { (void)hipSetDevice(live->device); free_all(live); delete live; }
        if (out) *out = nullptr;
        throw;
    }
}

static int one_set_stream(kr_handle *h, void *hip_stream) {
    if (!h) return KR_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mtx);
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    return KR_OK;
}

static int one_enable_timing(kr_handle *h, int on) {
    if (!h) return KR_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mtx);
    h->timing = on != 0;
    for (auto &n : h->ev_n) n = 0;  // (re)start the accumulation window
    return KR_OK;
}

static int ev_mark(kr_handle *h, int kind, bool end) {
    if (!h->timing) return KR_OK;
    if (h->ev_n[kind] >= 4096) return KR_OK;  // window full: stop recording, keep what we have
    const size_t idx = 2 * h->ev_n[kind] + (end ? 1 : 0);
    while (h->ev[kind].size() <= idx) {
        hipEvent_t e;
        HIPCHK(h, hipEventCreate(&e));
        h->ev[kind].push_back(e);
    }
    HIPCHK(h, hipEventRecord(h->ev[kind][idx], h->stream));
    if (end) h->ev_n[kind]++;
    return KR_OK;
}

static int one_get_timings(kr_handle *h, double *t) {
    if (!h || !t) return KR_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mtx);
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < 3; ++i) {
        double sum = 0;
        for (size_t q = 0; q < h->ev_n[i]; ++q) {
            float ms = 0;
            HIPCHK(h, hipEventElapsedTime(&ms, h->ev[i][2 * q], h->ev[i][2 * q + 1]));
            sum += ms;
        }
        t[2 * i] = sum;
        t[2 * i + 1] = (double)h->ev_n[i];
    }
    return KR_OK;
}

static int one_get_info(kr_handle *h, int32_t *info) {
    if (!h || !info) return KR_ERR_INVALID;
    memset(info, 0, 16 * sizeof(int32_t));
    info[0] = h->NP; info[1] = h->P; info[2] = h->KTILE; info[3] = h->threads; info[4] = h->KT;
    info[5] = h->LMAXv; info[6] = h->Kp; info[7] = h->store_fw ? 1 : 0; info[8] = h->JQ; info[9] = h->precomp ? 1 : 0; info[10] = h->rowsplit ? h->R : 0; info[11] = h->poll_delay; info[12] = (h->herm && h->NP <= 64) ? 1 : 0; info[13] = h->mfma_sweep ? 1 : 0; info[14] = h->gen_known ? 1 : 0; info[15] = (h->gen_update ? 1 : 0) | (h->exchange << 4) | ((h->econ || h->econ_gen) ? 256 : 0);
    if (h->mfma_sweep) info[3] = 256;
    // Taylor plan at zero pulse amplitude (drift only) for the mean step, for reporting
    return KR_OK;
}

static int one_debug_stamps(kr_handle *h, unsigned long long *out) {
    if (!h || !out) return KR_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mtx);
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(out, h->debug_d, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return KR_OK;
}

// Diagnostic builds (-DKR_ARRIVALS): per workgroup {publish, end of window work, completion, poll passes} x 2 exchanges, realtime ticks.
static int one_debug_arrivals(kr_handle *h, unsigned long long *out, int words) {
    if (!h || !out || words < 0 || words > 8 * 1024) return KR_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mtx);
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(out, h->debug_d + 8, (size_t)words * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return KR_OK;
}

// copy [K][N] host states into the zero-padded [Kp][N] device buffer (grouped by generator)
static int upload_states(kr_handle *h, double2 *dst, const double *src) {
    const size_t N = h->cfg.N, K = h->cfg.K;
    if (h->identity_perm) {
        HIPCHK(h, hipMemcpyAsync(dst, src, K * N * sizeof(double2), hipMemcpyHostToDevice, h->stream));
        return KR_OK;
    }
    h->stage.assign((size_t)h->Kp * N * 2, 0.0);
    for (size_t k = 0; k < K; ++k) memcpy(&h->stage[(size_t)h->perm[k] * N * 2], src + k * N * 2, N * sizeof(double2));
    HIPCHK(h, hipMemcpyAsync(dst, h->stage.data(), (size_t)h->Kp * N * sizeof(double2), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));  // stage is reused
    return KR_OK;
}

static int one_set_initial_states(kr_handle *h, const double *psi0) {
    if (!h || !psi0) return h ? fail(h, KR_ERR_INVALID, "kr_set_initial_states: null") : KR_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mtx);
    HIPCHK(h, hipSetDevice(h->device));
    int rc = upload_states(h, h->psi0_d, psi0);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_psi0 = true;
    return KR_OK;
}

static int one_set_target_states(kr_handle *h, const double *tgt) {
    if (!h || !tgt) return h ? fail(h, KR_ERR_INVALID, "kr_set_target_states: null") : KR_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mtx);
    HIPCHK(h, hipSetDevice(h->device));
    int rc = upload_states(h, h->tgt_d, tgt);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_tgt = true;
    return KR_OK;
}

static void base_args(kr_handle *h, KrSweepArgs &a) {
    memset(&a, 0, sizeof a);
    a.dt = h->dt_d;
    a.N = h->cfg.N; a.NT = h->cfg.NT; a.Kp = h->Kp; a.L = h->cfg.L;
    a.P = h->P; a.rank = h->cfg.rank; a.world = h->cfg.world;
    a.tol = h->tol;
    a.spin_ticks = h->spin_ticks;
    memcpy(a.thr, h->thr, sizeof a.thr);
    a.use_mfma = h->use_mfma ? (1 | (opt_int(h, "EXPM_PS", 1) != 0 ? 2 : 0)) : 0;   // bit 1: Paterson-Stockmeyer form of the matrix-core propagator build
    a.gmail = h->gmail_d;
    a.xcd_group = h->xcd_group ? 1 : 0;
    a.rs_sets = h->rs_sets;
    a.rs_fault = opt_int(h, "RS_FAULT", 0);
    if (opt_int(h, "PULSE_FAULT", 0) != 0) a.rs_fault = h->rowsplit ? 1001 : 2;   // (tests: workgroup 1 logs a wrong total at step 3 / step 1 of a row-split sweep)
    a.rs_scratch = h->rs_scratch_d;
    a.pack_stride = h->pack_stride;
    a.Hdense = h->Hdense_fw;
    a.dense_stride = h->dense_stride;
    a.gen_div = 1;
    a.wg_stride = 1;
    a.wg_gen = (h->G > 1) ? h->wg_gen_d : nullptr;
    a.nrm_dev = h->nrm_d;
    a.econ = nullptr;
    a.present_dev = h->present_d;
    a.status = h->status_d;
    a.debug = h->debug_on ? h->debug_d : nullptr;
    a.mail = h->mail_d;
    a.poll_delay = h->poll_delay;
    if (h->rowsplit && h->poll_delay < 0) a.poll_delay = opt_int(h, "RS_POLL_DELAY", 12);   // row-split update sweep: kr_exchange<.., SCHED = 1> (KR_RS_SCHED)
    a.xpoll_delay = opt_int(h, "XPOLL_DELAY", 0);
    // row-split all-gather: polling starts 9 x 128 cycles (0.48 us) after the block was handed to the publishing wave;
    // measured at C5 (forward sweep, us per step): 0 -> 32.4, 4 -> 31.5, 8 -> 30.6, 10 -> 30.4, 12 -> 31.0, 16 -> 32.8, 20 -> 34.7
    a.gather_delay = opt_int(h, "GATHER_DELAY", 9);
    a.gather_delay_local = opt_int(h, "GATHER_DELAY_LOCAL", 2);  // C5 forward sweep: 0 -> 26.4, 4 -> 25.8, 6 -> 25.5, 8 -> 26.1 us per step
    a.xmail = h->xmail_d;
    for (int r = 0; r < 8; ++r) a.peer_xmail[r] = h->peer_xmail[r];
}

// vector-mailbox tags of the row-split kernels: (launch id mod 256) << 24 | gather counter; the mailbox is cleared
// whenever the launch id wraps, and a launch may use at most 2^24 gathers (N_T * rounds per step)
static int zero_async(kr_handle *h, void *p, size_t words) {
    if (opt_int(h, "ZERO_KERNEL", 0) == 0) { HIPCHK(h, hipMemsetAsync(p, 0, words * sizeof(unsigned long long), h->stream)); return KR_OK; }
    const int blocks = (int)std::min<size_t>(512, (words + 255) / 256);
    hipLaunchKernelGGL(kr_zero_words, dim3(std::max(blocks, 1)), dim3(256), 0, h->stream, (unsigned long long *)p, words);
    HIPCHK(h, hipGetLastError());
    if (opt_int(h, "ZERO_SYNC", 0) != 0) HIPCHK(h, hipStreamSynchronize(h->stream));
    return KR_OK;
}
static int rowsplit_prelaunch(kr_handle *h, KrSweepArgs &a) {
    if ((long long)h->cfg.NT * (KR_MAX_ORDER + 2) * 4 >= (1ll << 24))
        return fail(h, KR_ERR_UNSUPPORTED, "row-split kernels: N_T = %d too large for the mailbox tag space (set KROTOV_HIP_ROWSPLIT=0)", h->cfg.NT);
    h->launch_id = (h->launch_id + 1) & 0xffu;
    if (h->launch_id == 0) h->launch_id = 1;
    // the all-gather validates its elements with one bit per double whose phase starts at 1 in every launch: start from zeros
    { const int zrc = zero_async(h, h->gmail_d, h->gmail_words); if (zrc) return zrc; }
    a.epoch_base2 = h->launch_id << 24;
    return KR_OK;   // (the status words are sticky: only check_status clears them)
}

// single-mode row-split sweeps end with a coherence word per workgroup (kr_rowsplit_sweep): compare them group by group
// (tiles = groups of R workgroups in the launch; each carries KT trajectories -- 2 in pair mode --, `sets` of them in turn, whose
//  mailboxes lie in front of the words)
static int rowsplit_postlaunch(kr_handle *h, int tiles, int sets = 1) {
    if (h->R <= 1) return KR_OK;
    hipLaunchKernelGGL(kr_rowsplit_check, dim3((tiles + 255) / 256), dim3(256), 0, h->stream,
                       h->gmail_d + (size_t)tiles * sets * h->KT * 2 * h->NP * 4, tiles, h->R, h->status_d);
    HIPCHK(h, hipGetLastError());
    return KR_OK;
}

// ---- polynomial step propagators (Chebyshev, Newton): per-step coefficient tables ------------------------
// Lambda bounds the spectral radius of H(eps) for every eps in the control range [-emax_l, emax_l]; a step with interval
// dt_n then needs a polynomial good on [-th, th], th = |dt_n| Lambda / s (s sub-steps).  Tables are cached per th (a uniform
// grid has one).  Row layout: kr_gen_propagate (kr_device_generic.h).
static int cheby_row(double th, double tol, double *row) {
    const int kmax = KR_MAX_ORDER + 1;
    double J[KR_MAX_ORDER + 2];
    for (int k = 0; k <= kmax; ++k) J[k] = std::cyl_bessel_j((double)k, th);
    int d = kmax;
    double tail = 0.0;
    for (int k = kmax; k >= 1; --k) {   // smallest d whose dropped terms sum below tol
        tail += 2.0 * std::fabs(J[k]);
        if (tail > tol) { d = k; break; }
        d = k - 1;
    }
    if (d < 1) d = 1;
    for (int k = 0; k <= d; ++k) row[3 + k] = J[k];
    return d;
}
static int newton_row(double th, double tol, double *row) {
    typedef std::complex<long double> lc;
    const int dmax = KR_MAX_ORDER;
    // Leja points of [-th, th] from a fine candidate set (greedy: maximise the product of distances)
    const int NC = 4001;
    std::vector<long double> cand(NC), logp(NC, 0.0L);
    for (int i = 0; i < NC; ++i) cand[i] = -(long double)th * std::cos((long double)M_PIl * i / (NC - 1));
    std::vector<long double> z;
    z.push_back((long double)th);
    for (int k = 1; k <= dmax; ++k) {
        int best = 0;
        long double bv = -1e300L;
        for (int i = 0; i < NC; ++i) {
            const long double dist = std::fabs(cand[i] - z.back());
            logp[i] += (dist > 0) ? std::log(dist) : -1e300L;
            if (logp[i] > bv) { bv = logp[i]; best = i; }
        }
        z.push_back(cand[best]);
    }
    // divided differences of f(z) = exp(-i z)
    std::vector<lc> dd(dmax + 1);
    for (int k = 0; k <= dmax; ++k) dd[k] = std::exp(lc(0, -1) * z[k]);
    for (int j = 1; j <= dmax; ++j)
        for (int k = dmax; k >= j; --k) dd[k] = (dd[k] - dd[k - 1]) / (z[k] - z[k - j]);
    // smallest degree whose interpolant is within tol of f on the circle |z| = th (the field of values lies inside it)
    int d = dmax;
    for (int cand_d = 1; cand_d <= dmax; ++cand_d) {
        long double err = 0;
        for (int q = 0; q < 48; ++q) {
            const lc zz = std::polar((long double)th, 2.0L * M_PIl * q / 48);
            lc acc = dd[cand_d];
            for (int k = cand_d - 1; k >= 0; --k) acc = acc * (zz - z[k]) + dd[k];
            err = std::max(err, std::abs(acc - std::exp(lc(0, -1) * zz)));
        }
        if (err <= (long double)tol * 4) { d = cand_d; break; }
    }
    for (int k = 0; k < d; ++k) row[3 + k] = (double)z[k];
    for (int k = 0; k <= d; ++k) { row[3 + d + 2 * k] = (double)dd[k].real(); row[3 + d + 2 * k + 1] = (double)dd[k].imag(); }
    return d;
}
// eps_dev: the pulses of this sweep ([L][NT], device); widen: factor on the control range (1 for known pulses, 5 for the update
// sweep, whose pulses are only known as it goes: transform_control_ranges, optimize.jl:238-244)
static int build_poly_tables(kr_handle *h, const double *eps_dev, double widen) {
    const int L = h->cfg.L, NT = h->cfg.NT;
    h->eps_stage.resize((size_t)L * NT);
    HIPCHK(h, hipMemcpyAsync(h->eps_stage.data(), eps_dev, (size_t)L * NT * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    double lam = h->lam0;
    std::vector<double> erange(L, 0.0);
    for (int l = 0; l < L; ++l) {
        double em = 0.0;
        for (int n = 0; n < NT; ++n) em = std::max(em, std::fabs(h->eps_stage[(size_t)l * NT + n]));
        erange[l] = widen * em;
        lam += erange[l] * h->lam_l[l];
    }
    const double cap = (h->method == KR_PROP_CHEBY) ? 6.0 : 1.5;
    const double tol = std::max(h->tol, 1e-17);
    h->ptab_host.assign((size_t)NT * h->ptab_stride, 0.0);
    std::vector<std::pair<double, std::vector<double>>> cache;
    for (int n = 0; n < NT; ++n) {
        const double full = std::fabs(h->dt_host[n]) * lam;
        const int s = std::max(1, (int)std::ceil(full / cap));
        const double th = std::max(full / s, 1e-300);
        double *row = &h->ptab_host[(size_t)n * h->ptab_stride];
        const std::vector<double> *hit = nullptr;
        for (auto &c : cache) if (c.first == th) { hit = &c.second; break; }
        if (!hit) {
            std::vector<double> r(h->ptab_stride, 0.0);
            const int d = (h->method == KR_PROP_CHEBY) ? cheby_row(th, tol, r.data()) : newton_row(th, tol, r.data());
            r[0] = d; r[2] = th;
            cache.emplace_back(th, std::move(r));
            hit = &cache.back().second;
        }
        memcpy(row, hit->data(), h->ptab_stride * sizeof(double));
        row[1] = s;
    }
    HIPCHK(h, hipMemcpyAsync(h->ptab_d, h->ptab_host.data(), h->ptab_host.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->erange_d, erange.data(), (size_t)L * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));   // (erange is a local)
    return KR_OK;
}

// ---- generic (multi-pass) sweeps -------------------------------------------------------------------
static void gen_args(kr_handle *h, KrGenArgs &g) {
    memset(&g, 0, sizeof g);
    g.gen_stride = h->col_stride;
    g.Heps = h->Heps_d;
    g.tile_gen = (h->G > 1) ? h->tile_gen_d : nullptr;
    g.nrm = h->nrm_gen_d;
    g.present = h->present_d;
    g.dt = h->dt_d;
    g.tol = h->tol;
    memcpy(g.thr, h->thr, sizeof g.thr);
    g.N = h->cfg.N; g.NT = h->cfg.NT; g.Kp = h->Kp; g.L = h->cfg.L; g.G = h->G;
    g.KTG = h->KTG; g.RP = h->RP; g.JQ = h->JQg; g.tiles = h->gen_tiles;
    g.poly = h->poly_d; g.poly_deg = h->poly_deg;
    g.econ = nullptr;
    g.method = h->method; g.ptab = h->ptab_d; g.ptab_stride = h->ptab_stride; g.erange = nullptr; g.status = h->status_d;
}

static int launch_known_generic(kr_handle *h, const double *eps_dev, bool backward, const double2 *init, double2 *store, double2 *fin) {
    KrGenArgs g;
    gen_args(h, g);
    g.H = backward ? h->Hcol_bw : h->Hcol_fw;
    g.eps = eps_dev;
    g.init = init; g.store = store; g.fin = fin;
    g.dir = backward ? -1 : +1;
    if (h->method == KR_PROP_CHEBY || h->method == KR_PROP_NEWTON) { const int rc = build_poly_tables(h, eps_dev, 1.0); if (rc) return rc; }
    if (h->econ_gen) {   // economized polynomial of every step (pulses known exactly: the margin covers rounding only)
        hipLaunchKernelGGL(kr_econ_rows_kernel, dim3(h->G * h->cfg.NT), dim3(64), 0, h->stream, eps_dev, h->dt_d, h->nrm_gen_d, (const double *)nullptr, h->G,
                           h->cfg.NT, h->cfg.L, 1 + h->cfg.L, h->tol, 1.0 + 1e-9, h->econ_d);
        HIPCHK(h, hipGetLastError());
        g.econ = h->econ_d;
    }
    HIPCHK(h, kr_launch_generic(0, h->KTG, h->gen_tiles, h->gen_lds, h->stream, &g));
    return KR_OK;
}

// Stepped update sweep (kr_device_generic.h): partial sums -> reduce -> exchange -> pulse update + H(eps_new) -> forward step,
// once per time step.  Exchange: none (one rank), ncclAllReduce on the handle's stream, or the host callback.
static int launch_update_stepped(kr_handle *h, const double *eps_guess_dev, double *eps_new_dev) {
    const kr_config &c = h->cfg;
    const int N = c.N, NT = c.NT, L = c.L;
    const size_t slab = (size_t)h->Kp * N;
    KrGenArgs g;
    gen_args(h, g);
    g.H = h->Hcol_fw;
    g.eps = eps_guess_dev; g.alpha = h->S_d; g.eps_new = eps_new_dev; g.g_a_int = h->gint_d;
    g.init = h->psi0_d; g.state = h->psiT_d; g.store = h->store_fw ? h->Phi_d : nullptr; g.fin = nullptr;
    g.X = h->X_d; g.partial = h->partial_d; g.du_local = h->du_local_d;
    g.dir = +1;
    const bool multi = c.world > 1;
    if (multi && h->exchange == KR_EXCHANGE_RCCL && !h->rccl_comm)
        return fail(h, KR_ERR_STATE, "exchange = RCCL with world = %d but kr_rccl_init has not been called", c.world);
    if (multi && h->exchange == KR_EXCHANGE_CALLBACK && !h->cb)
        return fail(h, KR_ERR_STATE, "exchange = CALLBACK with world = %d but kr_set_exchange_callback has not been called", c.world);
    const bool use_rccl = h->exchange == KR_EXCHANGE_RCCL && h->rccl_comm;
    const bool use_cb = h->exchange == KR_EXCHANGE_CALLBACK && h->cb;
    g.du_total = (use_rccl || use_cb) ? h->du_total_d : h->du_local_d;
    if (h->method == KR_PROP_CHEBY || h->method == KR_PROP_NEWTON) {
        const int rc = build_poly_tables(h, eps_guess_dev, 5.0);
        if (rc) return rc;
        g.erange = h->erange_d;
    }
    if (h->econ_gen) {   // rows from the guess pulses, 1 % margin (a step whose updated pulse leaves it takes Taylor's plan)
        hipLaunchKernelGGL(kr_econ_rows_kernel, dim3(h->G * NT), dim3(64), 0, h->stream, eps_guess_dev, h->dt_d, h->nrm_gen_d, (const double *)nullptr, h->G, NT, L,
                           1 + L, h->tol, 1.01, h->econ_d);
        HIPCHK(h, hipGetLastError());
        g.econ = h->econ_d;
    }
    HIPCHK(h, hipMemcpyAsync(h->psiT_d, h->psi0_d, slab * sizeof(double2), hipMemcpyDeviceToDevice, h->stream));
    if (h->store_fw) HIPCHK(h, hipMemcpyAsync(h->Phi_d, h->psi0_d, slab * sizeof(double2), hipMemcpyDeviceToDevice, h->stream));
    { const int zrc = zero_async(h, h->gint_d, (size_t)L); if (zrc) return zrc; }
    const int form_grid = (int)(((size_t)h->G * N * N + 255) / 256);
    // One rank (the sum stays on the device), L <= 32: TWO launches per time step instead of four -- the pulse update sums the
    // tiles' partials itself, the forward step adds the partial sums of the next step (kr_device_generic.h; round 3).  The
    // stepped sweep is a chain of dependent launches (a HIP graph of it replays no faster: DESIGN.md section 7).
    if (!use_rccl && !use_cb && L <= 32 && opt_int(h, "GENERIC_FUSED", 1) != 0) {
        g.n = 0;       // (step 0's partial sums: the plain kernel)
        HIPCHK(h, kr_launch_generic(1, h->KTG, h->gen_tiles, h->gen_lds, h->stream, &g));
        g.fused = 1;
        for (int n = 0; n < NT; ++n) {
            g.n = n;
            HIPCHK(h, kr_launch_generic(3, h->KTG, form_grid, 0, h->stream, &g));
            HIPCHK(h, kr_launch_generic(4, h->KTG, h->gen_tiles, h->gen_lds, h->stream, &g));
        }
        return KR_OK;
    }
    for (int n = 0; n < NT; ++n) {
        g.n = n;
        HIPCHK(h, kr_launch_generic(1, h->KTG, h->gen_tiles, h->gen_lds, h->stream, &g));
        HIPCHK(h, kr_launch_generic(2, h->KTG, 1, 0, h->stream, &g));
        if (use_rccl) {
            const ncclResult_t r = kr_rccl()->AllReduce(h->du_local_d, h->du_total_d, (size_t)L, ncclDouble, ncclSum, h->rccl_comm, h->stream);
            if (r != ncclSuccess) { h->rccl_bad = true; return fail(h, KR_ERR_HIP, "ncclAllReduce failed at step %d: %s", n, kr_rccl()->GetErrorString(r)); }
        } else if (use_cb) {
            HIPCHK(h, hipMemcpyAsync(h->du_host, h->du_local_d, (size_t)L * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            const int crc = h->cb(h->cb_ctx, h->du_host, L);
            if (crc != 0) return fail(h, KR_ERR_HIP, "exchange callback returned %d at step %d", crc, n);
            HIPCHK(h, hipMemcpyAsync(h->du_total_d, h->du_host, (size_t)L * sizeof(double), hipMemcpyHostToDevice, h->stream));
        }
        HIPCHK(h, kr_launch_generic(3, h->KTG, form_grid, 0, h->stream, &g));
        HIPCHK(h, kr_launch_generic(4, h->KTG, h->gen_tiles, h->gen_lds, h->stream, &g));
    }
    return KR_OK;
}

// one small launch in front of a sweep: the pack it reads still is what kr_create uploaded (kr_pack_check)
static int pack_precheck(kr_handle *h, int which) {   // which: 0 forward pack, 1 adjoint pack, 2 dense (matrix cores), 3 / 4 column-major fw / adjoint
    const double2 *bufs[5] = {h->Hpack_fw, h->Hpack_bw, h->Hdense_fw, h->Hcol_fw, h->Hcol_bw};
    const size_t n = h->const_elems[which];
    if (!h->pack_words_d || n == 0 || !bufs[which]) return KR_OK;
    const int blocks = (int)std::min<size_t>(256, (n + 2047) / 2048);
    hipLaunchKernelGGL(kr_pack_check, dim3(blocks), dim3(256), 0, h->stream, bufs[which], n, h->pack_sum[which], h->pack_words_d, which, h->status_d);
    HIPCHK(h, hipGetLastError());
    return KR_OK;
}

static int launch_known(kr_handle *h, int which_ev, const double *eps_dev, bool backward, const double2 *init, double2 *store, double2 *fin) {
    KrSweepArgs a;
    base_args(h, a);
    { const int prc = pack_precheck(h, h->gen_known ? (backward ? 4 : 3) : (backward ? 1 : 0)); if (prc) return prc; }
    a.Hpack = backward ? h->Hpack_bw : h->Hpack_fw;
    if (h->poly_deg > 0 && !h->gen_known) {   // non-linear controls: the persistent kernels combine the matrices with f_l(eps_l)
        const size_t n = (size_t)h->cfg.L * h->cfg.NT;
        hipLaunchKernelGGL(kr_poly_eval, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, eps_dev, (const double *)h->poly_d, h->poly_deg, h->cfg.L, h->cfg.NT,
                           h->amp_d, (double *)nullptr);
        HIPCHK(h, hipGetLastError());
        eps_dev = h->amp_d;
    }
    a.eps = eps_dev;
    a.init = init; a.store = store; a.fin = fin;
    a.dir = backward ? -1 : +1;
    a.U = h->U_d;
    int rc = ev_mark(h, which_ev, false);
    if (rc) return rc;
    if (h->gen_known) {
        rc = launch_known_generic(h, eps_dev, backward, init, store, fin);
        if (rc) return rc;
    } else if (h->rowsplit) {
        rc = rowsplit_prelaunch(h, a);
        if (rc) return rc;
        if (h->econ) {   // economized polynomial of every step (pulses known exactly: the margin covers rounding only)
            hipLaunchKernelGGL(kr_econ_rows_kernel, dim3(h->G * h->cfg.NT), dim3(64), 0, h->stream, eps_dev, h->dt_d, h->nrm_d, h->skw_d, h->G, h->cfg.NT,
                               std::min(h->cfg.L, KR_LMAX_CAP), 1 + KR_LMAX_CAP, h->tol, 1.0 + 1e-9, h->econ_d);
            HIPCHK(h, hipGetLastError());
            a.econ = h->econ_d;
        }
        if (h->rs_batch > 0) {   // batches of trajectories, one launch each (known-pulse sweeps: the trajectories are independent)
            const int N = h->cfg.N, K = (h->rs_turns > 0) ? h->Kp : h->cfg.K;   // (turns: the zero trajectories of the padding take part)
            for (int b0 = 0; b0 < K; b0 += h->rs_batch) {
                const int nb = std::min(h->rs_batch, K - b0);
                if (b0 > 0) { rc = rowsplit_prelaunch(h, a); if (rc) return rc; }
                a.init = init + (size_t)b0 * N;
                a.store = store ? store + (size_t)b0 * N : nullptr;
                a.fin = fin ? fin + (size_t)b0 * N : nullptr;
                a.wg_gen = (h->G > 1) ? h->wg_gen_d + b0 : nullptr;
                a.P = nb * h->R;
                a.xcd_group = (h->xcd_group && nb % 8 == 0) ? 1 : 0;
                HIPCHK(h, h->launch_rs(h->LMAXv, h->NR, 0, nb * h->R, h->stream, &a));
                rc = rowsplit_postlaunch(h, nb);
                if (rc) return rc;
            }
        } else {
            HIPCHK(h, h->launch_rs(h->LMAXv, h->NR, h->rs_pair ? 2 : 0, h->P, h->stream, &a));
            rc = rowsplit_postlaunch(h, h->P / h->R, h->rs_sets);
            if (rc) return rc;
        }
    } else if (h->precomp) {
        // all N_T step propagators in parallel, then one matvec round per step.  N = 64: both on the matrix cores
        // (kr_expm_mfma64 writes the propagators as MFMA A operands, kr_mf_apply_sweep applies them, two trajectories per
        // 4-column tile); KROTOV_HIP_MFMA_APPLY=0 keeps the vector apply kernel.
        // Measured (us per step incl. the build): K = 1024: 2.51 -> 2.22, K = 512: 1.56 -> 1.44, but K = 256 (one trajectory
        // per workgroup in the vector kernel): 1.14 -> 1.26 -- both forms are bound by streaming the 64 KB of U_n into every
        // CU each step, so the matrix cores only pay where the vector kernel is VALU-bound (several trajectories per team).
        const int mfa = opt_int(h, "MFMA_APPLY", -1);
        const bool mf_apply = h->NP == 64 && h->use_mfma && (mfa == 1 || (mfa == -1 && h->KT >= 2)) && (h->Kp % 2 == 0) &&
                              (h->G == 1 || h->KT % 2 == 0);
        const bool tb = h->tb_S > 0;   // blocks of time (kr_device_tblock.h)
        a.u_layout = (tb && !h->tb_small) ? 2 : (mf_apply ? 1 : 0);
        if (h->econ && h->NP == 64) {   // economized polynomial of every step (the pulses are known exactly: margin = rounding only;
                                        // smaller N: building the rows costs what the saved product gains -- C2 +6 us per sweep)
            const int rows = h->G * h->cfg.NT;
            hipLaunchKernelGGL(kr_econ_rows_kernel, dim3(rows), dim3(64), 0, h->stream, eps_dev, h->dt_d, h->nrm_d, h->skw_d, h->G, h->cfg.NT, std::min(h->cfg.L, KR_LMAX_CAP), 1 + KR_LMAX_CAP,
                               h->tol, 1.0 + 1e-9, h->econ_d);
            HIPCHK(h, hipGetLastError());
            a.econ = h->econ_d;
        }
        HIPCHK(h, h->launch(h->LMAXv, std::min(h->KT, 4), 2, h->cfg.NT * h->G, h->stream, &a));
        // the sequential apply sweep over q.NT propagators q.U (the step propagators, or the block propagators of the time-blocked sweep)
        auto apply_seq = [&](KrSweepArgs &q, const int kind = 3, const int blocks = 1) -> int {   // (kind 7, blocks = NB: the parallel pass at N <= 32)
            if (mf_apply) {
                int kt_a = (h->KT % 2 == 0) ? std::min(h->KT, 4) : 2;
                const int want_a = opt_int(h, "MFMA_APPLY_KT", 0);   // (tuning knob: 2 or 4 trajectories per workgroup)
                if ((want_a == 2 || want_a == 4) && h->Kp % want_a == 0 && (h->G == 1 || h->KT % want_a == 0 || want_a % h->KT == 0) && (h->G == 1 || want_a <= h->KT)) kt_a = want_a;
                q.gen_div = std::max(1, h->KT * h->sets / kt_a);
                HIPCHK(h, kr_launch_mf64(2, kt_a / 2, 3, h->Kp / kt_a, h->stream, &q));
                return KR_OK;
            }
            // The apply sweep is bound by streaming U_n (NP^2 * 16 B per workgroup per step) from L2: with a shared
            // generator, wider tiles (more trajectories per workgroup, fewer workgroups) cut that traffic.
            int kt_apply = std::min(h->KT, 4);   // (KT = 8 exists only for the matrix-core update sweep)
            if (h->G == 1) {
                const int want = opt_int(h, "APPLY_KT", 0);
                for (int cand : {4, 2}) {
                    if (cand <= kt_apply) break;
                    const bool pick = (cand == want);  // measured at C3: KT 1 -> 1.24 ms, 2 -> 1.44 ms, 4 -> 2.03 ms per sweep: default = the handle's KT
                    if (pick && h->Kp % (h->SG * cand) == 0) { kt_apply = cand; break; }
                }
            }
            q.gen_div = h->KT * h->sets / kt_apply;   // apply tiles per update tile (wg_gen is indexed by update tile)
            if (q.gen_div < 1) q.gen_div = 1;
            if (kind == 7) q.tb_nitems = h->Kp / (h->SG * kt_apply);
            HIPCHK(h, h->launch(h->LMAXv, kt_apply, kind, blocks * (h->Kp / (h->SG * kt_apply)), h->stream, &q));
            return KR_OK;
        };
        if (!tb) {
            rc = apply_seq(a);
            if (rc) return rc;
        } else if (h->tb_small) {
            // N <= 32: (ii) one chain of products per (generator, block), (iii) the sequential sweep over the block propagators, (iv) the
            // apply kernel again, one workgroup per (block, tile of trajectories), from the boundary states into the caller's storage
            KrSweepArgs pa = a;
            pa.tb_NT = h->cfg.NT; pa.tb_S = h->tb_S; pa.tb_NB = h->tb_NB; pa.tb_P = h->tb_P_d;
            HIPCHK(h, h->launch(h->LMAXv, 1, 8, h->G * h->tb_NB, h->stream, &pa));
            KrSweepArgs q = a;
            q.U = h->tb_P_d; q.NT = h->tb_NB; q.dir = +1; q.store = h->tb_bound_d;
            rc = apply_seq(q);
            if (rc) return rc;
            if (store) {
                KrSweepArgs r = a;
                r.tb_NT = h->cfg.NT; r.tb_S = h->tb_S; r.tb_NB = h->tb_NB; r.tb_bound = h->tb_bound_d; r.fin = nullptr;
                rc = apply_seq(r, 7, h->tb_NB);
                if (rc) return rc;
            }
        } else {
            // (ii) block propagators, in the operand order of the sequential kernel that consumes them
            KrSweepArgs pa = a;
            pa.tb_NT = h->cfg.NT; pa.tb_S = h->tb_S; pa.u_layout = mf_apply ? 1 : 0;
            if (h->tb_tree) {   // log2(S) levels of pairwise products, the last one in the consumer's operand order
                const double2 *in = h->U_d;
                int n = h->cfg.NT, lev = 0;
                for (int span = 1; span < h->tb_S; span *= 2, ++lev) {
                    const bool last = (2 * span >= h->tb_S);
                    double2 *out = last ? h->tb_P_d : h->tb_T_d[lev & 1];
                    HIPCHK(h, kr_launch_tbpair(h->G, n, lev == 0 ? 1 : 0, last ? 1 : 0, h->stream, &pa, in, out));
                    in = out;
                    n = (n + 1) / 2;
                }
            } else
            HIPCHK(h, kr_launch_tblock(0, h->G * h->tb_NB, h->stream, &pa, h->tb_P_d));
            // (iii) states at the block boundaries: the sequential sweep over the block propagators, in sweep order (dir = +1:
            //       boundary b + 1 after block b), final states as before
            KrSweepArgs q = a;
            q.U = h->tb_P_d; q.NT = h->tb_NB; q.dir = +1; q.store = h->tb_bound_d; q.u_layout = mf_apply ? 1 : 0;
            rc = apply_seq(q);
            if (rc) return rc;
            // (iv) all blocks in parallel from their boundary states, into the caller's storage
            if (store) {
                KrSweepArgs r = a;
                r.tb_NT = h->cfg.NT; r.tb_S = h->tb_S; r.tb_bound = h->tb_bound_d; r.tb_items = h->tb_items_d; r.tb_nitems = h->tb_nitems;
                HIPCHK(h, kr_launch_tblock(1, h->tb_NB * h->tb_nitems, h->stream, &r, nullptr));
            }
        }
    } else {
        const int kt_known = std::min(h->KT, 4);   // (KT = 8 exists only for the matrix-core update sweep)
        a.gen_div = h->KT * h->sets / kt_known;
        if (h->econ && h->NP > 64) {   // streamed-generator kernels: economized polynomial of every step (margin: rounding only)
            hipLaunchKernelGGL(kr_econ_rows_kernel, dim3(h->G * h->cfg.NT), dim3(64), 0, h->stream, eps_dev, h->dt_d, h->nrm_d, h->skw_d, h->G, h->cfg.NT,
                               std::min(h->cfg.L, KR_LMAX_CAP), 1 + KR_LMAX_CAP, h->tol, 1.0 + 1e-9, h->econ_d);
            HIPCHK(h, hipGetLastError());
            a.econ = h->econ_d;
        }
        HIPCHK(h, h->launch(h->LMAXv, kt_known, 0, h->Kp / (h->SG * kt_known), h->stream, &a));
    }
    return ev_mark(h, which_ev, true);
}

static int launch_update(kr_handle *h, const double *eps_guess_dev, double *eps_new_dev, int calib_steps = 0) {
    if (h->gen_update) {
        int rc0 = pack_precheck(h, 3);
        if (rc0) return rc0;
        rc0 = ev_mark(h, 2, false);
        if (rc0) return rc0;
        rc0 = launch_update_stepped(h, eps_guess_dev, eps_new_dev);
        if (rc0) return rc0;
        return ev_mark(h, 2, true);
    }
    if (h->cfg.world > 1 && !h->peers_connected && !calib_steps)
        return fail(h, KR_ERR_STATE, "world = %d but kr_ipc_connect has not been called", h->cfg.world);
    KrSweepArgs a;
    base_args(h, a);
    if (calib_steps) { a.NT = calib_steps; a.world = 1; a.rank = 0; }  // calibration: first steps only, this GPU alone
    else {
        int prc = pack_precheck(h, 0);
        if (prc == KR_OK && h->mfma_sweep) prc = pack_precheck(h, 2);   // the matrix-core sweep takes its A operands from the dense copies
        if (prc) return prc;
    }
    a.Hpack = h->Hpack_fw;
    a.eps = eps_guess_dev;
    a.init = h->psi0_d; a.store = h->store_fw ? h->Phi_d : nullptr; a.fin = h->psiT_d;
    a.X = h->X_d; a.alpha = h->S_d; a.eps_new = eps_new_dev; a.g_a_int = h->gint_d;
    a.dir = +1;
    a.dulog = h->dulog_d;
    const double *econ_src = eps_guess_dev;   // what the economized rows are made from: the amplitudes the matrices are combined with
    if (h->poly_fast && !calib_steps) {       // f_l(eps_guess), f_l'(eps_guess) of this sweep; the kernel evaluates f_l(eps_new) itself
        const size_t n = (size_t)h->cfg.L * h->cfg.NT;
        hipLaunchKernelGGL(kr_poly_eval, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, eps_guess_dev, (const double *)h->poly_d, h->poly_deg, h->cfg.L,
                           h->cfg.NT, h->amp_d, h->fprime_d);
        HIPCHK(h, hipGetLastError());
        a.amp_g = h->amp_d; a.fprime_g = h->fprime_d; a.poly = h->poly3_d;
        if (h->rowsplit) { a.eps = h->amp_d; a.amp_g = eps_guess_dev; }   // row-split kernels: amplitudes in the place of the pulses, the guess pulses beside them (kr_device_rowsplit.h)
        econ_src = h->amp_d;
    }
    a.epoch_base = h->epoch_base;
    h->last_epoch_base = h->epoch_base;
    // next launch: first tag = last tag + 3 (odd step: buffer parity keeps alternating); 32-bit wrap-around is
    // harmless because every slot is rewritten every second event and tags are only compared for equality
    h->epoch_base += (unsigned)h->cfg.NT + 2u;
    if (h->econ) {   // rows of the economized step polynomial from this sweep's guess pulses (device-resident pulses included)
        const int rows = h->G * h->cfg.NT;
        hipLaunchKernelGGL(kr_econ_rows_kernel, dim3(rows), dim3(64), 0, h->stream, econ_src, h->dt_d, h->nrm_d, h->skw_d, h->G, h->cfg.NT, std::min(h->cfg.L, KR_LMAX_CAP), 1 + KR_LMAX_CAP,
                           h->tol, 1.01, h->econ_d);
        HIPCHK(h, hipGetLastError());
        a.econ = h->econ_d;
    }
    // the mailbox inside the GPU validates its slots with one bit per double whose phase starts at 1 in every launch
    // (kr_exchange): start from zeros
    { const int zrc = zero_async(h, h->mail_d, h->mail_words); if (zrc) return zrc; }
    int rc = ev_mark(h, 2, false);
    if (rc) return rc;
    h->tl_launch0 = now_ms();
    if (h->rowsplit) {
        rc = rowsplit_prelaunch(h, a);
        if (rc) return rc;
        if (h->rs_turns > 0) a.rs_sets = h->rs_turns;
        HIPCHK(h, h->launch_rs(h->LMAXv, h->NR, h->rs_turns > 0 ? 5 : (h->rs_pair ? 3 : 1), h->P, h->stream, &a));
        rc = rowsplit_postlaunch(h, h->P / h->R, h->rs_turns > 0 ? h->rs_turns : h->rs_sets);
        if (rc) return rc;
    } else
    {   // update kernel variant: kind = 1 | SCHED << 4 | CAL << 5 | HERM << 6 (kr_launch_np); N <= 64 only
        int kind = 1;
        if (h->NP <= 64) kind |= ((h->poll_delay >= 0) ? 16 : 0) | (calib_steps ? 32 : 0) | (h->herm ? 64 : 0) | ((h->poly_fast && !calib_steps) ? 256 : 0);
        if (h->mfma_sweep) HIPCHK(h, kr_launch_mf64(h->LMAXv, (h->KT / 2) | (h->sets > 1 ? h->sets << 8 : 0), kind, h->P, h->stream, &a));
        else {
            a.wg_stride = (h->NP <= 64 && !calib_steps) ? h->wg_stride : 1;
            if (a.wg_stride > 1) kind = 1 | 128 | (h->herm ? 64 : 0);   // kr_sweep_update<.., XLOC = 1>: general variant, classic schedule / window variant, wave 1 publishes
            HIPCHK(h, h->launch(h->LMAXv, h->KT, kind, h->P * a.wg_stride, h->stream, &a));
        }
    }
    h->tl_launch1 = now_ms();
    rc = ev_mark(h, 2, true);
    if (rc) return rc;
    if (a.dulog && !calib_steps) {   // eps_new from workgroup 0's totals; every workgroup gathered the same totals in every step (~5 us)
        const size_t n = (size_t)h->cfg.L * h->cfg.NT;
        const unsigned slices = h->pulse_check ? (unsigned)std::max(1, std::min(64, h->P / 4)) : 1u;
        if (h->rowsplit) {   // the row-split update kernels write eps_new themselves: the logs are only compared
            hipLaunchKernelGGL(kr_pulse_finish, dim3((unsigned)((n + 255) / 256), slices), dim3(256), 0, h->stream, eps_guess_dev, (const double *)h->S_d, (const double *)nullptr,
                               (const double *)a.dulog, (double *)nullptr, h->P, h->cfg.L, KR_LMAX_CAP, h->cfg.NT, 1, h->status_d, 0);
            HIPCHK(h, hipGetLastError());
            return KR_OK;
        }
        hipLaunchKernelGGL(kr_pulse_finish, dim3((unsigned)((n + 255) / 256), slices), dim3(256), 0, h->stream, eps_guess_dev, (const double *)h->S_d, a.fprime_g, (const double *)a.dulog,
                           eps_new_dev, h->P, h->cfg.L, h->LMAXv, h->cfg.NT, h->pulse_check ? 1 : 0, h->status_d, 0);
        HIPCHK(h, hipGetLastError());
    }
    return KR_OK;
}

// The exchange of the update sweep has two schedules (kr_exchange): wave 0 publishes and polls (its first pass then also
// waits for the store's acknowledgement), or wave 1 publishes and wave 0 starts polling after a fixed delay, so that its
// FIRST pass already finds every tag.  The best delay is sharp (C3, P = 256: none 7.02, 0.43 us 6.96, 0.64 us 6.83, 0.85 us
// 6.66, 1.07 us 6.87, 1.28 us 7.05 us per step) and depends on the number of workgroups and on the memory system, so it is
// measured once per handle: a few short update sweeps (first <= 256 steps, zero states and pulses; results do not depend on
// the schedule, only the timing does).  KROTOV_HIP_POLL_DELAY fixes the value (-1 = classic), KROTOV_HIP_CALIBRATE=0
// skips the measurement.
static int calibrate_exchange(kr_handle *h) {
    const int fixed = opt_int(h, "POLL_DELAY", -2);
    if (fixed != -2) { h->poll_delay = fixed; return KR_OK; }
    // (N > 64: a step holds ~9 memory round trips of which the exchange is one: not worth 0.2 s of calibration)
    if (h->gen_update || h->P < 64 || h->NP > 64 || !opt_int(h, "CALIBRATE", 1)) return KR_OK;
    // One measurement per process and launch shape: later handles of the same shape reuse it (the schedule then no longer
    // depends on the timing noise of every single kr_create, and the ~30 calibration sweeps are paid once).
    // KROTOV_HIP_CALIBRATE_CACHE=0 measures every time.
    static std::mutex cal_mtx;
    static std::vector<std::pair<std::vector<int>, int>> cal_cache;
    const std::vector<int> key = {h->device, h->NP, h->LMAXv, h->KT, h->P, h->herm ? 1 : 0, h->mfma_sweep ? 1 : 0, h->cfg.L, h->cfg.N, h->sets, h->wg_stride};
    const bool use_cache = opt_int(h, "CALIBRATE_CACHE", 1) != 0;
    if (use_cache) {
        std::lock_guard<std::mutex> g(cal_mtx);
        for (auto &c : cal_cache) if (c.first == key) { h->poll_delay = c.second; return KR_OK; }
    }
    const int L = h->cfg.L, NT = h->cfg.NT, steps = std::min(NT, 256);
    const size_t slice = (size_t)h->Kp * h->cfg.N;
    const unsigned epoch_saved = h->epoch_base;   // calibration launches must not advance the tag sequence the ranks share
    HIPCHK(h, hipMemsetAsync(h->X_d, 0, slice * (size_t)(steps + 1) * sizeof(double2), h->stream));
    HIPCHK(h, hipMemsetAsync(h->eps_a, 0, (size_t)L * NT * sizeof(double), h->stream));
    HIPCHK(h, hipMemsetAsync(h->S_d, 0, (size_t)L * NT * sizeof(double), h->stream));
    hipEvent_t e0, e1;
    HIPCHK(h, hipEventCreate(&e0));
    HIPCHK(h, hipEventCreate(&e1));
    float best = 0.f;
    int best_delay = -1, rc = KR_OK;
    auto measure = [&](int delay) {
        h->poll_delay = delay;
        float tmin = 0.f;
        for (int rep = 0; rep < 3 && rc == KR_OK; ++rep) {  // rep 0 warms up
            if (hipEventRecord(e0, h->stream) != hipSuccess) { rc = fail(h, KR_ERR_HIP, "hipEventRecord failed"); break; }
            rc = launch_update(h, h->eps_a, h->eps_b, steps);
            if (rc != KR_OK) break;
            if (hipEventRecord(e1, h->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) { rc = fail(h, KR_ERR_HIP, "hipEventSynchronize failed"); break; }
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && (tmin == 0.f || ms < tmin)) tmin = ms;
        }
        if (rc == KR_OK) rc = check_status(h);
        if (rc == KR_OK && (best == 0.f || tmin < best)) { best = tmin; best_delay = delay; }
    };
    for (int d : {-1, 0, 4, 8, 12, 16, 20, 24}) if (rc == KR_OK) measure(d);   // coarse: 0.21 us apart (HERM: the window work
                                                                              // takes the place of most of the delay)
    if (rc == KR_OK && best_delay > 0) {                                // refine: 0.11 us apart
        const int c = best_delay;
        if (c >= 2) measure(c - 2);
        if (rc == KR_OK) measure(c + 2);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    h->poll_delay = best_delay;
    if (rc == KR_OK && use_cache) {
        std::lock_guard<std::mutex> g(cal_mtx);
        cal_cache.emplace_back(key, best_delay);
    }
    // The number of calibration launches depends on timing (refinement only for best_delay > 0), so ranks would leave
    // this function with different epoch bases and their inter-GPU tags would never match.  Restore the base and wipe the
    // calibration tags from the intra-GPU mailbox (calibration runs with world = 1: xmail was never written).
    h->epoch_base = epoch_saved;
    if (rc == KR_OK) {
        { const int zrc = zero_async(h, h->mail_d, h->mail_words); if (zrc) return zrc; }
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return rc;
}

static int check_status(kr_handle *h) {
    unsigned st[KR_STATUS_WORDS] = {};
    h->tl_wait0 = now_ms();
    HIPCHK(h, hipMemcpyAsync(st, h->status_d, sizeof st, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->tl_wait1 = now_ms();
    if (st[0] != 0) {
        // sticky until read (launches never clear it, so a failure inside a chain of *_dev calls is not lost)
        HIPCHK(h, hipMemsetAsync(h->status_d, 0, KR_STATUS_WORDS * sizeof(unsigned), h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (st[1] == 5)
            return fail(h, KR_ERR_UNSUPPORTED, "update sweep: at step %u an updated control left the range the %s propagator was initialised for (5 x the guess amplitude, transform_control_ranges); results of this call are invalid -- use prop_method = TAYLOR or a larger lambda_a",
                        st[2], h->method == KR_PROP_CHEBY ? "Chebyshev" : "Newton");
        if (st[1] == 7)
            return fail(h, KR_ERR_INTERNAL, "the %s generator pack on the device no longer matches the operators kr_create uploaded (checksum mismatch in front of a sweep): device memory was corrupted; results of this call are invalid -- create a new handle",
                        st[2] == 0 ? "forward" : st[2] == 1 ? "adjoint (backward-sweep)" : st[2] == 2 ? "dense (matrix-core)" : st[2] == 3 ? "column-major forward" : "column-major adjoint");
        if (st[1] == 8)
            return fail(h, KR_ERR_INTERNAL, "update sweep: workgroup %u gathered other totals than the reference workgroup and formed another pulse value (control %u, step %u: a partial sum of the per-step exchange was lost or accepted stale in one workgroup's gather); results of this call are invalid",
                        st[2], st[3] / (unsigned)h->cfg.NT, st[3] % (unsigned)h->cfg.NT);
        if (st[1] == 9)
            return fail(h, KR_ERR_INTERNAL, "mailbox self-test failed (workgroup %u, round %u): device memory of this process loses stores or hides them from some XCDs", st[2] & 0xffffu, (st[2] >> 16) & 0xffu);
        if (st[1] == 6)
            return fail(h, KR_ERR_INTERNAL, "row-split sweep: the %d workgroups that share trajectory tile %u ended the sweep with different copies of its state vector (an all-gather element was lost or accepted stale); results of this call are invalid",
                        h->R, st[2]);
        const char *what = st[1] == 4 ? "cross-rank sum of the taus (J_T_sm)"
                         : st[1] == 2 ? "inter-GPU exchange of the update sweep"
                         : st[1] == 3 ? "row-split all-gather (the R workgroups of a trajectory)"
                                      : "inter-workgroup exchange of the update sweep";
        std::string seen;
        if (st[1] == 2 && h->xmail_d && h->cfg.world > 1) {   // what did arrive: the tags in this GPU's mailbox, both buffers, one row per rank
            const int W = h->cfg.world, LM = h->LMAXv;
            std::vector<unsigned long long> xm((size_t)2 * W * 2 * LM, 0ull);
            if (hipMemcpyAsync(xm.data(), h->xmail_d, xm.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream) == hipSuccess && hipStreamSynchronize(h->stream) == hipSuccess) {
                char b[96];
                snprintf(b, sizeof b, "; expected tag %u in buffer %u, mailbox holds", h->last_epoch_base + st[2] + 1u, (h->last_epoch_base + st[2] + 1u) & 1u);
                seen = b;
                for (int par = 0; par < 2; ++par)
                    for (int r = 0; r < W; ++r) {
                        snprintf(b, sizeof b, " [buf %d rank %d:", par, r);
                        seen += b;
                        for (int w = 0; w < 2 * std::min(h->cfg.L, LM); ++w) { snprintf(b, sizeof b, " %u", (unsigned)(xm[((size_t)par * W + r) * 2 * LM + w] >> 32)); seen += b; }
                        seen += "]";
                    }
            }
        }
        if ((st[1] == 2 || st[1] == 1) && h->mail_d && !h->rowsplit && h->wg_stride == 1) {
            // ... and inside this GPU: how many of the P workgroups' partial sums of that step each replica of the mailbox holds
            // (layout: the packed slots of kr_exchange -- slot q at word q * LMAX of every replica, KR_SLOT_GRANULES; the one-XCD
            //  placement with its one slot per 128-byte line is excluded above; buffer and phase by the kernel's own helpers)
            const int LM = h->LMAXv, lpar = kr_mail_buffer(st[2]);
            const unsigned phase = kr_mail_phase(st[2]);
            std::vector<unsigned long long> mb((size_t)KR_MAIL_REPLICAS * KR_MAIL_RSTRIDE, 0ull);
            if (hipMemcpyAsync(mb.data(), h->mail_d + (size_t)lpar * KR_MAIL_REPLICAS * KR_MAIL_RSTRIDE, mb.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream) == hipSuccess &&
                hipStreamSynchronize(h->stream) == hipSuccess) {
                seen += "; partial sums of that step in the mailbox replicas (slot stride " + std::to_string(LM) + " words):";
                char b[48];
                for (int r = 0; r < KR_MAIL_REPLICAS; ++r) {
                    int cnt = 0;
                    for (int q = 0; q < h->P; ++q) cnt += ((unsigned)(mb[(size_t)r * KR_MAIL_RSTRIDE + (size_t)q * LM] & 1ull) == phase) ? 1 : 0;
                    snprintf(b, sizeof b, " %d", cnt);
                    seen += b;
                    if (cnt < h->P) {   // what the replica holds instead: its first words, and how far the damage reaches
                        char w[400];
                        const unsigned long long *rp = &mb[(size_t)r * KR_MAIL_RSTRIDE];
                        size_t nz = 0, last = 0;
                        for (size_t q = 0; q < KR_MAIL_RSTRIDE; ++q) if (rp[q]) { ++nz; last = q; }
                        snprintf(w, sizeof w, " {%016llx %016llx %016llx %016llx %016llx %016llx %016llx %016llx; nonzero words %zu, last at %zu}", rp[0], rp[1], rp[2], rp[3], rp[4], rp[5], rp[6], rp[7], nz, last);
                        seen += w;
                    }
                }
                snprintf(b, sizeof b, " of %d", h->P);
                seen += b;
            }
        }
        return fail(h, KR_ERR_TIMEOUT, "sweep aborted: bounded spin of the %s gave up at %s %u (status code %u); results of this call are invalid%s",
                    what, st[1] == 3 ? "gather" : "step", st[2], st[1], seen.c_str());
    }
    return KR_OK;
}

static int download_states(kr_handle *h, double *dst, const double2 *src) {
    if (!dst) return KR_OK;
    const size_t N = h->cfg.N, K = h->cfg.K;
    if (h->identity_perm) {
        HIPCHK(h, hipMemcpyAsync(dst, src, K * N * sizeof(double2), hipMemcpyDeviceToHost, h->stream));
        return KR_OK;
    }
    h->stage.resize((size_t)h->Kp * N * 2);
    HIPCHK(h, hipMemcpyAsync(h->stage.data(), src, (size_t)h->Kp * N * sizeof(double2), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (size_t k = 0; k < K; ++k) memcpy(dst + k * N * 2, &h->stage[(size_t)h->perm[k] * N * 2], N * sizeof(double2));
    return KR_OK;
}

static int one_forward(kr_handle *h, const double *eps, double *psiT_out) {
    if (!h || !eps) return h ? fail(h, KR_ERR_INVALID, "kr_forward: null") : KR_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mtx);
    if (!h->have_psi0) return fail(h, KR_ERR_STATE, "kr_forward: initial states not set");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t pb = (size_t)h->cfg.L * h->cfg.NT * sizeof(double);
    HIPCHK(h, hipMemcpyAsync(h->eps_a, eps, pb, hipMemcpyHostToDevice, h->stream));
    int rc = launch_known(h, 0, h->eps_a, false, h->psi0_d, h->store_fw ? h->Phi_d : nullptr, h->psiT_d);
    if (rc) return rc;
    rc = check_status(h); if (rc) return rc;   // pack checksum; bounded spins of the row-split all-gather
    rc = download_states(h, psiT_out, h->psiT_d);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_final = true;
    return KR_OK;
}

static int one_backward(kr_handle *h, const double *eps_guess, const double *chiT) {
    if (!h || !eps_guess || !chiT) return h ? fail(h, KR_ERR_INVALID, "kr_backward: null") : KR_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mtx);
    HIPCHK(h, hipSetDevice(h->device));
    const size_t pb = (size_t)h->cfg.L * h->cfg.NT * sizeof(double);
    HIPCHK(h, hipMemcpyAsync(h->eps_a, eps_guess, pb, hipMemcpyHostToDevice, h->stream));
    int rc = upload_states(h, h->chiT_d, chiT);
    if (rc) return rc;
    rc = launch_known(h, 1, h->eps_a, true, h->chiT_d, h->X_d, nullptr);
    if (rc) return rc;
    rc = check_status(h); if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_X = true;
    return KR_OK;
}

static int set_params_locked(kr_handle *h, const double *S, const double *lambda) {
    for (int l = 0; l < h->cfg.L; ++l)
        if (!(lambda[l] != 0.0)) return fail(h, KR_ERR_INVALID, "lambda[%d] must be non-zero", l);
    // alpha[l][n] = S[l][n] / lambda[l] exactly as optimize.jl:354 rounds it (one IEEE division)
    const int L = h->cfg.L, NT = h->cfg.NT;
    h->stage.resize((size_t)L * NT);
    for (int l = 0; l < L; ++l)
        for (int n = 0; n < NT; ++n) h->stage[(size_t)l * NT + n] = S[(size_t)l * NT + n] / lambda[l];
    HIPCHK(h, hipMemcpyAsync(h->S_d, h->stage.data(), (size_t)L * NT * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));  // stage is reused
    h->have_params = true;
    return KR_OK;
}

static int one_set_update_params(kr_handle *h, const double *S, const double *lambda) {
    if (!h || !S || !lambda) return h ? fail(h, KR_ERR_INVALID, "kr_set_update_params: null") : KR_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mtx);
    HIPCHK(h, hipSetDevice(h->device));
    int rc = set_params_locked(h, S, lambda);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return KR_OK;
}

static int update_forward_locked(kr_handle *h, const double *eps_guess, const double *S, const double *lambda,
                                 double *eps_new_out, double *g_a_int_out, double *psiT_out) {
    if (!h->have_psi0) return fail(h, KR_ERR_STATE, "update sweep: initial states not set");
    if (!h->have_X) return fail(h, KR_ERR_STATE, "update sweep: no backward sweep has been run");
    const size_t pb = (size_t)h->cfg.L * h->cfg.NT * sizeof(double);
    HIPCHK(h, hipMemcpyAsync(h->eps_a, eps_guess, pb, hipMemcpyHostToDevice, h->stream));
    // eps_new starts as a copy of the guess (optimize.jl: pulses1 = copy(pulses0), workspace.jl:125)
    HIPCHK(h, hipMemcpyAsync(h->eps_b, h->eps_a, pb, hipMemcpyDeviceToDevice, h->stream));
    int rc = set_params_locked(h, S, lambda);
    if (rc) return rc;
    rc = launch_update(h, h->eps_a, h->eps_b);
    if (rc) return rc;
    rc = check_status(h);
    if (rc) return rc;
    if (eps_new_out) HIPCHK(h, hipMemcpyAsync(eps_new_out, h->eps_b, pb, hipMemcpyDeviceToHost, h->stream));
    if (g_a_int_out) HIPCHK(h, hipMemcpyAsync(g_a_int_out, h->gint_d, h->cfg.L * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    rc = download_states(h, psiT_out, h->psiT_d);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_final = true;
    return KR_OK;
}

static int one_update_forward(kr_handle *h, const double *eps_guess, const double *S, const double *lambda,
                                 double *eps_new_out, double *g_a_int_out, double *psiT_out) {
    if (!h || !eps_guess || !S || !lambda) return h ? fail(h, KR_ERR_INVALID, "kr_update_forward: null") : KR_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mtx);
    HIPCHK(h, hipSetDevice(h->device));
    return update_forward_locked(h, eps_guess, S, lambda, eps_new_out, g_a_int_out, psiT_out);
}

static int one_iteration(kr_handle *h, const double *eps_guess, const double *chiT, const double *S,
                            const double *lambda, double *eps_new_out, double *g_a_int_out, double *psiT_out) {
    if (!h || !eps_guess || !chiT || !S || !lambda) return h ? fail(h, KR_ERR_INVALID, "kr_iteration: null") : KR_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mtx);
    HIPCHK(h, hipSetDevice(h->device));
    const size_t pb = (size_t)h->cfg.L * h->cfg.NT * sizeof(double);
    HIPCHK(h, hipMemcpyAsync(h->eps_a, eps_guess, pb, hipMemcpyHostToDevice, h->stream));
    int rc = upload_states(h, h->chiT_d, chiT);
    if (rc) return rc;
    rc = launch_known(h, 1, h->eps_a, true, h->chiT_d, h->X_d, nullptr);
    if (rc) return rc;
    // (no status check here: the status words are sticky -- only check_status clears them -- so a backward sweep that
    //  gave up is reported by update_forward_locked's check after the update sweep)
    h->have_X = true;
    return update_forward_locked(h, eps_guess, S, lambda, eps_new_out, g_a_int_out, psiT_out);
}

// ---- device-resident chain ---------------------------------------------------------------------
static int one_forward_dev(kr_handle *h, const void *eps_dev) {
    if (!h || !eps_dev) return h ? fail(h, KR_ERR_INVALID, "kr_forward_dev: null") : KR_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mtx);
    if (!h->have_psi0) return fail(h, KR_ERR_STATE, "kr_forward_dev: initial states not set");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = launch_known(h, 0, (const double *)eps_dev, false, h->psi0_d, h->store_fw ? h->Phi_d : nullptr, h->psiT_d);
    if (rc) return rc;
    h->have_final = true;
    return KR_OK;
}

static int launch_tau(kr_handle *h) {
    const int K = h->Kp, N = h->cfg.N;  // padded slots have zero targets -> tau = 0
    hipLaunchKernelGGL(kr_tau_kernel, dim3((K + 3) / 4), dim3(256), 0, h->stream, h->tgt_d, h->psiT_d, h->tau_d, N, K);
    HIPCHK(h, hipGetLastError());
    return KR_OK;
}

static int one_iteration_functional_dev(kr_handle *h, int functional, const void *eps_guess_dev, void *eps_new_dev) {
    if (!h || !eps_guess_dev || !eps_new_dev) return h ? fail(h, KR_ERR_INVALID, "kr_iteration_functional_dev: null") : KR_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mtx);
    if (functional < KR_J_T_SM || functional > KR_J_T_RE) return fail(h, KR_ERR_INVALID, "unknown functional %d", functional);
    if (!h->have_psi0 || !h->have_tgt) return fail(h, KR_ERR_STATE, "device chain: initial and target states must be set");
    if (!h->have_final) return fail(h, KR_ERR_STATE, "device chain: no forward sweep has produced final states yet");
    if (!h->have_params) return fail(h, KR_ERR_STATE, "device chain: kr_set_update_params has not been called");
    // J_T_sm couples ALL trajectories: the sum of the taus crosses the ranks through the handle's own carrier -- the tau slots of
    // the peer mailboxes, one ncclAllReduce of 2 doubles, or the host callback
    const bool tau_rccl = h->cfg.world > 1 && h->gen_update && h->exchange == KR_EXCHANGE_RCCL;
    const bool tau_cb = h->cfg.world > 1 && h->gen_update && h->exchange == KR_EXCHANGE_CALLBACK;
    if (functional == KR_J_T_SM && h->cfg.world > 1) {
        if (tau_rccl && !h->rccl_comm) return fail(h, KR_ERR_STATE, "J_T_sm across ranks with exchange = RCCL: kr_rccl_init has not been called");
        if (tau_cb && !h->cb) return fail(h, KR_ERR_STATE, "J_T_sm across ranks with exchange = CALLBACK: kr_set_exchange_callback has not been called");
        if (!tau_rccl && !tau_cb && !h->peers_connected)
            return fail(h, KR_ERR_STATE, "J_T_sm couples trajectories across ranks: the sum of the taus travels through the peer mailboxes (kr_ipc_connect, or a device list)");
    }
    HIPCHK(h, hipSetDevice(h->device));
    const int K = h->Kp, N = h->cfg.N;
    int rc = launch_tau(h);
    if (rc) return rc;
    if (functional == KR_J_T_SM) {
        KrPeers peers;
        for (int r = 0; r < 8; ++r) peers.p[r] = h->peer_xmail[r];
        const unsigned tag = ++h->func_epoch;
        const bool via_mail = h->cfg.world > 1 && !tau_rccl && !tau_cb;
        hipLaunchKernelGGL(kr_tausum_kernel, dim3(1), dim3(256), 0, h->stream, h->tau_d, h->tausum_d, K, via_mail ? h->cfg.world : 1, h->cfg.rank, tag, peers);
        HIPCHK(h, hipGetLastError());
        if (tau_rccl) {   // (sum order = RCCL's: identical on every rank, as the update sweep's all-reduce is)
            const ncclResult_t r = kr_rccl()->AllReduce(h->tausum_d, h->tausum_d, 2, ncclDouble, ncclSum, h->rccl_comm, h->stream);
            if (r != ncclSuccess) { h->rccl_bad = true; return fail(h, KR_ERR_HIP, "ncclAllReduce of the tau sum failed: %s", kr_rccl()->GetErrorString(r)); }
        } else if (tau_cb) {
            double buf[2];
            HIPCHK(h, hipMemcpyAsync(buf, h->tausum_d, sizeof buf, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            const int crc = h->cb(h->cb_ctx, buf, 2);
            if (crc != 0) return fail(h, KR_ERR_HIP, "exchange callback returned %d for the tau sum", crc);
            HIPCHK(h, hipMemcpyAsync(h->tausum_d, buf, sizeof buf, hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));   // (buf is a local)
        } else if (h->cfg.world > 1) {
            hipLaunchKernelGGL(kr_taugather_kernel, dim3(1), dim3(64), 0, h->stream, h->xmail_d, h->tausum_d, h->cfg.world, tag, h->status_d, h->spin_ticks);
            HIPCHK(h, hipGetLastError());
        }
    }
    const size_t tot = (size_t)h->Kp * N;
    hipLaunchKernelGGL(kr_chi_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, functional, h->tgt_d,
                       h->tau_d, h->tausum_d, h->chiT_d, N, K, h->Kp, (double)h->cfg.K_global);
    HIPCHK(h, hipGetLastError());
    rc = launch_known(h, 1, (const double *)eps_guess_dev, true, h->chiT_d, h->X_d, nullptr);
    if (rc) return rc;
    h->have_X = true;
    const size_t pb = (size_t)h->cfg.L * h->cfg.NT * sizeof(double);
    HIPCHK(h, hipMemcpyAsync(eps_new_dev, eps_guess_dev, pb, hipMemcpyDeviceToDevice, h->stream));
    return launch_update(h, (const double *)eps_guess_dev, (double *)eps_new_dev);
}

static int one_synchronize(kr_handle *h) {
    if (!h) return KR_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mtx);
    HIPCHK(h, hipSetDevice(h->device));
    return check_status(h);
}

static int one_final_states(kr_handle *h, double *psiT_out) {
    if (!h || !psiT_out) return h ? fail(h, KR_ERR_INVALID, "kr_final_states: null") : KR_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mtx);
    if (!h->have_final) return fail(h, KR_ERR_STATE, "kr_final_states: no sweep has run");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = download_states(h, psiT_out, h->psiT_d);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return KR_OK;
}

static int one_taus(kr_handle *h, double *tau_out) {
    if (!h || !tau_out) return h ? fail(h, KR_ERR_INVALID, "kr_taus: null") : KR_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mtx);
    if (!h->have_final || !h->have_tgt) return fail(h, KR_ERR_STATE, "kr_taus: needs target states and a finished sweep");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = launch_tau(h);
    if (rc) return rc;
    h->stage.resize((size_t)h->Kp * 2);
    HIPCHK(h, hipMemcpyAsync(h->stage.data(), h->tau_d, (size_t)h->Kp * sizeof(double2), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int k = 0; k < h->cfg.K; ++k) { tau_out[2 * k] = h->stage[2 * (size_t)h->perm[k]]; tau_out[2 * k + 1] = h->stage[2 * (size_t)h->perm[k] + 1]; }
    return KR_OK;
}

static int one_g_a_int(kr_handle *h, double *out) {
    if (!h || !out) return h ? fail(h, KR_ERR_INVALID, "kr_g_a_int: null") : KR_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mtx);
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(out, h->gint_d, h->cfg.L * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return KR_OK;
}

static int one_read_storage(kr_handle *h, int which, int k, int n, double *out) {
    if (!h || !out) return h ? fail(h, KR_ERR_INVALID, "kr_read_storage: null") : KR_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mtx);
    if (k < 0 || k >= h->cfg.K || n < 0 || n > h->cfg.NT) return fail(h, KR_ERR_INVALID, "kr_read_storage: index (k=%d, n=%d) out of range", k, n);
    const double2 *base = nullptr;
    if (which == KR_STORAGE_BW) { if (!h->have_X) return fail(h, KR_ERR_STATE, "backward storage is empty"); base = h->X_d; }
    else if (which == KR_STORAGE_FW) { if (!h->store_fw) return fail(h, KR_ERR_STATE, "forward storage disabled (KR_NO_STORE_FW)"); if (!h->have_final) return fail(h, KR_ERR_STATE, "forward storage is empty"); base = h->Phi_d; }
    else return fail(h, KR_ERR_INVALID, "kr_read_storage: which = %d", which);
    HIPCHK(h, hipSetDevice(h->device));
    const size_t off = ((size_t)n * h->Kp + h->perm[k]) * h->cfg.N;
    HIPCHK(h, hipMemcpyAsync(out, base + off, (size_t)h->cfg.N * sizeof(double2), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return KR_OK;
}

// ---- per-step exchange outside the kernel ----------------------------------------------------------
extern "C" int kr_rccl_unique_id_size(void) { return (int)sizeof(ncclUniqueId); }

extern "C" int kr_rccl_get_unique_id(void *id_out) {
    return kr_guard(nullptr, "kr_rccl_get_unique_id", [&]() -> int {
        if (!id_out) return fail(nullptr, KR_ERR_INVALID, "kr_rccl_get_unique_id: null");
        KrRccl *r = kr_rccl();
        if (!r) return fail(nullptr, KR_ERR_UNSUPPORTED, "%s", kr_rccl_why());
        ncclUniqueId id;
        const ncclResult_t rc = r->GetUniqueId(&id);
        if (rc != ncclSuccess) return fail(nullptr, KR_ERR_HIP, "ncclGetUniqueId failed: %s", r->GetErrorString(rc));
        memcpy(id_out, &id, sizeof id);
        return KR_OK;
    });
}

// Communicators are kept for the life of the process (round 5): destroying one makes RCCL free its device buffers, and a block that
// lands on the address range of a freed uncached block loses stores -- handles with exchange = RCCL created and destroyed
// in turn were exactly that pattern.  A handle returns its communicator to a process-wide cache keyed by (device, world, rank) -- or
// by the device list --, the next handle of the same shape takes it from there and ignores its unique id.  All ranks must then make
// the same choice, which they do when they create their handles in the same order (SPMD); KROTOV_HIP_RCCL_CACHE=0: ncclCommDestroy
// as before.
struct KrCommEntry { std::vector<int> key; ncclComm_t comm; bool in_use; };
static std::mutex g_comm_mtx;
static std::vector<KrCommEntry> g_comms;
static ncclComm_t rccl_cache_take(const std::vector<int> &key) {
    std::lock_guard<std::mutex> g(g_comm_mtx);
    for (auto &e : g_comms) if (!e.in_use && e.key == key) { e.in_use = true; return e.comm; }
    return nullptr;
}
static void rccl_cache_add(const std::vector<int> &key, ncclComm_t c) {
    std::lock_guard<std::mutex> g(g_comm_mtx);
    g_comms.push_back({key, c, true});
}
// `bad`: a collective on this communicator failed -- it is never handed out again (and, like every cached one, never destroyed)
static bool rccl_cache_unuse(ncclComm_t c, const bool bad = false) {
    std::lock_guard<std::mutex> g(g_comm_mtx);
    for (auto &e : g_comms) if (e.comm == c) { e.in_use = bad; return true; }
    return false;
}
static void rccl_release(kr_handle *h) {
    if (!h->rccl_comm) return;
    if (!rccl_cache_unuse(h->rccl_comm, h->rccl_bad) && kr_rccl()) (void)kr_rccl()->CommDestroy(h->rccl_comm);   // (not cached: KROTOV_HIP_RCCL_CACHE=0)
    h->rccl_comm = nullptr;
}

static int one_rccl_init(kr_handle *h, const void *unique_id) {
    if (!h || !unique_id) return h ? fail(h, KR_ERR_INVALID, "kr_rccl_init: null") : KR_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mtx);
    KrRccl *r = kr_rccl();
    if (!r) return fail(h, KR_ERR_UNSUPPORTED, "%s", kr_rccl_why());
    if (h->rccl_comm) return fail(h, KR_ERR_STATE, "kr_rccl_init: already initialised");
    HIPCHK(h, hipSetDevice(h->device));
    const bool cache = opt_int(h, "RCCL_CACHE", 1) != 0;
    const std::vector<int> key = {0, h->device, h->cfg.world, h->cfg.rank};
    if (cache && (h->rccl_comm = rccl_cache_take(key)) != nullptr) return KR_OK;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    const ncclResult_t rc = r->CommInitRank(&h->rccl_comm, h->cfg.world, id, h->cfg.rank);
    if (rc != ncclSuccess) { h->rccl_comm = nullptr; return fail(h, KR_ERR_HIP, "ncclCommInitRank failed: %s", r->GetErrorString(rc)); }
    if (cache) rccl_cache_add(key, h->rccl_comm);
    return KR_OK;
}

static int one_set_exchange_callback(kr_handle *h, kr_allreduce_fn fn, void *ctx) {
    if (!h) return KR_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mtx);
    h->cb = fn;
    h->cb_ctx = ctx;
    return KR_OK;
}

// ---- multi-GPU mailbox wiring --------------------------------------------------------------------
// What a rank exports (kr_ipc_handle_size() bytes, opaque to the caller): the IPC handle of its mailbox, how often its process has
// handed that block out (special_alloc), and whether the block is pooled (never freed while the exporter lives).
struct KrIpcBlob { hipIpcMemHandle_t mh; unsigned long long gen; unsigned pooled, pad; };
extern "C" int kr_ipc_handle_size(void) { return (int)sizeof(KrIpcBlob); }

static int one_ipc_export(kr_handle *h, void *handle_out) {
    if (!h || !handle_out) return h ? fail(h, KR_ERR_INVALID, "kr_ipc_export: null") : KR_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mtx);
    if (h->cfg.world <= 1) return fail(h, KR_ERR_STATE, "kr_ipc_export: world = 1");
    if (h->gen_update) return fail(h, KR_ERR_STATE, "kr_ipc_export: this handle exchanges per step (RCCL / callback), it has no mailboxes to wire");
    HIPCHK(h, hipSetDevice(h->device));
    KrIpcBlob b;
    memset(&b, 0, sizeof b);
    HIPCHK(h, hipIpcGetMemHandle(&b.mh, h->xmail_d));
    b.gen = h->xmail_gen;
    b.pooled = opt_int(h, "UC_POOL", 1) != 0 ? 1u : 0u;
    memcpy(handle_out, &b, sizeof b);
    return KR_OK;
}

// A peer's mailbox is an uncached block of another process; unmapping it here (hipIpcCloseMemHandle) is the same event as freeing an
// uncached block of our own (special_alloc): mappings are opened once per process and exporter block and kept (the exporter pools its
// blocks, so a later handle of the same peer presents the same IPC handle again and gets the mapping that is already there).
static std::mutex g_ipc_mtx;
static std::vector<std::tuple<int, hipIpcMemHandle_t, void *>> g_ipc_open;   // (device, handle, mapping)
// (a mapping is cached only when the EXPORTER keeps its block: the cache is keyed by the raw handle bytes, and a block that its
//  owner really frees could come back under the same bytes with the old mapping pointing at released memory)
static hipError_t ipc_open(kr_handle *h, const hipIpcMemHandle_t &mh, bool exporter_pooled, void **p, bool *cached) {
    *cached = false;
    if (opt_int(h, "UC_POOL", 1) == 0 || !exporter_pooled) return hipIpcOpenMemHandle(p, mh, hipIpcMemLazyEnablePeerAccess);
    std::lock_guard<std::mutex> g(g_ipc_mtx);
    *cached = true;
    for (auto &e : g_ipc_open)
        if (std::get<0>(e) == h->device && memcmp(&std::get<1>(e), &mh, sizeof mh) == 0) { *p = std::get<2>(e); return hipSuccess; }
    const hipError_t rc = hipIpcOpenMemHandle(p, mh, hipIpcMemLazyEnablePeerAccess);
    if (rc == hipSuccess) g_ipc_open.emplace_back(h->device, mh, *p);
    return rc;
}
static int one_ipc_connect(kr_handle *h, const void *all_handles) {
    if (!h || !all_handles) return h ? fail(h, KR_ERR_INVALID, "kr_ipc_connect: null") : KR_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mtx);
    if (h->cfg.world <= 1) return fail(h, KR_ERR_STATE, "kr_ipc_connect: world = 1");
    HIPCHK(h, hipSetDevice(h->device));
    unsigned long long mix = 0ull;
    for (int r = 0; r < h->cfg.world; ++r) {
        KrIpcBlob b;
        memcpy(&b, (const char *)all_handles + (size_t)r * sizeof b, sizeof b);
        mix = mix * 1000003ull + b.gen;
        if (r == h->cfg.rank) continue;
        void *p = nullptr;
        bool cached = false;
        HIPCHK(h, ipc_open(h, b.mh, b.pooled != 0, &p, &cached));
        h->peer_xmail[r] = (unsigned long long *)p;
        h->peer_cached[r] = cached;
    }
    // Tag base of this set of handles: every rank sees the same array, so every rank forms the same base -- and a set that recycles a
    // pooled mailbox forms another one than the set that had it before (a late store of the earlier set cannot satisfy a poll).
    h->epoch_base = (unsigned)((mix * 0x9E3779B97F4A7C15ull) >> 34);   // 30 bits: far from the 32-bit wrap
    h->func_epoch = h->epoch_base;
    h->peers_connected = true;
    return KR_OK;
}

// ==== one handle, several GPUs (kr_config.n_devices > 1) ================================================
// The reference threads over trajectories inside ONE process (@threadsif, /root/reference/src/optimize.jl:182,303,321,360);
// a Julia host therefore wants ONE handle that drives all GPUs of the node.  The group handle block-partitions the
// trajectories over P shard handles (rank i of world P, one per listed device), wires their exchange in-process (peer access
// instead of IPC for the mailboxes; ncclCommInitAll for RCCL) and runs every call on all shards from P threads at once --
// the update sweeps of the shards wait for each other inside their kernels, so they must be in flight together.
static void group_quiesce(kr_handle *g);
template <class F>
static int group_run(kr_handle *g, F f) {
    const int P = (int)g->subs.size();
    std::vector<int> rcs(P, KR_OK);
    std::vector<std::thread> th;
    // a shard's body never lets an exception leave its thread (that would be std::terminate inside a ccall)
    // (KROTOV_HIP_SHARD_SKEW_MS, a test facility: shard i starts i x that many ms late -- the shards of other processes / hosts are never in step either)
    const int skew = opt_int(g->subs[0], "SHARD_SKEW_MS", 0);
    auto body = [&rcs, &f, g, skew](int i) {
        if (skew > 0 && i > 0) std::this_thread::sleep_for(std::chrono::milliseconds((long)skew * i));
        rcs[i] = kr_guard(g->subs[i], "shard thread", [&] { return f(g->subs[i], i); });
    };
    try {
        for (int i = 1; i < P; ++i) th.emplace_back(body, i);
    } catch (...) {   // std::system_error: could not start a thread -- the shards already running are joined before the error leaves
        for (auto &t : th) t.join();
        throw;
    }
    body(0);
    for (auto &t : th) t.join();
    for (int i = 0; i < P; ++i)
        if (rcs[i] != KR_OK) { group_quiesce(g); break; }   // a failed call returns with every shard idle (asynchronous chain calls do not wait otherwise)
    for (int i = 0; i < P; ++i)
        if (rcs[i] != KR_OK) {
            std::string tl;
            if (rcs[i] == KR_ERR_TIMEOUT) {   // who was late: every shard's host timeline of its last update sweep
                char b[256];
                tl = "; timelines [ms]:";
                for (int j = 0; j < P; ++j) {
                    const kr_handle *s = g->subs[j];
                    snprintf(b, sizeof b, " shard %d launch %.2f..%.2f wait %.2f..%.2f (rc %d);", j, s->tl_launch0, s->tl_launch1, s->tl_wait0, s->tl_wait1, rcs[j]);
                    tl += b;
                    if (j != i && rcs[j] != KR_OK) tl += " shard " + std::to_string(j) + " reports: " + s->err + ";";
                }
            }
            return fail(g, rcs[i], "shard %d of %d (device %d): %s%s", i, P, g->subs[i]->device, g->subs[i]->err.c_str(), tl.c_str());
        }
    return KR_OK;
}
static inline bool is_group(const kr_handle *h) { return h && !h->subs.empty(); }
static inline size_t soff(const kr_handle *g, int i) { return (size_t)g->k_off[i] * g->cfg.N * 2; }   // doubles into a [K][N] complex array

// Every shard's work has left its device.  The shards' persistent kernels store into EACH OTHER's mailboxes, so no shard's memory may
// be released (and no failed call may return to a caller who will release it) while any sibling's kernel can still be running: a
// sweep that timed out on one shard returns there at once, while a late sibling keeps spinning -- and storing -- up to its own bound.
static void group_quiesce(kr_handle *g) {
    for (kr_handle *s : g->subs)
        if (s && s->stream) { (void)hipSetDevice(s->device); (void)hipStreamSynchronize(s->stream); }
}
static int group_destroy(kr_handle *g) {
    group_quiesce(g);   // ALL streams first: one_destroy waits for its own shard's stream only
    bool bad = false;   // a failed collective on one shard's communicator retires the whole set (rccl_release)
    for (kr_handle *s : g->subs) if (s && s->rccl_bad) bad = true;
    for (kr_handle *s : g->subs) if (s && bad) s->rccl_bad = true;
    for (kr_handle *s : g->subs) if (s) (void)one_destroy(s);
    delete g;
    return KR_OK;
}

static int group_create(const kr_config *cfg, kr_handle **out) {
    const int P = cfg->n_devices;
    if (P > 8) return fail(nullptr, KR_ERR_INVALID, "kr_create: n_devices = %d > 8", P);
    if (cfg->rank != 0 || cfg->world != 1) return fail(nullptr, KR_ERR_INVALID, "kr_create: a device list needs rank = 0, world = 1 (the handle shards by itself)");
    if (cfg->K < P) return fail(nullptr, KR_ERR_INVALID, "kr_create: K = %d trajectories over %d devices", cfg->K, P);
    if (cfg->exchange == KR_EXCHANGE_CALLBACK) return fail(nullptr, KR_ERR_UNSUPPORTED, "kr_create: exchange = CALLBACK with a device list (the shards exchange among themselves)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, KR_ERR_NO_DEVICE, "no HIP device available; libkrotov_hip has no CPU fallback");
    int share[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < P; ++i) {
        if (cfg->devices[i] < 0 || cfg->devices[i] >= ndev) return fail(nullptr, KR_ERR_INVALID, "kr_create: devices[%d] = %d out of range (%d devices)", i, cfg->devices[i], ndev);
        for (int j = 0; j < P; ++j) if (cfg->devices[j] == cfg->devices[i]) share[i]++;
        // Repeating an ordinal is a TEST facility (one-GPU boxes): the shards' kernels wait for each other, so each needs a
        // hardware queue of its own, and HIP multiplexes a process's streams over few of them (GPU_MAX_HW_QUEUES, default 4;
        // four shards on one device timed out in their first exchange).
        if (share[i] > 3) return fail(nullptr, KR_ERR_UNSUPPORTED, "kr_create: device %d is listed %d times; at most 3 shards can share one GPU (hardware queues per process)", cfg->devices[i], share[i]);
    }
    kr_handle *g = new kr_handle();
    try {
    g->cfg = *cfg;
    g->cfg.traj_generator = nullptr; g->cfg.tlist = nullptr; g->cfg.H0 = nullptr; g->cfg.Hc = nullptr; g->cfg.hc_present = nullptr;
    g->k_off.assign(P + 1, 0);
    for (int i = 0; i < P; ++i) g->k_off[i + 1] = g->k_off[i] + cfg->K / P + (i < cfg->K % P ? 1 : 0);
    g->subs.assign(P, nullptr);
    auto bail = [&](int code) { g_create_error = g->err; (void)group_destroy(g); return code; };
    for (int i = 0; i < P; ++i) {
        kr_config c = *cfg;
        c.n_devices = 0;
        c.device = cfg->devices[i];
        c.rank = i; c.world = P;
        c.K = g->k_off[i + 1] - g->k_off[i];
        c.K_global = cfg->K;
        if (cfg->traj_generator) c.traj_generator = cfg->traj_generator + g->k_off[i];
        if (share[i] > 1) {   // shards that share a GPU split its CUs (their persistent kernels must all be resident at once)
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, c.device) != hipSuccess) { fail(g, KR_ERR_HIP, "hipGetDeviceProperties failed"); return bail(KR_ERR_HIP); }
            // ... XCD by XCD: the dispatcher deals a launch's workgroups out to the 8 XCDs in turn, so a kernel of P workgroups
            // needs ceil(P / 8) CUs on some XCDs (one persistent workgroup per CU in the big variants), and the shards' kernels all start
            // with the same XCD: three shards of 82 workgroups (85 = 256 / 3 allowed) put 33 on the 32 CUs of XCD 0 and 1 -- found by
            // the fuzzer (ORACLE_FUZZ_DEVLIST=3, round 4) as timeouts of the first exchange.  Whole rounds of 8 only.
            // Three kernels that fill the chip to 240 of 256 CUs still missed each other in 1-5 of 150 fuzzed shapes (more hardware
            // queues -- GPU_MAX_HW_QUEUES=8 -- 1 of 150; 64 workgroups per shard: 0 of 150; two shards of 128 never, 962 shapes): three
            // shards take a quarter of the chip each.
            const int cap = std::max(1, (prop.multiProcessorCount / 8 / (share[i] > 2 ? share[i] + 1 : share[i])) * 8);
            c.max_workgroups = (cfg->max_workgroups > 0) ? std::min(cfg->max_workgroups, cap) : cap;
        }
        const int rc = one_create(&c, &g->subs[i]);
        if (rc != KR_OK) { fail(g, rc, "shard %d (device %d): %s", i, c.device, g_create_error.c_str()); return bail(rc); }
    }
    // exchange wiring, in-process
    const int ex = g->subs[0]->exchange;
    for (int i = 1; i < P; ++i)
        if (g->subs[i]->exchange != ex || g->subs[i]->gen_update != g->subs[0]->gen_update) { fail(g, KR_ERR_UNSUPPORTED, "the shards chose different exchange paths (uneven trajectory counts); set kr_config.exchange explicitly"); return bail(KR_ERR_UNSUPPORTED); }
    if (!g->subs[0]->gen_update) {   // mailboxes: every shard writes into every other shard's xmail directly
        for (int i = 0; i < P; ++i) {
            if (hipSetDevice(g->subs[i]->device) != hipSuccess) { fail(g, KR_ERR_HIP, "hipSetDevice failed"); return bail(KR_ERR_HIP); }
            for (int j = 0; j < P; ++j) {
                if (g->subs[j]->device != g->subs[i]->device) {
                    const hipError_t e = hipDeviceEnablePeerAccess(g->subs[j]->device, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { fail(g, KR_ERR_HIP, "hipDeviceEnablePeerAccess(%d -> %d) failed: %s", g->subs[i]->device, g->subs[j]->device, hipGetErrorString(e)); return bail(KR_ERR_HIP); }
                    (void)hipGetLastError();
                }
                g->subs[i]->peer_xmail[j] = g->subs[j]->xmail_d;
            }
            g->subs[i]->peers_connected = true;
            g->subs[i]->peers_in_process = true;
        }
        {   // one tag base per group, another for every group this process creates (pooled mailboxes: see one_ipc_connect)
            static std::mutex seq_mtx;
            static unsigned seq = 0;
            unsigned mine;
            { std::lock_guard<std::mutex> lk(seq_mtx); mine = ++seq; }
            const unsigned base = (unsigned)(((unsigned long long)mine * 0x9E3779B97F4A7C15ull) >> 34);
            for (int i = 0; i < P; ++i) { g->subs[i]->epoch_base = base; g->subs[i]->func_epoch = base; }
        }
    } else if (ex == KR_EXCHANGE_RCCL) {
        KrRccl *r = kr_rccl();
        if (!r) { fail(g, KR_ERR_UNSUPPORTED, "%s", kr_rccl_why()); return bail(KR_ERR_UNSUPPORTED); }
        for (int i = 0; i < P; ++i) if (share[i] > 1) { fail(g, KR_ERR_UNSUPPORTED, "exchange = RCCL needs distinct devices (device %d is listed %d times)", cfg->devices[i], share[i]); return bail(KR_ERR_UNSUPPORTED); }
        std::vector<ncclComm_t> comms(P, nullptr);
        const bool cache = opt_int(g->subs[0], "RCCL_CACHE", 1) != 0;
        auto key_of = [&](int i) { std::vector<int> k = {1, P, i}; for (int j = 0; j < P; ++j) k.push_back(cfg->devices[j]); return k; };
        bool have = cache;
        for (int i = 0; i < P && have; ++i) have = (comms[i] = rccl_cache_take(key_of(i))) != nullptr;
        if (!have) {
            for (int i = 0; i < P; ++i) if (comms[i]) { (void)rccl_cache_unuse(comms[i]); comms[i] = nullptr; }   // (a partial set goes back)
            const ncclResult_t rc = r->CommInitAll(comms.data(), P, cfg->devices);
            if (rc != ncclSuccess) { fail(g, KR_ERR_HIP, "ncclCommInitAll failed: %s", r->GetErrorString(rc)); return bail(KR_ERR_HIP); }
            if (cache) for (int i = 0; i < P; ++i) rccl_cache_add(key_of(i), comms[i]);
        }
        for (int i = 0; i < P; ++i) g->subs[i]->rccl_comm = comms[i];
    }
    } catch (...) {   // no half-built group (and none of its shards' device memory) survives an exception
        (void)group_destroy(g);
        throw;
    }
    *out = g;
    return KR_OK;
}

// ---- the exported entry points: one handle or a group ---------------------------------------------------
// Every body runs inside kr_guard (nothing throws across the boundary).
#define KR_NOGROUP(h, what) if (is_group(h)) return fail(h, KR_ERR_STATE, what ": not available on a handle with a device list")
#define KR_ENTRY(h, name, ...) return kr_guard((h), name, [&]() -> int { __VA_ARGS__ })
extern "C" int kr_create(const kr_config *cfg, kr_handle **out) {
    if (out) *out = nullptr;
    KR_ENTRY(nullptr, "kr_create",
        if (cfg && out && cfg->abi_version == KR_ABI_VERSION && cfg->n_devices > 1) return group_create(cfg, out);
        return one_create(cfg, out););
}
extern "C" int kr_destroy(kr_handle *h) { KR_ENTRY(nullptr, "kr_destroy", return is_group(h) ? group_destroy(h) : one_destroy(h);); }
extern "C" int kr_set_stream(kr_handle *h, void *st) { KR_ENTRY(h, "kr_set_stream", KR_NOGROUP(h, "kr_set_stream"); return one_set_stream(h, st);); }
extern "C" int kr_enable_timing(kr_handle *h, int on) {
    KR_ENTRY(h, "kr_enable_timing",
        if (is_group(h)) return group_run(h, [on](kr_handle *s, int) { return one_enable_timing(s, on); });
        return one_enable_timing(h, on););
}
extern "C" int kr_get_timings(kr_handle *h, double *t) { KR_ENTRY(h, "kr_get_timings", return one_get_timings(is_group(h) ? h->subs[0] : h, t);); }
extern "C" int kr_get_info(kr_handle *h, int32_t *info) { KR_ENTRY(h, "kr_get_info", return one_get_info(is_group(h) ? h->subs[0] : h, info);); }
extern "C" int kr_debug_stamps(kr_handle *h, unsigned long long *o) { KR_ENTRY(h, "kr_debug_stamps", return one_debug_stamps(is_group(h) ? h->subs[0] : h, o);); }
static int one_get_norm_bounds(kr_handle *h, double *out) {
    if (!h || !out) return h ? fail(h, KR_ERR_INVALID, "kr_get_norm_bounds: null") : KR_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mtx);
    std::copy(h->nrm_host.begin(), h->nrm_host.end(), out);
    return KR_OK;
}
extern "C" int kr_get_norm_bounds(kr_handle *h, double *out) { KR_ENTRY(h, "kr_get_norm_bounds", return one_get_norm_bounds(is_group(h) ? h->subs[0] : h, out);); }
extern "C" int kr_debug_arrivals(kr_handle *h, unsigned long long *o, int words) { KR_ENTRY(h, "kr_debug_arrivals", return one_debug_arrivals(is_group(h) ? h->subs[0] : h, o, words);); }
extern "C" const char *kr_plan_description(const kr_handle *h) {
    if (!h) return "";
    const kr_handle *s = is_group(h) ? h->subs[0] : h;
    return s->plan_text.c_str();
}
extern "C" int kr_set_initial_states(kr_handle *h, const double *psi0) {
    KR_ENTRY(h, "kr_set_initial_states",
        if (is_group(h)) { if (!psi0) return fail(h, KR_ERR_INVALID, "kr_set_initial_states: null"); return group_run(h, [h, psi0](kr_handle *s, int i) { return one_set_initial_states(s, psi0 + soff(h, i)); }); }
        return one_set_initial_states(h, psi0););
}
extern "C" int kr_set_target_states(kr_handle *h, const double *tgt) {
    KR_ENTRY(h, "kr_set_target_states",
        if (is_group(h)) { if (!tgt) return fail(h, KR_ERR_INVALID, "kr_set_target_states: null"); return group_run(h, [h, tgt](kr_handle *s, int i) { return one_set_target_states(s, tgt + soff(h, i)); }); }
        return one_set_target_states(h, tgt););
}
extern "C" int kr_forward(kr_handle *h, const double *eps, double *psiT_out) {
    KR_ENTRY(h, "kr_forward",
        if (is_group(h)) return group_run(h, [h, eps, psiT_out](kr_handle *s, int i) { return one_forward(s, eps, psiT_out ? psiT_out + soff(h, i) : nullptr); });
        return one_forward(h, eps, psiT_out););
}
extern "C" int kr_backward(kr_handle *h, const double *eps_guess, const double *chiT) {
    KR_ENTRY(h, "kr_backward",
        if (is_group(h)) { if (!chiT) return fail(h, KR_ERR_INVALID, "kr_backward: null"); return group_run(h, [h, eps_guess, chiT](kr_handle *s, int i) { return one_backward(s, eps_guess, chiT + soff(h, i)); }); }
        return one_backward(h, eps_guess, chiT););
}
extern "C" int kr_set_update_params(kr_handle *h, const double *S, const double *lambda) {
    KR_ENTRY(h, "kr_set_update_params",
        if (is_group(h)) return group_run(h, [S, lambda](kr_handle *s, int) { return one_set_update_params(s, S, lambda); });
        return one_set_update_params(h, S, lambda););
}
extern "C" int kr_update_forward(kr_handle *h, const double *eps_guess, const double *S, const double *lambda, double *eps_new_out,
                                 double *g_a_int_out, double *psiT_out) {
    KR_ENTRY(h, "kr_update_forward",
        if (is_group(h))   // every shard computes the identical pulse: shard 0 delivers it
            return group_run(h, [=](kr_handle *s, int i) { return one_update_forward(s, eps_guess, S, lambda, i == 0 ? eps_new_out : nullptr, i == 0 ? g_a_int_out : nullptr, psiT_out ? psiT_out + soff(h, i) : nullptr); });
        return one_update_forward(h, eps_guess, S, lambda, eps_new_out, g_a_int_out, psiT_out););
}
extern "C" int kr_iteration(kr_handle *h, const double *eps_guess, const double *chiT, const double *S, const double *lambda,
                            double *eps_new_out, double *g_a_int_out, double *psiT_out) {
    KR_ENTRY(h, "kr_iteration",
        if (is_group(h)) {
            if (!chiT) return fail(h, KR_ERR_INVALID, "kr_iteration: null");
            return group_run(h, [=](kr_handle *s, int i) { return one_iteration(s, eps_guess, chiT + soff(h, i), S, lambda, i == 0 ? eps_new_out : nullptr, i == 0 ? g_a_int_out : nullptr, psiT_out ? psiT_out + soff(h, i) : nullptr); });
        }
        return one_iteration(h, eps_guess, chiT, S, lambda, eps_new_out, g_a_int_out, psiT_out););
}
// device-resident chain on a group: the pulse pointers must be readable and writable from every listed device (memory of
// one of them -- peer access among the listed devices is enabled at kr_create -- or managed memory); all shards write the
// identical updated pulse
extern "C" int kr_forward_dev(kr_handle *h, const void *eps_dev) {
    KR_ENTRY(h, "kr_forward_dev",
        if (is_group(h)) return group_run(h, [eps_dev](kr_handle *s, int) { return one_forward_dev(s, eps_dev); });
        return one_forward_dev(h, eps_dev););
}
extern "C" int kr_iteration_functional_dev(kr_handle *h, int functional, const void *eps_guess_dev, void *eps_new_dev) {
    KR_ENTRY(h, "kr_iteration_functional_dev",
        if (is_group(h)) return group_run(h, [=](kr_handle *s, int) { return one_iteration_functional_dev(s, functional, eps_guess_dev, eps_new_dev); });
        return one_iteration_functional_dev(h, functional, eps_guess_dev, eps_new_dev););
}
extern "C" int kr_synchronize(kr_handle *h) {
    KR_ENTRY(h, "kr_synchronize",
        if (is_group(h)) return group_run(h, [](kr_handle *s, int) { return one_synchronize(s); });
        return one_synchronize(h););
}
extern "C" int kr_final_states(kr_handle *h, double *psiT_out) {
    KR_ENTRY(h, "kr_final_states",
        if (is_group(h)) { if (!psiT_out) return fail(h, KR_ERR_INVALID, "kr_final_states: null"); return group_run(h, [h, psiT_out](kr_handle *s, int i) { return one_final_states(s, psiT_out + soff(h, i)); }); }
        return one_final_states(h, psiT_out););
}
extern "C" int kr_taus(kr_handle *h, double *tau_out) {
    KR_ENTRY(h, "kr_taus",
        if (is_group(h)) { if (!tau_out) return fail(h, KR_ERR_INVALID, "kr_taus: null"); return group_run(h, [h, tau_out](kr_handle *s, int i) { return one_taus(s, tau_out + 2 * (size_t)h->k_off[i]); }); }
        return one_taus(h, tau_out););
}
extern "C" int kr_g_a_int(kr_handle *h, double *out) { KR_ENTRY(h, "kr_g_a_int", return one_g_a_int(is_group(h) ? h->subs[0] : h, out);); }
extern "C" int kr_read_storage(kr_handle *h, int which, int k, int n, double *out) {
    KR_ENTRY(h, "kr_read_storage",
        if (is_group(h)) {
            if (k < 0 || k >= h->cfg.K) return fail(h, KR_ERR_INVALID, "kr_read_storage: k = %d out of range", k);
            int i = 0;
            while (k >= h->k_off[i + 1]) ++i;
            const int rc = one_read_storage(h->subs[i], which, k - h->k_off[i], n, out);
            return rc == KR_OK ? rc : fail(h, rc, "%s", h->subs[i]->err.c_str());
        }
        return one_read_storage(h, which, k, n, out););
}
extern "C" int kr_rccl_init(kr_handle *h, const void *id) { KR_ENTRY(h, "kr_rccl_init", KR_NOGROUP(h, "kr_rccl_init"); return one_rccl_init(h, id);); }
extern "C" int kr_set_exchange_callback(kr_handle *h, kr_allreduce_fn fn, void *ctx) { KR_ENTRY(h, "kr_set_exchange_callback", KR_NOGROUP(h, "kr_set_exchange_callback"); return one_set_exchange_callback(h, fn, ctx);); }
extern "C" int kr_ipc_export(kr_handle *h, void *o) { KR_ENTRY(h, "kr_ipc_export", KR_NOGROUP(h, "kr_ipc_export"); return one_ipc_export(h, o);); }
extern "C" int kr_ipc_connect(kr_handle *h, const void *a) { KR_ENTRY(h, "kr_ipc_connect", KR_NOGROUP(h, "kr_ipc_connect"); return one_ipc_connect(h, a);); }
