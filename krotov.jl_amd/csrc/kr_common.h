// Shared host/device declarations of the Krotov sweep kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define KR_MAX_THREADS 512      // NP=64 uses 512-thread workgroups (2 waves/SIMD), smaller NP 256
#define KR_LMAX_CAP 4           // largest L of the register-resident kernels
#define KR_THETA_MAX 1.0        // largest ||H dt|| handled by one Taylor sub-step
#define KR_MAX_ORDER 48
#define KR_SPIN_TIMEOUT_TICKS 400000000ull  // 4 s of s_memrealtime (100 MHz): bounded spin

// status words (device): [0] abort/timeout flag, [1] where, [2] step, [3] spare
#define KR_STATUS_WORDS 4

struct KrSweepArgs {
    // generator slices, thread-major: Hpack[m][jj][tid_in_team] (double2), m = 0..L
    const double2 *Hpack;
    const double *eps;       // [L][NT] pulses that drive this sweep (known-pulse sweeps) / guess (update)
    const double *dt;        // [NT]
    const double2 *init;     // [Kp][N] initial states (psi0, or chi(T) for the backward sweep)
    double2 *store;          // [(NT+1)][Kp][N] storage to write (or nullptr)
    double2 *fin;            // [Kp][N] final states (or nullptr)
    // update sweep only
    const double2 *X;        // [(NT+1)][Kp][N] co-states from the backward sweep
    const double *S;         // [L][NT]
    const double *lambda;    // [L]
    double *eps_new;         // [L][NT]
    double *g_a_int;         // [L]
    unsigned long long *mail;  // [2][P][2*LMAX] granules {tag32 | half of a double}
    unsigned long long *xmail; // [2][world][2*LMAX] inter-GPU granules (local memory, peers write)
    unsigned long long *peer_xmail[8]; // every rank's xmail (peer_xmail[rank] == xmail)
    unsigned *status;        // [KR_STATUS_WORDS]
    double nrm[1 + KR_LMAX_CAP]; // 2-norm bounds of H0, H_1..H_L
    double tol;
    unsigned present_mask;   // bit l set: control l has a derivative (optimize.jl:344)
    int N, NT, Kp, L;
    int dir;                 // +1 forward (store slot n+1), -1 backward (store slot n)
    int P;                   // workgroups
    int rank, world;
    unsigned epoch_base;     // tags = epoch_base + n + 1
    int n_begin, n_end;      // step range of this launch (update sweep may be split)
};

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

typedef hipError_t (*kr_launch_fn)(int variant_lmax, int variant_kt, int kind, int grid,
                                   hipStream_t stream, const KrSweepArgs *args);
// kind: 0 = known-pulse sweep (forward or backward), 1 = update sweep
hipError_t kr_launch_np8(int lmax, int kt, int kind, int grid, hipStream_t s, const KrSweepArgs *a);
hipError_t kr_launch_np16(int lmax, int kt, int kind, int grid, hipStream_t s, const KrSweepArgs *a);
hipError_t kr_launch_np32(int lmax, int kt, int kind, int grid, hipStream_t s, const KrSweepArgs *a);
hipError_t kr_launch_np64(int lmax, int kt, int kind, int grid, hipStream_t s, const KrSweepArgs *a);
