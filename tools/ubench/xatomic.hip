// Micro-benchmark (round 5): the per-step all-to-all sum of the update sweep, flat (what kr_exchange does) against a two-level form
// whose FIRST level is an integer atomic in the XCD's own L2.
//
//   variant 0  flat: every workgroup stores its two partial sums (16 B) into 8 mailbox replicas (one per XCD), wave 0 of every
//              workgroup polls all P slots of its XCD's replica (16-byte loads that bypass the vector L1) and sums them.
//   variant 1  two levels: the partial sums are converted to fixed point (exact: scaling by a power of two, split into a high and a
//              low word so that 82 bits below the bound survive) and ADDED with 64-bit atomics -- no scope bits: executed in the L2 of
//              the XCD the workgroup runs on -- into that XCD's accumulators of the step; every word carries an arrival count in
//              its low byte, so the atomic's RETURN tells the last arriver the complete XCD sum.  That workgroup alone publishes
//              the word to the 8 replicas; everybody polls 8 XCDs x 4 words = 32 words (one 8-byte load per lane).  Integer sums do
//              not depend on the order of arrival: bitwise reproducible like the fixed tree of the flat form.
//   usage: xatomic VARIANT [ITERS=2000] [WORK=0 (units of 64 cycles of s_sleep between exchanges)] [P=256]
// Prints cycles (s_memtime, 100 MHz realtime converted by the measured clock) per exchange of workgroup 0 and checks the totals.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef unsigned long long u64;
typedef long long i64;
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define NREP 8
#define RING 64   // accumulator entries in rotation (zeroed at launch; entry n % RING is re-zeroed by step n + RING / 2's leader)

__device__ __forceinline__ unsigned xcc_id() {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    return id & 7u;
}
__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct Args {
    u64 *mail;      // variant 0: [2][NREP][P][2] doubles-as-bits with LSB phase; variant 1: [2][NREP][8 xcd][4 words]
    u64 *acc;       // variant 1: [RING][8 xcd][4 words]
    double *out;    // [iters][2] totals seen by workgroup 0
    u64 *cycles;    // [1]
    int P, iters, work, members;   // members: workgroups per XCD (P / 8)
    double scale;   // 2^e: fixed-point scale
};

__global__ __launch_bounds__(128) void k_flat(Args a) {
    const int wg = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ double stot[2];
    __shared__ int sdead;
    if (threadIdx.x == 0) sdead = 0;
    __syncthreads();
    const unsigned xcc = xcc_id();
    u64 t0 = 0;
    for (int n = 0; n < a.iters; ++n) {
        if (n == 10 && wg == 0 && threadIdx.x == 0) t0 = __builtin_amdgcn_s_memtime();
        const int par = n & 1;
        const unsigned phase = ((n >> 1) & 1) ^ 1;
        const double p0 = 1e-3 * (wg + 1) + 1e-6 * n, p1 = -2e-3 * (wg + 1) + 3e-6 * n;
        if (wave == 1 && lane < 2 * NREP) {   // publish into every replica
            const int r = lane >> 1, l = lane & 1;
            const u64 bits = ((u64)__double_as_longlong(l ? p1 : p0) & ~1ull) | phase;
            u64 *dst = a.mail + (((size_t)par * NREP + r) * a.P + wg) * 2 + l;
            asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(dst), "v"(bits) : "memory");
        }
        if (wave == 0) {
            double t[2];
            const u64 spin0 = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                if (__builtin_amdgcn_s_memrealtime() - spin0 > 50000000ull) { sdead = 1; break; }   // 0.5 s: never hang the box
                bool ok = true;
                t[0] = t[1] = 0.0;
                for (int q = lane; q < a.P; q += 64) {
                    const u64 *src = a.mail + (((size_t)par * NREP + xcc) * a.P + q) * 2;
                    u4 g;
                    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g) : "v"(src) : "memory");
                    const u64 g0 = ((u64)g.y << 32) | g.x, g1 = ((u64)g.w << 32) | g.z;
                    ok = ok && ((g0 & 1ull) == phase) && ((g1 & 1ull) == phase);
                    t[0] += __longlong_as_double((i64)g0);
                    t[1] += __longlong_as_double((i64)g1);
                }
                if (__all(ok)) break;
            }
            t[0] = wave_sum(t[0]); t[1] = wave_sum(t[1]);
            if (lane == 0) { stot[0] = t[0]; stot[1] = t[1]; }
        }
        __syncthreads();
        if (sdead) { if (threadIdx.x == 0) a.cycles[1] = 1ull + (u64)n; return; }
        if (wg == 0 && threadIdx.x == 0) { a.out[2 * n] = stot[0]; a.out[2 * n + 1] = stot[1]; }
        for (int w = 0; w < a.work; ++w) __builtin_amdgcn_s_sleep(1);
        __syncthreads();
    }
    if (wg == 0 && threadIdx.x == 0) a.cycles[0] = __builtin_amdgcn_s_memtime() - t0;
}

// value v -> two fixed-point words (hi: multiples of 2^32, lo in [0, 2^32)), each shifted left by 8 with an arrival count of 1
__device__ __forceinline__ void to_fixed(const double v, const double scale, u64 &hi, u64 &lo) {
    const double t = v * scale;                         // exact (power of two)
    const double h = floor(t * (1.0 / 4294967296.0));   // |h| < 2^50
    const double l = t - h * 4294967296.0;              // [0, 2^32), exact
    hi = ((u64)(i64)h << 8) | 1ull;
    lo = ((u64)(i64)l << 8) | 1ull;
}

__global__ __launch_bounds__(128) void k_two(Args a) {
    const int wg = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ double stot[2];
    __shared__ int sdead;
    if (threadIdx.x == 0) sdead = 0;
    __syncthreads();
    const unsigned xcc = xcc_id();
    u64 t0 = 0;
    for (int n = 0; n < a.iters; ++n) {
        if (n == 10 && wg == 0 && threadIdx.x == 0) t0 = __builtin_amdgcn_s_memtime();
        const int par = n & 1;
        const u64 tag = (u64)(n % 255) + 1ull;   // never 0: the mailbox starts zeroed
        const double p0 = 1e-3 * (wg + 1) + 1e-6 * n, p1 = -2e-3 * (wg + 1) + 3e-6 * n;
        if (wave == 1 && lane < 4) {
            // lane w adds word w (0: hi of l = 0, 1: lo of l = 0, 2: hi of l = 1, 3: lo of l = 1) into this XCD's accumulator of the step
            u64 hi0, lo0, hi1, lo1;
            to_fixed(p0, a.scale, hi0, lo0);
            to_fixed(p1, a.scale, hi1, lo1);
            const u64 mine = (lane == 0) ? hi0 : (lane == 1) ? lo0 : (lane == 2) ? hi1 : lo1;
            u64 *accw = a.acc + (((size_t)(n % RING)) * 8 + xcc) * 4 + lane;
            const u64 old = __hip_atomic_fetch_add(accw, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // no scope bits: the XCD's L2
            if ((int)(old & 0xffull) == a.members - 1) {   // last arriver of this word: old + mine is the XCD's sum
                const u64 sum = old + mine;
                const u64 val = (sum & ~0xffull) | tag;
                for (int r = 0; r < NREP; ++r) {
                    u64 *dst = a.mail + (((size_t)par * NREP + r) * 8 + xcc) * 4 + lane;
                    asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(dst), "v"(val) : "memory");
                }
                // re-zero the entry half a ring ahead (last used RING / 2 steps ago: everybody is past it)
                a.acc[(((size_t)((n + RING / 2) % RING)) * 8 + xcc) * 4 + lane] = 0ull;
            }
        }
        if (wave == 0) {
            i64 w = 0;
            const u64 spin0 = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                if (__builtin_amdgcn_s_memrealtime() - spin0 > 50000000ull) { sdead = 1; break; }   // 0.5 s: never hang the box
                u64 g = 0;
                if (lane < 32) {
                    const u64 *src = a.mail + (((size_t)par * NREP + xcc) * 8) * 4 + lane;   // lane = 4 xcd + word
                    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g) : "v"(src) : "memory");
                }
                const bool ok = (lane >= 32) || ((g & 0xffull) == tag);
                if (__all(ok)) { w = (lane < 32) ? ((i64)g >> 8) : 0; break; }
            }
            // sum over the 8 XCDs per word (lanes 4 x + word): integer adds, any order
            for (int o = 4; o < 32; o <<= 1) w += __shfl_xor(w, o, 64);
            const i64 hi0 = __shfl(w, 0, 64), lo0 = __shfl(w, 1, 64), hi1 = __shfl(w, 2, 64), lo1 = __shfl(w, 3, 64);
            if (lane == 0) {
                stot[0] = ((double)hi0 * 4294967296.0 + (double)lo0) / a.scale;
                stot[1] = ((double)hi1 * 4294967296.0 + (double)lo1) / a.scale;
            }
        }
        __syncthreads();
        if (sdead) { if (threadIdx.x == 0) a.cycles[1] = 1ull + (u64)n; return; }
        if (wg == 0 && threadIdx.x == 0) { a.out[2 * n] = stot[0]; a.out[2 * n + 1] = stot[1]; }
        for (int w = 0; w < a.work; ++w) __builtin_amdgcn_s_sleep(1);
        __syncthreads();
    }
    if (wg == 0 && threadIdx.x == 0) a.cycles[0] = __builtin_amdgcn_s_memtime() - t0;
}

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 0, iters = argc > 2 ? atoi(argv[2]) : 2000, work = argc > 3 ? atoi(argv[3]) : 0;
    const int P = argc > 4 ? atoi(argv[4]) : 256;
    Args a;
    a.P = P; a.iters = iters; a.work = work; a.members = P / 8;
    a.scale = ldexp(1.0, 70);   // bound of |partial sums| here: < 2^10 -> |t| < 2^80: high word < 2^48
    const size_t mailw = (size_t)2 * NREP * (P + 64) * 2;
    CHK(hipExtMallocWithFlags((void **)&a.mail, mailw * 8, hipDeviceMallocUncached));
    CHK(hipMemset(a.mail, 0, mailw * 8));
    CHK(hipMalloc((void **)&a.acc, (size_t)RING * 8 * 4 * 8));
    CHK(hipMemset(a.acc, 0, (size_t)RING * 8 * 4 * 8));
    CHK(hipMalloc((void **)&a.out, (size_t)iters * 2 * 8));
    CHK(hipMalloc((void **)&a.cycles, 16));
    CHK(hipMemset(a.cycles, 0, 16));
    CHK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0));
    if (variant == 0) hipLaunchKernelGGL(k_flat, dim3(P), dim3(128), 0, 0, a);
    else hipLaunchKernelGGL(k_two, dim3(P), dim3(128), 0, 0, a);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<double> out((size_t)iters * 2);
    u64 cyc2[2] = {0, 0};
    CHK(hipMemcpy(out.data(), a.out, out.size() * 8, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(cyc2, a.cycles, 16, hipMemcpyDeviceToHost));
    const u64 cyc = cyc2[0];
    if (cyc2[1]) { printf("variant %d: a bounded spin gave up at exchange %llu (workgroups not spread evenly over the XCDs?)\n", variant, cyc2[1] - 1ull); return 2; }
    // check
    int bad = 0;
    double worst = 0;
    for (int n = 0; n < iters; ++n) {
        double s0 = 0, s1 = 0;
        for (int w = 0; w < P; ++w) {
            const double p0 = 1e-3 * (w + 1) + 1e-6 * n, p1 = -2e-3 * (w + 1) + 3e-6 * n;
            if (variant == 0) {   // the forced LSB moves a value by at most one ulp
                s0 += p0; s1 += p1;
            } else { s0 += p0; s1 += p1; }
        }
        const double d = fmax(fabs(out[2 * n] - s0) / fabs(s0), fabs(out[2 * n + 1] - s1) / fabs(s1));
        worst = fmax(worst, d);
        if (!(d < 1e-12)) ++bad;
    }
    printf("variant %d P=%d work=%d: %.3f us per exchange (kernel %.3f ms / %d), %.0f s_memtime ticks per exchange of workgroup 0; totals: %d bad, worst relative deviation %.2e\n",
           variant, P, work, 1e3 * ms / iters, ms, iters, (double)cyc / (iters - 10), bad, worst);
    return bad ? 1 : 0;
}
