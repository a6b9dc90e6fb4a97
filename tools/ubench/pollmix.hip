// What does a poll pass of the update sweep's exchange cost WHILE the partial sums travel, and why?
// 256 workgroups of two waves: wave 1 publishes a 16-byte slot {step, step} into the mailbox replicas at a common instant
// (paced by s_memrealtime), wave 0 polls its XCD's replica pass after pass (four 16-byte loads per lane = 256 slots, as
// kr_poll_pass).  Reported: the duration of a pass by the time since the workgroup's own publish (256-cycle bins), and the
// delay from the publish to the first pass that saw all slots of the step.
//   mode 0: publish into all 8 replicas (what kr_exchange does)
//   mode 1: publish into the 7 OTHER replicas only, slots of the own XCD ignored (are loads held behind this XCD's own
//           outstanding stores to the polled lines?)
//   mode 3: as 1, and the own XCD's slots go through a separate region with sc0 stores / sc1 loads (its L2)
//   ./pollmix [STEPS=400] [PERIOD_TICKS=600 (10 ns each)] [MEM=1 uncached]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned long long u64;
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define NBIN 48
#define OUTW (2 * NBIN + 4)

__device__ __forceinline__ unsigned xcc_id() {
    unsigned id;
    asm("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    return id & 7u;
}
__device__ __forceinline__ u64 mt() { u64 t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }
__device__ __forceinline__ u64 rt() { u64 t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }

__global__ void __launch_bounds__(128) pollmix(unsigned char *buf, unsigned char *loc, const int mode, const int nsteps, const unsigned period,
                                               u64 *sync, unsigned *claim, unsigned *out, const unsigned sstride, const unsigned nrep, const unsigned rspace, const unsigned window) {
    __shared__ volatile unsigned s_step, s_done;
    __shared__ volatile u64 s_pub;
    __shared__ unsigned hs[NBIN], hc[NBIN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned x = xcc_id();
    __shared__ unsigned s_rank;
    if (tid == 0) { s_rank = atomicAdd(&claim[x], 1u); s_step = 0; s_done = 0; s_pub = 0; }
    if (tid < NBIN) { hs[tid] = 0; hc[tid] = 0; }
    __syncthreads();
    const unsigned rank = s_rank;
    if (rank >= 32u) return;                       // placement not 32 per XCD: this workgroup sits out (both waves)
    const unsigned slot = x * 32u + rank;
    if (wave == 1) {
        if (slot == 0 && lane == 0) __hip_atomic_store(sync, rt() + 20000ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // 200 us from now
        u64 start = 0;
        for (unsigned spin = 0; spin < (1u << 22) && start == 0; ++spin) start = __hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (start != 0) {
            for (int s = 1; s <= nsteps; ++s) {
                const u64 due = start + (u64)s * period;
                for (unsigned spin = 0; spin < (1u << 20) && rt() < due; ++spin) __builtin_amdgcn_s_sleep(1);
                const u64 t = mt();
                const v4u val = {(unsigned)s, 0u, (unsigned)s, 0u};
                if ((unsigned)lane < nrep && !((mode & 1) && (unsigned)lane == x)) {
                    unsigned char *dst = buf + (size_t)lane * rspace + (size_t)slot * sstride;
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(val) : "memory");
                }
                if ((mode & 2) && lane == 8) {
                    unsigned char *dst = loc + (size_t)x * 4096 + (size_t)rank * 16;
                    asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(dst), "v"(val) : "memory");
                }
                if (lane == 0) { s_pub = t; s_step = (unsigned)s; }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) s_done = 1;
        return;
    }
    // wave 0: poll
    const unsigned voff = (unsigned)lane * sstride, vo1 = voff + 64u * sstride, vo2 = voff + 128u * sstride, vo3 = voff + 192u * sstride;
    const unsigned char *sb = buf + (size_t)(x % nrep) * rspace;
    const unsigned char *lb = loc + (size_t)x * 4096;
    unsigned lastdone = 0, ccount = 0, cmax = 0;
    u64 csum = 0;
    for (unsigned pass = 0; pass < (1u << 22) && !s_done; ++pass) {
        const unsigned cur = s_step;
        const u64 pt = s_pub;
        if (window) {   // as the update sweep: nothing is polled before the window work is over, nothing after the step is complete
            if (cur == lastdone) { __builtin_amdgcn_s_sleep(2); continue; }
            while (mt() < pt + window) __builtin_amdgcn_s_sleep(1);
        }
        v4u g0, g1, g2, g3, gl = {0u, 0u, 0u, 0u};
        const u64 t0 = mt();
        if (mode & 2) {
            asm volatile("global_load_dwordx4 %0, %5, %6 sc1\n\tglobal_load_dwordx4 %1, %5, %6 offset:1024 sc1\n\t"
                         "global_load_dwordx4 %2, %5, %6 offset:2048 sc1\n\tglobal_load_dwordx4 %3, %5, %6 offset:3072 sc1\n\t"
                         "global_load_dwordx4 %4, %5, %7 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3), "=&v"(gl) : "v"(voff), "s"(sb), "s"(lb) : "memory");
        } else {
            asm volatile("global_load_dwordx4 %0, %4, %8 sc1\n\tglobal_load_dwordx4 %1, %5, %8 sc1\n\t"
                         "global_load_dwordx4 %2, %6, %8 sc1\n\tglobal_load_dwordx4 %3, %7, %8 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3) : "v"(voff), "v"(vo1), "v"(vo2), "v"(vo3), "s"(sb) : "memory");
        }
        const u64 t1 = mt();
        bool ok = true;
        const unsigned v[4] = {g0.x, g1.x, g2.x, g3.x};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned q = (unsigned)(c * 64 + lane);
            const bool own = (q >> 5) == x;
            if (!((mode & 1) && own)) ok = ok && (v[c] >= cur);
        }
        if ((mode & 2) && lane < 32) ok = ok && (gl.x >= cur);
        const bool all = __all(ok);
        if (lane == 0 && cur != 0) {
            const u64 since = t0 - pt;
            const unsigned bin = (since >> 8) < (u64)NBIN ? (unsigned)(since >> 8) : (unsigned)(NBIN - 1);
            hs[bin] += (unsigned)(t1 - t0);
            hc[bin] += 1u;
        }
        if (all && cur > lastdone) {
            lastdone = cur;
            const unsigned d = (unsigned)(t1 - pt);
            csum += d; cmax = d > cmax ? d : cmax; ++ccount;
        }
    }
    if (lane < NBIN) {
        out[(size_t)slot * OUTW + lane] = hs[lane];
        out[(size_t)slot * OUTW + NBIN + lane] = hc[lane];
    }
    if (lane == 0) {
        out[(size_t)slot * OUTW + 2 * NBIN + 0] = ccount ? (unsigned)(csum / ccount) : 0u;
        out[(size_t)slot * OUTW + 2 * NBIN + 1] = cmax;
        out[(size_t)slot * OUTW + 2 * NBIN + 2] = ccount;
        out[(size_t)slot * OUTW + 2 * NBIN + 3] = 1u;
    }
}

int main(int argc, char **argv) {
    const int nsteps = argc > 1 ? atoi(argv[1]) : 400;
    const unsigned period = argc > 2 ? (unsigned)atoi(argv[2]) : 600u;
    const int mem = argc > 3 ? atoi(argv[3]) : 1;
    unsigned char *buf, *loc;
    const size_t bytes = (size_t)20 << 20;
    if (mem) CK(hipExtMallocWithFlags((void **)&buf, bytes, hipDeviceMallocUncached));
    else CK(hipMalloc((void **)&buf, bytes));
    CK(hipMalloc((void **)&loc, 8 * 4096));
    u64 *sync; unsigned *claim, *out;
    CK(hipExtMallocWithFlags((void **)&sync, 4096, hipDeviceMallocUncached));
    CK(hipMalloc((void **)&claim, 64));
    CK(hipMalloc((void **)&out, (size_t)256 * OUTW * sizeof(unsigned)));
    struct Cfg { int mode; unsigned sstride, nrep, rspace; const char *what; unsigned window; };
    const Cfg cfgs[] = {
        {0, 16, 8, 65536, "library layout, polling all the time", 0},
        {0, 16, 8, 65536, "library layout, first pass 1 cycle after the publish, none after completion", 1},
        {0, 16, 8, 65536, "window 500 cycles", 500},
        {0, 16, 8, 65536, "window 1000 cycles", 1000},
        {0, 16, 8, 65536, "window 1500 cycles", 1500},
        {0, 16, 8, 65536, "window 1900 cycles (the update sweep's at C3)", 1900},
        {0, 16, 8, 65536, "window 2400 cycles", 2400},
        {0, 16, 8, 65536, "window 3000 cycles", 3000},
        {0, 16, 1, 65536, "ONE copy, window 1900", 1900},
        {3, 16, 8, 65536, "own XCD through L2, window 1900", 1900},
    };
    for (const Cfg &cf : cfgs) {
        const int mode = cf.mode;
        CK(hipMemset(buf, 0, bytes)); CK(hipMemset(loc, 0, 8 * 4096)); CK(hipMemset(sync, 0, 4096));
        CK(hipMemset(claim, 0, 64)); CK(hipMemset(out, 0, (size_t)256 * OUTW * sizeof(unsigned)));
        hipLaunchKernelGGL(pollmix, dim3(256), dim3(128), 0, 0, buf, loc, mode, nsteps, period, sync, claim, out, cf.sstride, cf.nrep, cf.rspace, cf.window);
        CK(hipDeviceSynchronize());
        std::vector<unsigned> h((size_t)256 * OUTW);
        CK(hipMemcpy(h.data(), out, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
        int present = 0; double cmean = 0; unsigned cmax = 0, cmin = ~0u; u64 steps_seen = 0;
        std::vector<double> bs(NBIN, 0.0), bc(NBIN, 0.0);
        std::vector<unsigned> cm;
        for (int w = 0; w < 256; ++w) {
            const unsigned *o = &h[(size_t)w * OUTW];
            if (!o[2 * NBIN + 3]) continue;
            ++present;
            cmean += o[2 * NBIN]; cmax = std::max(cmax, o[2 * NBIN + 1]); cmin = std::min(cmin, o[2 * NBIN]); steps_seen += o[2 * NBIN + 2];
            cm.push_back(o[2 * NBIN]);
            for (int b = 0; b < NBIN; ++b) { bs[b] += o[b]; bc[b] += o[NBIN + b]; }
        }
        std::sort(cm.begin(), cm.end());
        printf("%s\n   mode %d (%s memory, %d workgroups took part, period %u0 ns): publish -> first complete pass: mean over workgroups %.0f cycles (min %u, median %u, max of means %u), "
               "worst single step %u; steps completed per workgroup %.1f of %d\n", cf.what, mode, mem ? "uncached" : "hipMalloc", present, period,
               present ? cmean / present : 0.0, cmin, cm.empty() ? 0u : cm[cm.size() / 2], cm.empty() ? 0u : cm.back(), cmax, present ? (double)steps_seen / present : 0.0, nsteps);
        printf("   pass duration by time since own publish (256-cycle bins; cycles, passes):");
        for (int b = 0; b < 24; ++b) printf(" %d:%.0f(%.0f)", b, bc[b] > 0 ? bs[b] / bc[b] : 0.0, bc[b] / std::max(present, 1));
        printf(" ... last:%.0f(%.0f)\n", bc[NBIN - 1] > 0 ? bs[NBIN - 1] / bc[NBIN - 1] : 0.0, bc[NBIN - 1] / std::max(present, 1));
    }
    return 0;
}
