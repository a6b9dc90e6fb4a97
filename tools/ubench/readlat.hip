// Round trip of one poll pass (four 16-byte loads per lane, 4 KB, as kr_poll_pass at P = 256) by XCD and by the ADDRESS of the
// 4-KB piece inside an uncached allocation: does it matter where a mailbox replica lies?
//   ./readlat [STRIDE_BYTES=4096] [PIECES=256] [REPS=64]
// Output: per XCD the min / median / max over the pieces of the per-piece MINIMUM latency (cycles), then the table.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned xcc_id() {
    unsigned id;
    asm("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    return id & 7u;
}
__device__ __forceinline__ unsigned long long rt() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
    return t;
}

__global__ void __launch_bounds__(64) readlat(const unsigned char *buf, const size_t stride, const int pieces, const int reps,
                                              unsigned *claim, unsigned *out /*[8][pieces][2]*/, unsigned *sink) {
    const unsigned x = xcc_id();
    unsigned first = 0;
    if (threadIdx.x == 0) first = atomicAdd(&claim[x], 1u);
    first = __builtin_amdgcn_readfirstlane(first);
    if (first != 0u) return;   // one wave per XCD measures
    const unsigned voff = threadIdx.x * 16u;
    unsigned acc = 0;
    for (int p = 0; p < pieces; ++p) {
        const unsigned char *sb = buf + (size_t)p * stride;
        unsigned long long best = ~0ull, sum = 0;
        for (int r = 0; r < reps; ++r) {
            v4u g0, g1, g2, g3;
            const unsigned long long t0 = rt();
            asm volatile("global_load_dwordx4 %0, %4, %5 sc1\n\tglobal_load_dwordx4 %1, %4, %5 offset:1024 sc1\n\t"
                         "global_load_dwordx4 %2, %4, %5 offset:2048 sc1\n\tglobal_load_dwordx4 %3, %4, %5 offset:3072 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3) : "v"(voff), "s"(sb) : "memory");
            const unsigned long long t1 = rt();
            acc += g0.x + g1.x + g2.x + g3.x;
            const unsigned long long d = t1 - t0;
            best = d < best ? d : best;
            sum += d;
        }
        if (threadIdx.x == 0) {
            out[((size_t)x * pieces + p) * 2 + 0] = (unsigned)best;
            out[((size_t)x * pieces + p) * 2 + 1] = (unsigned)(sum / (unsigned long long)reps);
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}

// CONTENTION: `npoll` waves per XCD (one per workgroup, as in the update sweep) repeat the pass on their XCD's 4-KB piece (pieces 64 KB
// apart); mean cycles per pass of each poller.  With `wr` != 0 every poller also stores 16 bytes into each of the 8 pieces per pass
// (the publish of kr_exchange).
__global__ void __launch_bounds__(64) contend(unsigned char *buf, const int npoll, const int reps, const int wr, unsigned *claim,
                                              unsigned *out /*[8][32]*/, unsigned *sink) {
    const unsigned x = xcc_id();
    unsigned rank = 0;
    if (threadIdx.x == 0) rank = atomicAdd(&claim[x], 1u);
    rank = __builtin_amdgcn_readfirstlane(rank);
    if ((int)rank >= npoll) return;
    const unsigned voff = threadIdx.x * 16u;
    const unsigned char *sb = buf + (size_t)x * 65536;
    unsigned acc = 0;
    const unsigned long long t0 = rt();
    for (int r = 0; r < reps; ++r) {
        if (wr && threadIdx.x < 8) {
            unsigned char *dst = buf + (size_t)threadIdx.x * 65536 + (size_t)(x * 32 + rank) * 16;
            const unsigned long long val = (unsigned long long)r;
            asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(dst), "v"(val) : "memory");
        }
        v4u g0, g1, g2, g3;
        asm volatile("global_load_dwordx4 %0, %4, %5 sc1\n\tglobal_load_dwordx4 %1, %4, %5 offset:1024 sc1\n\t"
                     "global_load_dwordx4 %2, %4, %5 offset:2048 sc1\n\tglobal_load_dwordx4 %3, %4, %5 offset:3072 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3) : "v"(voff), "s"(sb) : "memory");
        acc += g0.x + g1.x + g2.x + g3.x;
    }
    const unsigned long long t1 = rt();
    if (threadIdx.x == 0) out[x * 32 + rank] = (unsigned)((t1 - t0) / (unsigned long long)reps);
    if (acc == 0x12345678u) *sink = acc;
}

int main(int argc, char **argv) {
    const size_t stride = argc > 1 ? (size_t)atoll(argv[1]) : 4096;
    const int pieces = argc > 2 ? atoi(argv[2]) : 256;
    const int reps = argc > 3 ? atoi(argv[3]) : 64;
    const int mem = argc > 4 ? atoi(argv[4]) : 1;   // 1 = uncached (the library's mailbox), 0 = hipMalloc
    const size_t bytes = std::max(stride * (size_t)pieces + 8192, (size_t)8 * 65536 + 8192);
    unsigned char *buf = nullptr;
    if (mem) CK(hipExtMallocWithFlags((void **)&buf, bytes, hipDeviceMallocUncached));
    else CK(hipMalloc((void **)&buf, bytes));
    CK(hipMemset(buf, 0, bytes));
    unsigned *claim, *out, *sink;
    CK(hipMalloc((void **)&claim, 8 * sizeof(unsigned)));
    CK(hipMalloc((void **)&out, (size_t)8 * pieces * 2 * sizeof(unsigned)));
    CK(hipMalloc((void **)&sink, sizeof(unsigned)));
    CK(hipMemset(claim, 0, 8 * sizeof(unsigned)));
    CK(hipMemset(out, 0, (size_t)8 * pieces * 2 * sizeof(unsigned)));
    if (argc > 5) {
        const int wr = argc > 6 ? atoi(argv[6]) : 0;
        printf("contention, %s memory, %s: mean cycles per pass of 4 KB (min / max over the pollers of all XCDs)\n", mem ? "uncached" : "hipMalloc", wr ? "with publishing stores" : "loads only");
        for (int npoll = 1; npoll <= 32; npoll *= 2) {
            CK(hipMemset(claim, 0, 8 * sizeof(unsigned)));
            CK(hipMemset(out, 0, 8 * 32 * sizeof(unsigned)));
            hipLaunchKernelGGL(contend, dim3(256), dim3(64), 0, 0, buf, npoll, 4000, wr, claim, out, sink);
            CK(hipDeviceSynchronize());
            unsigned h[256];
            CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
            unsigned lo = ~0u, hi = 0; double mean = 0; int cnt = 0;
            for (int i = 0; i < 256; ++i) if (h[i]) { lo = std::min(lo, h[i]); hi = std::max(hi, h[i]); mean += h[i]; ++cnt; }
            printf("  %2d pollers per XCD (%3d in all): %5u / %5u, mean %6.0f\n", npoll, cnt, lo, hi, mean / cnt);
        }
        return 0;
    }
    hipLaunchKernelGGL(readlat, dim3(256), dim3(64), 0, 0, buf, stride, pieces, reps, claim, out, sink);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> h((size_t)8 * pieces * 2);
    CK(hipMemcpy(h.data(), out, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    const double ns = 1.0;   // s_memtime: shader clock cycles
    printf("stride %zu B, %d pieces, %d reps, %s memory; shader-clock cycles per pass of 4 KB\n", stride, pieces, reps, mem ? "uncached" : "hipMalloc");
    for (int x = 0; x < 8; ++x) {
        std::vector<unsigned> v(pieces);
        for (int p = 0; p < pieces; ++p) v[p] = h[((size_t)x * pieces + p) * 2];
        std::vector<unsigned> s = v;
        std::sort(s.begin(), s.end());
        double mean = 0;
        for (int p = 0; p < pieces; ++p) mean += h[((size_t)x * pieces + p) * 2 + 1];
        printf("XCD %d: best-of-reps per piece: min %4.0f  p10 %4.0f  median %4.0f  p90 %4.0f  max %4.0f   mean of means %5.0f\n", x, s[0] * ns, s[pieces / 10] * ns,
               s[pieces / 2] * ns, s[pieces * 9 / 10] * ns, s[pieces - 1] * ns, mean / pieces * ns);
    }
    printf("table (rows = piece, columns = XCD 0..7, best-of-reps in cycles):\n");
    for (int p = 0; p < pieces && p < 64; ++p) {
        printf("%4d:", p);
        for (int x = 0; x < 8; ++x) printf(" %5.0f", h[((size_t)x * pieces + p) * 2] * ns);
        printf("\n");
    }
    return 0;
}
