// issue rate of fp64 FMA forms on gfx950, 256 workgroups x 4 waves (chip busy: clocks up), 8 independent chains, distinct operands
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned long long mt() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }
template <int MODE>
__global__ void __launch_bounds__(256) rate(double *out, unsigned *cyc, int iters) {
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    double x0 = 1.0000001, x1 = 1.0000002, x2 = 0.9999, x3 = 0.99999, h0 = 0.9999999, h1 = 1.00001, h2 = 0.999, h3 = 1.0001;
    const unsigned long long t0 = mt();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {   // v_fmac_f64_e32
            asm volatile("v_fmac_f64_e32 %0, %8, %12\n\tv_fmac_f64_e32 %1, %9, %13\n\tv_fmac_f64_e32 %2, %10, %14\n\tv_fmac_f64_e32 %3, %11, %15\n\t"
                         "v_fmac_f64_e32 %4, %8, %13\n\tv_fmac_f64_e32 %5, %9, %14\n\tv_fmac_f64_e32 %6, %10, %15\n\tv_fmac_f64_e32 %7, %11, %12"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(h0), "v"(h1), "v"(h2), "v"(h3));
        } else if (MODE == 1) {   // v_fmac_f64_dpp row_newbcast
#define D(o, x, h, j) "v_fmac_f64_dpp %" #o ", %" #x ", %" #h " row_newbcast:" #j " row_mask:0xf bank_mask:0xf\n\t"
            asm volatile(D(0, 8, 12, 0) D(1, 9, 13, 1) D(2, 10, 14, 2) D(3, 11, 15, 3) D(4, 8, 13, 4) D(5, 9, 14, 5) D(6, 10, 15, 6) D(7, 11, 12, 7)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(h0), "v"(h1), "v"(h2), "v"(h3));
        } else {   // v_fma_f64 (VOP3)
            asm volatile("v_fma_f64 %0, %8, %12, %0\n\tv_fma_f64 %1, %9, %13, %1\n\tv_fma_f64 %2, %10, %14, %2\n\tv_fma_f64 %3, %11, %15, %3\n\t"
                         "v_fma_f64 %4, %8, %13, %4\n\tv_fma_f64 %5, %9, %14, %5\n\tv_fma_f64 %6, %10, %15, %6\n\tv_fma_f64 %7, %11, %12, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(h0), "v"(h1), "v"(h2), "v"(h3));
        }
    }
    const unsigned long long t1 = mt();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = (unsigned)(t1 - t0);
}
int main() {
    double *out; unsigned *cyc; unsigned h[256];
    hipMalloc(&out, 256 * 256 * 8); hipMalloc(&cyc, 1024);
    const int iters = 200000;
    for (int threads = 64; threads <= 256; threads *= 2) {
        for (int mode = 0; mode < 3; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
            else if (mode == 1) hipLaunchKernelGGL(rate<1>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
            else hipLaunchKernelGGL(rate<2>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, cyc, 1024, hipMemcpyDeviceToHost);
            printf("%3d threads per workgroup, %s: %.2f s_memtime ticks per wave-FMA; %.3f ns per wave-FMA per wave (events)\n", threads,
                   mode == 0 ? "v_fmac_f64_e32      " : mode == 1 ? "v_fmac_f64_dpp bcast" : "v_fma_f64 (VOP3)    ", h[0] / (8.0 * iters), ms * 1e6 / (8.0 * iters));
        }
    }
    return 0;
}
