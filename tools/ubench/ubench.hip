// Micro-benchmarks that calibrate the per-round cost model of the sweep kernels on gfx950:
//  (1) cycles per v_fma_f64 (wave64) with W waves per SIMD and C independent chains
//  (2) LDS write -> barrier -> LDS read round trip with W waves per workgroup
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CH>
__global__ void fma_chain(double *out, unsigned long long *cyc, int iters) {
    double a[CH];
    const double x = 1.0000001 + threadIdx.x * 1e-9, y = 1e-9;
#pragma unroll
    for (int c = 0; c < CH; ++c) a[c] = c + threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int c = 0; c < CH; ++c) a[c] = fma(a[c], x, y);
        }
    }
    __syncthreads();  // all waves of the workgroup done (wave 0 is the oldest and wins arbitration otherwise)
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += a[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void lds_roundtrip(double *out, unsigned long long *cyc, int iters) {
    __shared__ double2 buf[2][1024];
    const int tid = threadIdx.x;
    double2 v = make_double2(tid, 1.0);
    buf[0][tid] = v;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int w = 0;
    for (int it = 0; it < iters; ++it) {
        const double2 r = buf[w][(tid * 7 + 3) % blockDim.x];  // read someone else's value
        v.x += r.x * 1e-9;
        buf[w ^ 1][tid] = v;
        __syncthreads();
        w ^= 1;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + tid] = v.x;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void barrier_only(unsigned long long *cyc, int iters) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    double *out;
    unsigned long long *cyc, h;
    (void)hipMalloc(&out, 1 << 24);
    (void)hipMalloc(&cyc, 64);
    const int iters = 2000;
    for (int threads : {64, 256, 512, 1024}) {
        auto run = [&](auto kern, int ch) {
            hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            printf("fma_f64 threads/WG=%4d (waves/SIMD=%.2f) chains=%d : %.2f cycles per wave-FMA (per wave), %.2f cycles per FMA per SIMD\n",
                   threads, threads / 256.0, ch, (double)h / (iters * 16.0 * ch),
                   (double)h / (iters * 16.0 * ch) / (threads >= 256 ? threads / 256.0 : 1.0));
        };
        run(fma_chain<1>, 1);
        run(fma_chain<4>, 4);
        run(fma_chain<8>, 8);
    }
    for (int threads : {64, 256, 512, 1024}) {
        hipLaunchKernelGGL(lds_roundtrip, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("lds write->barrier->read threads=%4d : %.1f cycles per round\n", threads, (double)h / iters);
        hipLaunchKernelGGL(barrier_only, dim3(256), dim3(threads), 0, 0, cyc, iters);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("barrier only               threads=%4d : %.1f cycles per barrier\n", threads, (double)h / iters);
    }
    return 0;
}
