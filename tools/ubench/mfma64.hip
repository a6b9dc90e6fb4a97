// Calibrates v_mfma_f64_16x16x4_f64 on gfx950: cycles per instruction (dependent / independent accumulators,
// 1..4 waves per SIMD) and the fragment layout (A = I check with an asymmetric B).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void mfma_rate(double *out, unsigned long long *cyc, int iters) {
    d4 acc[NACC];
#pragma unroll
    for (int c = 0; c < NACC; ++c) acc[c] = d4{0, 0, 0, 0};
    const double a = 1.0 + threadIdx.x * 1e-6, b = 1.0 - threadIdx.x * 1e-6;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < NACC; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int c = 0; c < NACC; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// layout probe: D = A*B with A[i][k], B[k][j] given by functions; every lane writes what it holds
__global__ void mfma_layout(double *D) {
    const int l = threadIdx.x;
    // A (16x4): lane l holds A[i = l & 15][k = l >> 4];  B (4x16): lane l holds B[k = l >> 4][j = l & 15]
    const int i = l & 15, k = l >> 4, j = l & 15;
    const double a = (i == k) ? 1.0 : 0.0;              // A = [I_4 ; 0]  -> D[i][j] = B[i][j] for i < 4
    const double b = 100.0 * k + j;                      // asymmetric B
    d4 acc = d4{0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = acc[r];
}

int main() {
    double *out; unsigned long long *cyc, h;
    (void)hipMalloc(&out, 1 << 22); (void)hipMalloc(&cyc, 64);
    const int iters = 1000;
    for (int threads : {64, 256, 512, 1024}) {
#define RUN(N) hipLaunchKernelGGL(mfma_rate<N>, dim3(256), dim3(threads), 0, 0, out, cyc, iters); (void)hipDeviceSynchronize(); \
        (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); \
        printf("mfma_f64_16x16x4 threads=%4d acc=%d : %.1f cycles per MFMA per wave, %.1f per SIMD\n", threads, N, \
               (double)h / (iters * 8.0 * N), (double)h / (iters * 8.0 * N) / (threads >= 256 ? threads / 256.0 : 1.0));
        RUN(1) RUN(4)
    }
    double *D, hD[256];
    (void)hipMalloc(&D, sizeof hD);
    hipLaunchKernelGGL(mfma_layout, dim3(1), dim3(64), 0, 0, D);
    (void)hipMemcpy(hD, D, sizeof hD, hipMemcpyDeviceToHost);
    // expected D[i][j] = B[i][j] = 100 i + j for i < 4, 0 otherwise.  Report which (row, col) each (lane, reg) holds.
    printf("layout: lane reg -> value (decode row = value/100, col = value%%100 for rows < 4)\n");
    for (int l : {0, 1, 15, 16, 17, 32, 48, 63}) printf("  lane %2d: %6.0f %6.0f %6.0f %6.0f\n", l, hD[l * 4], hD[l * 4 + 1], hD[l * 4 + 2], hD[l * 4 + 3]);
    return 0;
}
