// Calibrates v_mfma_f64_4x4x4_4b_f64 on gfx950 (4 blocks of 4x4x4 per instruction, one f64 of A, B and D per lane):
// cycles per instruction (dependent / independent accumulators, 1..2 waves per SIMD) and the lane maps of A, B, D,
// found by testing every assignment of the two 2-bit fields of (lane & 15) against a host product on random data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>

template <int NACC>
__global__ void mfma_rate(double *out, unsigned long long *cyc, int iters) {
    double acc[NACC];
#pragma unroll
    for (int c = 0; c < NACC; ++c) acc[c] = 0.0;
    const double a = 1.0 + threadIdx.x * 1e-6, b = 1.0 - threadIdx.x * 1e-6;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < NACC; ++c) acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[c], 0, 0, 0);
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int c = 0; c < NACC; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// fp64 FMA reference: NCH independent chains
template <int NCH>
__global__ void fma_rate(double *out, unsigned long long *cyc, int iters) {
    double acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = threadIdx.x * 1e-9 + c;
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[c] = fma(acc[c], a, b);
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void mfma_once(const double *a, const double *b, double *d) {
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}

int main() {
    double *out; unsigned long long *cyc, h;
    (void)hipMalloc(&out, 1 << 22); (void)hipMalloc(&cyc, 64);
    const int iters = 1000;
    for (int threads : {64, 256, 512}) {
#define RUN(N) hipLaunchKernelGGL(mfma_rate<N>, dim3(256), dim3(threads), 0, 0, out, cyc, iters); (void)hipDeviceSynchronize(); \
        (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); \
        printf("mfma_f64_4x4x4_4b threads=%4d acc=%d : %.1f cycles per MFMA per wave, %.1f per SIMD\n", threads, N, \
               (double)h / (iters * 8.0 * N), (double)h / (iters * 8.0 * N) / (threads >= 256 ? threads / 256.0 : 1.0));
        RUN(1) RUN(2) RUN(4) RUN(8)
#undef RUN
#define RUN(N) hipLaunchKernelGGL(fma_rate<N>, dim3(256), dim3(threads), 0, 0, out, cyc, iters); (void)hipDeviceSynchronize(); \
        (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); \
        printf("v_fma_f64 threads=%4d chains=%d : %.1f cycles per FMA per wave, %.1f per SIMD\n", threads, N, \
               (double)h / (iters * 8.0 * N), (double)h / (iters * 8.0 * N) / (threads >= 256 ? threads / 256.0 : 1.0));
        RUN(1) RUN(4) RUN(8)
#undef RUN
    }
    // layout: the three 2-bit fields of the lane, f0 = l & 3, f1 = (l >> 2) & 3, f2 = l >> 4, are assigned to the roles
    // (i, k, blk) of A, (k, j, blk) of B and (i, j, blk) of D in every possible way (6^3 hypotheses) and compared with a
    // host product on random data.
    double ha[64], hb[64], hd[64], *da, *db, *dd;
    srand(7);
    for (int l = 0; l < 64; ++l) { ha[l] = rand() / (double)RAND_MAX - 0.5; hb[l] = rand() / (double)RAND_MAX - 0.5; }
    (void)hipMalloc(&da, 512); (void)hipMalloc(&db, 512); (void)hipMalloc(&dd, 512);
    (void)hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mfma_once, dim3(1), dim3(64), 0, 0, da, db, dd);
    (void)hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
    static const int perm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    int found = 0;
    for (int pa = 0; pa < 6; ++pa) for (int pb = 0; pb < 6; ++pb) for (int pd = 0; pd < 6; ++pd) {
        double A[4][4][4] = {}, B[4][4][4] = {};   // [blk][i][k], [blk][k][j]
        for (int l = 0; l < 64; ++l) {
            const int f[3] = {l & 3, (l >> 2) & 3, l >> 4};
            A[f[perm[pa][2]]][f[perm[pa][0]]][f[perm[pa][1]]] = ha[l];   // roles (i, k, blk) <- fields perm[pa][0..2]
            B[f[perm[pb][2]]][f[perm[pb][0]]][f[perm[pb][1]]] = hb[l];   // roles (k, j, blk)
        }
        double err = 0;
        for (int l = 0; l < 64; ++l) {
            const int f[3] = {l & 3, (l >> 2) & 3, l >> 4};
            const int i = f[perm[pd][0]], j = f[perm[pd][1]], blk = f[perm[pd][2]];
            double sum = 0;
            for (int k = 0; k < 4; ++k) sum += A[blk][i][k] * B[blk][k][j];
            err = fmax(err, fabs(sum - hd[l]));
        }
        if (err < 1e-14) {
            ++found;
            printf("MATCH: A (i,k,blk) = fields (%d,%d,%d); B (k,j,blk) = fields (%d,%d,%d); D (i,j,blk) = fields (%d,%d,%d)   [field 0 = l&3, 1 = (l>>2)&3, 2 = l>>4]  err %.1e\n",
                   perm[pa][0], perm[pa][1], perm[pa][2], perm[pb][0], perm[pb][1], perm[pb][2], perm[pd][0], perm[pd][1], perm[pd][2], err);
        }
    }
    if (!found) {
        printf("no field hypothesis matches; raw indicator probe follows (a = [l==p], b = 1000+l): D per lane\n");
        for (int pp = 0; pp < 64; pp += 1) {
            for (int l = 0; l < 64; ++l) { ha[l] = (l == pp) ? 1.0 : 0.0; hb[l] = 1000.0 + l; }
            (void)hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(mfma_once, dim3(1), dim3(64), 0, 0, da, db, dd);
            (void)hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
            printf("p=%2d:", pp);
            for (int l = 0; l < 64; ++l) if (hd[l] != 0.0) printf(" D[%d]=B%d", l, (int)(hd[l] - 1000.0));
            printf("\n");
        }
    }
    return 0;
}
