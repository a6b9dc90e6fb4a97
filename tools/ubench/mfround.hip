// One Horner round of the matrix-core sweep (kr_device_mfma.h) in isolation: cycles per round for TILES = 1, 2 and for
// the multi-matrix product of phase A (3 matrices sharing the B operands).  Build:
//   hipcc --offload-arch=gfx950 -O3 -I krotov.jl_amd/csrc -o tools/ubench/mfround tools/ubench/mfround.hip
#include "kr_device_mfma.h"
#include <cstdio>

template <int TILES, int MODE>
__global__ __launch_bounds__(256, 1) void round_kernel(const double *Hin, double *out, unsigned long long *cyc, int iters) {
    constexpr int TW = KR_MF_TW;
    __shared__ __attribute__((aligned(16))) double2 sy[2][TILES][TW];
    __shared__ __attribute__((aligned(16))) double2 sH0[16 * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f0 = lane & 3, blk = (lane >> 2) & 3, f2 = lane >> 4;
    const int drow = 16 * wave + 4 * blk + f2, boff = f2 * 4 + f0, doff = drow * 4 + f0, zoff = drow * 4 + (f0 ^ 1);
    const double zs = (f0 & 1) ? 1.0 : -1.0;
    double Er[2][16], Ei[2][16];
    for (int kb = 0; kb < 16; ++kb) {
        for (int m = 0; m < 2; ++m) { Er[m][kb] = Hin[(m * 32 + kb) * 256 + tid] * 1e-3; Ei[m][kb] = Hin[(m * 32 + 16 + kb) * 256 + tid] * 1e-3; }
        sH0[kb * 256 + tid] = make_double2(Hin[kb * 256 + tid] * 1e-3, Hin[(16 + kb) * 256 + tid] * 1e-3);
    }
    double v[TILES], y[TILES];
    for (int t = 0; t < TILES; ++t) { v[t] = 1.0 + 1e-3 * tid; y[t] = v[t]; kr_mf_put(sy[0][t], doff, zoff, zs, v[t]); }
    __syncthreads();
    int wb = 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 7) {   // Taylor rounds as the kernel runs them: own rows first, barrier inside the apply
        for (int it = 0; it < iters; ++it) {
            double d[TILES];
            kr_mf_apply<TILES, true>(Er[0], Ei[0], &sy[wb ^ 1][0][0], boff, 4 * wave, d);
            for (int t = 0; t < TILES; ++t) { y[t] = fma(0.125, d[t], v[t]); kr_mf_put(sy[wb][t], doff, zoff, zs, y[t]); }
            wb ^= 1;
        }
    } else
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // Taylor round: one matrix from registers, lane-local update, LDS write, barrier
            double d[TILES];
            kr_mf_apply<TILES>(Er[0], Ei[0], &sy[wb ^ 1][0][0], boff, 4 * wave, d);
            for (int t = 0; t < TILES; ++t) { y[t] = fma(0.125, d[t], v[t]); kr_mf_put(sy[wb][t], doff, zoff, zs, y[t]); }
            __syncthreads();
            wb ^= 1;
        } else if (MODE >= 3) {   // variants of the Taylor round: 3 = no barrier, 4 = no LDS write, 5 = neither, 6 = raw s_barrier
            double d[TILES];
            kr_mf_apply<TILES>(Er[0], Ei[0], &sy[wb ^ 1][0][0], boff, 4 * wave, d);
            for (int t = 0; t < TILES; ++t) {
                y[t] = fma(0.125, d[t], v[t]);
                if (MODE == 3 || MODE == 6) { kr_mf_put(sy[wb][t], doff, zoff, zs, y[t]); }
                else v[t] += 1e-9 * y[t];
            }
            if (MODE == 4) __syncthreads();
            if (MODE == 6) { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier(); }
            if (MODE == 3 || MODE == 6) wb ^= 1;
        } else if (MODE == 1) {   // phase A: 3 matrices (drift from LDS), no LDS write
            double D[3][TILES];
            kr_mf_multi<TILES, 3, true, false>(sH0, Er, Ei, &sy[0][0][0], boff, 4 * wave, D);
            for (int t = 0; t < TILES; ++t) y[t] += D[0][t] + D[1][t] + D[2][t];
        } else if (MODE == 8 || MODE == 9 || MODE == 10) {   // bare MFMA stream + barrier (8), + LDS write before it (9), + dependent VALU only (10)
            double p0 = 0, p1 = 0, p2 = 0, p3 = 0;
            for (int kb = 0; kb < 16; ++kb) {
                p0 = kr_mfma4(Er[0][kb], y[0], p0); p1 = kr_mfma4(Ei[0][kb], v[0], p1);
                if (TILES == 2) { p2 = kr_mfma4(Er[0][kb], v[0], p2); p3 = kr_mfma4(Ei[0][kb], y[0], p3); }
            }
            y[0] = 1e-9 * (p0 + p1 + p2 + p3) + v[0];
            if (MODE == 9) { kr_mf_put(sy[wb][0], doff, zoff, zs, y[0]); wb ^= 1; }
            if (MODE != 10) __syncthreads();
        } else {                  // bare MFMA stream of the same length as a Taylor round (no LDS): the floor
            double p0 = 0, p1 = 0, p2 = 0, p3 = 0;
            for (int kb = 0; kb < 16; ++kb) {
                p0 = kr_mfma4(Er[0][kb], y[0], p0); p1 = kr_mfma4(Ei[0][kb], v[0], p1);
                if (TILES == 2) { p2 = kr_mfma4(Er[0][kb], v[0], p2); p3 = kr_mfma4(Ei[0][kb], y[0], p3); }
            }
            y[0] = 1e-9 * (p0 + p1 + p2 + p3) + v[0];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int t = 0; t < TILES; ++t) s += y[t];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    double *H, *out; unsigned long long *cyc, h;
    (void)hipMalloc(&H, 64 * 256 * 8); (void)hipMalloc(&out, 256 * 256 * 8); (void)hipMalloc(&cyc, 64);
    double *hH = new double[64 * 256];
    for (int i = 0; i < 64 * 256; ++i) hH[i] = (i * 7919 % 1000) / 1000.0 - 0.5;
    (void)hipMemcpy(H, hH, 64 * 256 * 8, hipMemcpyHostToDevice);
    const int iters = 2000;
#define RUN(T, M, name) hipLaunchKernelGGL((round_kernel<T, M>), dim3(256), dim3(256), 0, 0, H, out, cyc, iters); (void)hipDeviceSynchronize(); \
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("TILES=%d %-34s: %7.1f cycles per iteration\n", T, name, (double)h / iters);
    RUN(1, 2, "bare MFMA stream (32)") RUN(2, 2, "bare MFMA stream (64)")
    RUN(1, 0, "Taylor round") RUN(2, 0, "Taylor round")
    RUN(1, 1, "phase A, 3 matrices") RUN(2, 1, "phase A, 3 matrices")
    RUN(1, 3, "Taylor round, no barrier") RUN(2, 3, "Taylor round, no barrier")
    RUN(1, 4, "Taylor round, no LDS write") RUN(2, 4, "Taylor round, no LDS write")
    RUN(1, 5, "Taylor round, neither") RUN(2, 5, "Taylor round, neither")
    RUN(1, 6, "Taylor round, raw s_barrier") RUN(2, 6, "Taylor round, raw s_barrier")
    RUN(1, 7, "Taylor round, own rows first") RUN(2, 7, "Taylor round, own rows first")
    RUN(1, 8, "bare MFMA (32) + barrier") RUN(1, 9, "bare MFMA (32) + LDS write + barrier") RUN(2, 8, "bare MFMA (64) + barrier") RUN(2, 9, "bare MFMA (64) + LDS write + barrier")
    return 0;
}
