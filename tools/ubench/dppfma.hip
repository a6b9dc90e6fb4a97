// v_fmac_f64_dpp ... row_newbcast:n on gfx950 (DP-ALU DPP: the source operand comes from lane n of the 16-lane row): rate and
// semantics.  A matvec with one row per LANE (N <= 16, one trajectory per DPP row) needs no LDS and no reduction if this runs at
// the plain FMA rate.   ./dppfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long mt() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }

// semantics: out[lane] = sum_j x[row*16 + j] * h[lane][j]
__global__ void __launch_bounds__(64) check(const double *x, const double *h, double *out) {
    const int lane = threadIdx.x;
    const double xv = x[lane];
    double hh[16];
    for (int j = 0; j < 16; ++j) hh[j] = h[lane * 16 + j];
    double acc = 0.0;
#define STEP(j) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #j " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(xv), "v"(hh[j]));
    STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7) STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15)
    out[lane] = acc;
}

template <int CHAINS, int DPP>
__global__ void __launch_bounds__(256) rate(double *out, unsigned *cyc, int iters, int waves) {
    if ((int)(threadIdx.x >> 6) >= waves) return;
    double a0 = threadIdx.x, a1 = 1.0, a2 = 2.0, a3 = 3.0;
    const double x = 1.0000001, h = 0.9999999;
    const unsigned long long t0 = mt();
    for (int i = 0; i < iters; ++i) {
#define F_DPP(acc, n) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #n " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(h));
#define F_PL(acc) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc) : "v"(x), "v"(h));
        if (DPP) {
            if (CHAINS == 1) { F_DPP(a0, 0) F_DPP(a0, 1) F_DPP(a0, 2) F_DPP(a0, 3) F_DPP(a0, 4) F_DPP(a0, 5) F_DPP(a0, 6) F_DPP(a0, 7) }
            else if (CHAINS == 2) { F_DPP(a0, 0) F_DPP(a1, 1) F_DPP(a0, 2) F_DPP(a1, 3) F_DPP(a0, 4) F_DPP(a1, 5) F_DPP(a0, 6) F_DPP(a1, 7) }
            else { F_DPP(a0, 0) F_DPP(a1, 1) F_DPP(a2, 2) F_DPP(a3, 3) F_DPP(a0, 4) F_DPP(a1, 5) F_DPP(a2, 6) F_DPP(a3, 7) }
        } else {
            if (CHAINS == 1) { F_PL(a0) F_PL(a0) F_PL(a0) F_PL(a0) F_PL(a0) F_PL(a0) F_PL(a0) F_PL(a0) }
            else if (CHAINS == 2) { F_PL(a0) F_PL(a1) F_PL(a0) F_PL(a1) F_PL(a0) F_PL(a1) F_PL(a0) F_PL(a1) }
            else { F_PL(a0) F_PL(a1) F_PL(a2) F_PL(a3) F_PL(a0) F_PL(a1) F_PL(a2) F_PL(a3) }
        }
    }
    const unsigned long long t1 = mt();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
    if (threadIdx.x == 0) cyc[blockIdx.x] = (unsigned)(t1 - t0);
}

int main() {
    double hx[64], hh[64 * 16], ho[64];
    for (int i = 0; i < 64; ++i) hx[i] = 0.5 + 0.01 * i;
    for (int i = 0; i < 64 * 16; ++i) hh[i] = 1.0 + 0.001 * (i % 97);
    double *dx, *dh, *dout; unsigned *dc;
    CK(hipMalloc(&dx, sizeof hx)); CK(hipMalloc(&dh, sizeof hh)); CK(hipMalloc(&dout, 256 * 8 * sizeof(double))); CK(hipMalloc(&dc, 64));
    CK(hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice)); CK(hipMemcpy(dh, hh, sizeof hh, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, dx, dh, dout);
    CK(hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int l = 0; l < 64; ++l) {
        double ref = 0;
        for (int j = 0; j < 16; ++j) ref += hx[(l / 16) * 16 + j] * hh[l * 16 + j];
        worst = fmax(worst, fabs(ref - ho[l]));
    }
    printf("semantics: out[lane] = sum_j x[16 (lane / 16) + j] h[lane][j], max |error| %.3e\n", worst);
    const int iters = 400000;
    for (int waves = 1; waves <= 4; waves *= 2) {
        for (int dpp = 0; dpp < 2; ++dpp) {
            unsigned c[3];
            hipLaunchKernelGGL((rate<1, 0>), dim3(1), dim3(256), 0, 0, dout, dc, 10, waves);   // warm-up
            if (dpp) { hipLaunchKernelGGL((rate<1, 1>), dim3(1), dim3(256), 0, 0, dout, dc, iters, waves); } else { hipLaunchKernelGGL((rate<1, 0>), dim3(1), dim3(256), 0, 0, dout, dc, iters, waves); }
            CK(hipMemcpy(&c[0], dc, 4, hipMemcpyDeviceToHost));
            if (dpp) { hipLaunchKernelGGL((rate<2, 1>), dim3(1), dim3(256), 0, 0, dout, dc, iters, waves); } else { hipLaunchKernelGGL((rate<2, 0>), dim3(1), dim3(256), 0, 0, dout, dc, iters, waves); }
            CK(hipMemcpy(&c[1], dc, 4, hipMemcpyDeviceToHost));
            if (dpp) { hipLaunchKernelGGL((rate<4, 1>), dim3(1), dim3(256), 0, 0, dout, dc, iters, waves); } else { hipLaunchKernelGGL((rate<4, 0>), dim3(1), dim3(256), 0, 0, dout, dc, iters, waves); }
            CK(hipMemcpy(&c[2], dc, 4, hipMemcpyDeviceToHost));
            printf("%d wave(s) per workgroup (one per SIMD), %s: cycles per wave-FMA with 1 / 2 / 4 chains: %.2f / %.2f / %.2f\n", waves,
                   dpp ? "v_fmac_f64_dpp row_newbcast" : "v_fmac_f64 (plain)      ", c[0] / (8.0 * iters), c[1] / (8.0 * iters), c[2] / (8.0 * iters));
        }
    }
    return 0;
}
