// Micro-benchmark of one "Horner round" of the sweep kernels: JS x ds_read_b128 of the x slice, 4*JS fp64
// FMAs, DPP reduce over 4 lanes, one ds_write_b128 by the row owner, one workgroup barrier.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CTRL>
__device__ __forceinline__ double dppmov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

template <int JS, int VAR, int JQ = 4, int THREADS = 256>
__global__ __launch_bounds__(THREADS, THREADS / 256) void round_kernel(double *out, unsigned long long *cyc, int iters) {
    constexpr int NP = JQ * JS;
    constexpr int NPP = NP + JQ;
    constexpr int RT = NP * JQ, SG = THREADS / RT;
    __shared__ __attribute__((aligned(16))) double2 sy[2][SG][NPP];
    __shared__ __attribute__((aligned(16))) double2 sbig[VAR == 6 ? 64 * JS : 1];
    if (VAR == 6) for (int e = threadIdx.x; e < 64 * JS; e += 256) sbig[e] = make_double2(1e-3 * e, 0.1);
    const int tid = threadIdx.x, team = tid / RT, tt = tid % RT, i = tt / JQ, jq = tt % JQ;
    const int pi_ = i + i / JS, xoff = jq * (JS + 1);
    double Er[JS], Ei[JS];
#pragma unroll
    for (int jj = 0; jj < JS; ++jj) { Er[jj] = 1e-3 * (tid + jj); Ei[jj] = 1e-3 * (tid - jj); }
    if (jq == 0) { sy[0][team][pi_] = make_double2(1.0, 0.5); sy[1][team][pi_] = make_double2(1.0, 0.5); }
    double pr = 0.3, pim = 0.2;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int w = 0;
    for (int it = 0; it < iters; ++it) {
        const double2 *xp = &sy[w ^ 1][team][xoff];
        double2 x[JS];
        if (VAR == 4) {  // VAR 4: one LDS read only (dependency), the other operands stay in registers
            const double2 x0 = xp[0];
#pragma unroll
            for (int jj = 0; jj < JS; ++jj) x[jj] = make_double2(x0.x + jj, x0.y - jj);
        } else if (VAR == 6) {  // VAR 6: same reads but every lane its own address (no broadcast), conflict-free
            const double2 *xq = &sbig[(tid & 63)];
#pragma unroll
            for (int jj = 0; jj < JS; ++jj) x[jj] = xq[jj * 64];
        } else {
#pragma unroll
            for (int jj = 0; jj < JS; ++jj) x[jj] = xp[jj];
        }
        double a = 0, b = 0, c = 0, d = 0;
        if (VAR == 5) {  // VAR 5: reads only, 1 add per element instead of 4 FMAs
#pragma unroll
            for (int jj = 0; jj < JS; ++jj) { a += x[jj].x; c += x[jj].y; }
        } else {
#pragma unroll
            for (int jj = 0; jj < JS; ++jj) {
                a = fma(Er[jj], x[jj].x, a); b = fma(Ei[jj], x[jj].y, b);
                c = fma(Er[jj], x[jj].y, c); d = fma(Ei[jj], x[jj].x, d);
            }
        }
        double sr = a - b, si = c + d;
        if (VAR == 7) {  // VAR 7 (round 3): the 4-lane part of the reduction on the matrix core -- v_mfma_f64_4x4x4 with B = ones sums lanes
                         // l, l ^ 16, l ^ 32, l ^ 48 --, then ONE DPP stage (row_half_mirror); timing only (lane roles not modelled)
            sr = __builtin_amdgcn_mfma_f64_4x4x4f64(sr, 1.0, 0.0, 0, 0, 0);
            si = __builtin_amdgcn_mfma_f64_4x4x4f64(si, 1.0, 0.0, 0, 0, 0);
            sr += dppmov<0x141>(sr); si += dppmov<0x141>(si);
        } else if (VAR == 8) {  // VAR 8: transposed reduce -- after the first exchange a lane carries ONE of the two values
            const bool odd = threadIdx.x & 1;
            const double keep = odd ? si : sr, send = odd ? sr : si;
            double v = keep + dppmov<0xB1>(send);
            v += dppmov<0x4E>(v);
            if (JQ == 8) v += dppmov<0x141>(v);
            sr = v; si = dppmov<0xB1>(v);
        } else
        if (VAR != 1) {  // VAR 1: no DPP reduction
            sr += dppmov<0xB1>(sr); si += dppmov<0xB1>(si);
            sr += dppmov<0x4E>(sr); si += dppmov<0x4E>(si);
            if (JQ == 8) { sr += dppmov<0x141>(sr); si += dppmov<0x141>(si); }
        }
        const double yr = fma(1e-3, si, pr), yi = fma(-1e-3, sr, pim);
        if (VAR == 2) {  // VAR 2: every lane writes (no exec masking), to its own padded slot
            sy[w][team][pi_] = make_double2(yr, yi);
        } else {
            if (jq == 0) sy[w][team][pi_] = make_double2(yr, yi);
        }
        if (VAR == 3) __builtin_amdgcn_s_barrier(); else __syncthreads();
        w ^= 1;
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * THREADS + tid] = pr + sy[0][team][pi_].x;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    double *out; unsigned long long *cyc, h;
    (void)hipMalloc(&out, 1 << 22); (void)hipMalloc(&cyc, 64);
    const int iters = 4000;
#define RUN(JS, VAR, label) \
    hipLaunchKernelGGL((round_kernel<JS, VAR>), dim3(256), dim3(256), 0, 0, out, cyc, iters); \
    (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); \
    printf("JS=%2d (NP=%2d) %-28s: %.1f cycles per round\n", JS, 4 * JS, label, (double)h / iters);
    RUN(4, 0, "baseline") RUN(4, 1, "no DPP") RUN(4, 2, "all lanes write") RUN(4, 3, "raw s_barrier")
    RUN(8, 0, "baseline") RUN(8, 1, "no DPP")
    RUN(16, 0, "baseline") RUN(16, 1, "no DPP") RUN(16, 2, "all lanes write") RUN(16, 3, "raw s_barrier")
    RUN(16, 4, "FMAs, 1 LDS read") RUN(16, 5, "16 LDS reads, adds only") RUN(16, 6, "non-broadcast reads")
#define RUN2(JS, JQv, TH, label) \
    hipLaunchKernelGGL((round_kernel<JS, 0, JQv, TH>), dim3(256), dim3(TH), 0, 0, out, cyc, iters); \
    (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); \
    printf("JS=%2d JQ=%d threads=%d (NP=%2d) %-12s: %.1f cycles per round\n", JS, JQv, TH, JQv * JS, label, (double)h / iters);
    RUN2(8, 8, 512, "512-thread") RUN2(4, 16, 1024, "1024-thread") RUN2(16, 4, 256, "256-thread")
#define RUN3(VAR, label) \
    hipLaunchKernelGGL((round_kernel<8, VAR, 8, 512>), dim3(256), dim3(512), 0, 0, out, cyc, iters); \
    (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); \
    printf("JS= 8 JQ=8 threads=512 (NP=64) %-28s: %.1f cycles per round\n", label, (double)h / iters);
    RUN3(0, "baseline (3 DPP stages)") RUN3(1, "no reduction") RUN3(7, "MFMA 4-lane sum + 1 DPP stage") RUN3(8, "transposed reduce") RUN3(3, "raw s_barrier")
    RUN(4, 4, "FMAs, 1 LDS read") RUN(4, 5, "4 LDS reads, adds only") RUN(4, 6, "non-broadcast reads")
    return 0;
}
