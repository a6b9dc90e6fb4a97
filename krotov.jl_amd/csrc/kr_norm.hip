// Rigorous upper bounds of the spectral norms of the generator's operators (kr_create).
//
// The Taylor plans take their degree from theta = |dt| (||H0|| + sum |eps_l| ||H_l||), and the Chebyshev-economized step polynomial
// is accurate on [-theta_c, theta_c] ONLY: a norm that is merely estimated from below (power iteration stops on a plateau, or
// at its iteration cap) lets a step use a polynomial outside the interval it was built for, where its error grows exponentially.
// So the norms are BOUNDS:  for C = A^H A (Hermitian, positive semi-definite) and p = 2^J
//      ||A||_2^(2p) = lambda_max(C)^p = lambda_max(C^p) <= ||C^p||_F <= sqrt(n) lambda_max(C)^p ,
// i.e. ||A||_2 <= ||C^p||_F^(1/(2p)) <= n^(1/(4p)) ||A||_2: J squarings of C (C^H C = C^2 for Hermitian C, so one Gram kernel
// serves every step), each followed by a normalisation whose logarithm the host adds up; J is chosen such that the overestimate
// n^(1/(4p)) is below 1.003 (n = 64: p = 512, n = 1024: p = 1024).  Ten small GEMMs per operator on the device that the handle is
// being created on replace up to 2000 host matrix-vector products (N = 1024: 1.9 s of kr_create -> milliseconds).
// Rounding: the products are taken in fp64 (relative error ~ n * 1e-16 per squaring, far inside the 0.3 % of slack); the result
// is widened by 1e-9 relative and never exceeds sqrt(||A||_1 ||A||_inf), which is rigorous as well.
#include <hip/hip_runtime.h>
#include <cmath>
#include <complex>
#include <vector>
#include <algorithm>

typedef std::complex<double> zc;

#define KRN_T 32   // output tile edge; 256 threads, 2 x 2 outputs each
#define KRN_K 16   // inner-dimension slab

// C = X^H X for a column-major n x n matrix X (for Hermitian X: C = X^2).  Both operands are columns of X: contiguous along k.
__global__ __launch_bounds__(256) void krn_gram(const double2 *X, double2 *C, int n) {
    __shared__ double2 sa[KRN_K][KRN_T + 1], sb[KRN_K][KRN_T + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i0 = blockIdx.y * KRN_T, j0 = blockIdx.x * KRN_T;
    double cr[2][2] = {{0, 0}, {0, 0}}, ci[2][2] = {{0, 0}, {0, 0}};
    for (int k0 = 0; k0 < n; k0 += KRN_K) {
        // 16 x 32 elements per operand tile, 512 per pair of loads: thread t loads (kk = t & 15, c = t >> 4) and (kk, c + 16)
        for (int h = 0; h < 2; ++h) {
            const int kk = threadIdx.x & 15, c = (threadIdx.x >> 4) + 16 * h;
            const int k = k0 + kk;
            double2 va = make_double2(0.0, 0.0), vb = make_double2(0.0, 0.0);
            if (k < n && i0 + c < n) va = X[(size_t)(i0 + c) * n + k];
            if (k < n && j0 + c < n) vb = X[(size_t)(j0 + c) * n + k];
            sa[kk][c] = va; sb[kk][c] = vb;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KRN_K; ++kk) {
            double2 a[2], b[2];
            a[0] = sa[kk][ty]; a[1] = sa[kk][ty + 16];
            b[0] = sb[kk][tx]; b[1] = sb[kk][tx + 16];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int v = 0; v < 2; ++v) {   // conj(a) * b
                    cr[u][v] = fma(a[u].x, b[v].x, fma(a[u].y, b[v].y, cr[u][v]));
                    ci[u][v] = fma(a[u].x, b[v].y, fma(-a[u].y, b[v].x, ci[u][v]));
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int i = i0 + ty + 16 * u, j = j0 + tx + 16 * v;
            if (i < n && j < n) C[(size_t)j * n + i] = make_double2(cr[u][v], ci[u][v]);
        }
}

// out[slot] = sum |c|^2 (one workgroup; fixed order: strided partial sums, then an LDS tree -- reproducible)
__global__ __launch_bounds__(1024) void krn_fro2(const double2 *C, size_t count, double *out, int slot) {
    __shared__ double s[1024];
    double acc = 0.0;
    for (size_t i = threadIdx.x; i < count; i += 1024) { const double2 v = C[i]; acc = fma(v.x, v.x, fma(v.y, v.y, acc)); }
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[slot] = s[0];
}
// C /= sqrt(f2[slot])  (a zero matrix stays zero)
__global__ void krn_scale(double2 *C, size_t count, const double *f2, int slot) {
    const double q = f2[slot];
    const double r = (q > 0.0) ? 1.0 / sqrt(q) : 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        double2 v = C[i];
        v.x *= r; v.y *= r;
        C[i] = v;
    }
}

// Upper bounds of ||A_q||_2 for `count` column-major n x n matrices (host pointers); bounds[q] >= the exact norm, <= 1.003 x it (and
// never above sqrt(||A||_1 ||A||_inf)).  Runs on the current device, default stream, synchronous.
hipError_t kr_norm2_upper_bounds(int n, int count, const zc *const *mats, double *bounds) {
    const size_t nn = (size_t)n * n;
    int J = 1;
    while (std::pow((double)n, 1.0 / (4.0 * std::ldexp(1.0, J))) > 1.003 && J < 16) ++J;
    double2 *X = nullptr, *C = nullptr;
    double *f2 = nullptr;
    hipError_t e = hipMalloc((void **)&X, nn * sizeof(double2));
    if (e == hipSuccess) e = hipMalloc((void **)&C, nn * sizeof(double2));
    if (e == hipSuccess) e = hipMalloc((void **)&f2, (size_t)(J + 1) * sizeof(double));
    std::vector<double> f2h(J + 1);
    const dim3 grid((n + KRN_T - 1) / KRN_T, (n + KRN_T - 1) / KRN_T);
    const int sblocks = (int)std::min<size_t>(1024, (nn + 255) / 256);
    for (int q = 0; q < count && e == hipSuccess; ++q) {
        const zc *A = mats[q];
        double n1 = 0.0, ninf = 0.0;
        std::vector<double> rows(n, 0.0);
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int i = 0; i < n; ++i) { const double v = std::abs(A[i + (size_t)j * n]); s += v; rows[i] += v; }
            n1 = std::max(n1, s);
        }
        for (int i = 0; i < n; ++i) ninf = std::max(ninf, rows[i]);
        const double cap = std::sqrt(n1 * ninf);
        if (cap == 0.0) { bounds[q] = 0.0; continue; }
        e = hipMemcpy(X, A, nn * sizeof(double2), hipMemcpyHostToDevice);
        if (e != hipSuccess) break;
        double2 *src = X, *dst = C;
        for (int j = 0; j <= J; ++j) {   // j = 0: C = A^H A; j >= 1: C <- C^2; each normalised to Frobenius norm 1
            hipLaunchKernelGGL(krn_gram, grid, dim3(256), 0, 0, src, dst, n);
            hipLaunchKernelGGL(krn_fro2, dim3(1), dim3(1024), 0, 0, dst, nn, f2, j);
            hipLaunchKernelGGL(krn_scale, dim3(sblocks), dim3(256), 0, 0, dst, nn, f2, j);
            std::swap(src, dst);
        }
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpy(f2h.data(), f2, (size_t)(J + 1) * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) break;
        // log ||C^(2^J)||_F:  L_0 = log f_0,  L_j = 2 L_(j-1) + log f_j  (f_j = Frobenius norm of the squared NORMALISED matrix)
        long double Lg = 0.0L;
        bool zero = false;
        for (int j = 0; j <= J; ++j) {
            if (!(f2h[j] > 0.0)) { zero = true; break; }
            Lg = (j == 0 ? 0.0L : 2.0L * Lg) + 0.5L * std::log((long double)f2h[j]);
        }
        if (zero || !std::isfinite((double)Lg)) { bounds[q] = cap; continue; }   // (underflow of a tiny operator: the cap is a bound too)
        const double b = (double)std::exp(Lg / (2.0L * std::ldexp(1.0L, J)));    // ||C^p||_F^(1/(2p)),  C = A^H A,  p = 2^J
        bounds[q] = std::min(cap, b * (1.0 + 1e-9));
    }
    if (X) (void)hipFree(X);
    if (C) (void)hipFree(C);
    if (f2) (void)hipFree(f2);
    return e;
}
