// Background load for robustness runs: a few dozen workgroups that stream memory in launch after launch for a given time, so that
// the persistent kernels of another process get an UNEVEN chip -- some CUs taken, the dispatcher's round-robin over the XCDs
// disturbed (the row-split groups then miss their one-XCD layout and take the placement-independent path), the L2s and the
// memory system busy.  hipcc --offload-arch=gfx950 -O3 -o stressor stressor.hip ; ./stressor SECONDS [BLOCKS=40]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void churn(double2 *p, size_t n, int rounds) {
    double2 acc = make_double2(0.0, 0.0);
    for (int r = 0; r < rounds; ++r)
        for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
            double2 v = p[i];
            acc.x += v.x; acc.y += v.y;
            v.x = acc.y * 1e-9; p[i] = v;
        }
    if (acc.x == 1.2345) p[0] = acc;
}
int main(int argc, char **argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 10.0;
    const int blocks = argc > 2 ? atoi(argv[2]) : 40;
    const size_t n = (size_t)1 << 22;   // 64 MB
    double2 *p;
    if (hipMalloc(&p, n * sizeof(double2)) != hipSuccess) return 1;
    (void)hipMemset(p, 0, n * sizeof(double2));
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        hipLaunchKernelGGL(churn, dim3(blocks), dim3(256), 0, 0, p, n, 1);
        (void)hipDeviceSynchronize();
        ++launches;
    }
    printf("stressor: %ld launches of %d blocks in %.1f s\n", launches, blocks, secs);
    return 0;
}
