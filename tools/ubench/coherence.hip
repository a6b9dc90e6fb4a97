// Does data written between two kernels (host-to-device copy, memset, another kernel, recycled allocation) always reach a kernel
// that reads it from every XCD -- also when every XCD's L2 held the OLD contents of those lines?  (The sweeps read generator
// packs, pulses, economized rows and storages with plain / buffer loads and rely on exactly that.)
// hipcc --offload-arch=gfx950 -O3 -o coherence coherence.hip ; ./coherence [iterations]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// every block reads the whole buffer (so that every XCD's L2 holds all of it); counts words != expect(i)
template <int BUF>
__global__ __launch_bounds__(256) void read_all(const unsigned long long *p, size_t n, unsigned long long salt, unsigned long long *bad, unsigned long long *sink) {
    unsigned long long acc = 0, nb = 0;
    if (BUF) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)(n * 8), 0x00020000);
        for (size_t i = threadIdx.x * 2; i < n; i += 512) {
            const u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(i * 8), 0, 0);
            const unsigned long long a = ((unsigned long long)v.y << 32) | v.x, b = ((unsigned long long)v.w << 32) | v.z;
            nb += (a != (i * 0x9E3779B97F4A7C15ull ^ salt)) + (b != ((i + 1) * 0x9E3779B97F4A7C15ull ^ salt));
            acc += a + b;
        }
    } else {
        for (size_t i = threadIdx.x; i < n; i += 256) {
            const unsigned long long v = p[i];
            nb += (v != (i * 0x9E3779B97F4A7C15ull ^ salt));
            acc += v;
        }
    }
    if (nb) atomicAdd(bad, nb);
    if (acc == 0x1234567ull) sink[0] = acc;
}
__global__ void write_all(unsigned long long *p, size_t n, unsigned long long salt) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = i * 0x9E3779B97F4A7C15ull ^ salt;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 40;
    const size_t n = (size_t)1 << 18;   // 2 MB: the size of a generator pack at N = 256
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned long long *bad, *sink, hb;
    CK(hipMalloc(&bad, 8)); CK(hipMalloc(&sink, 8));
    std::vector<unsigned long long> host(n);
    auto fill_host = [&](unsigned long long salt) { for (size_t i = 0; i < n; ++i) host[i] = i * 0x9E3779B97F4A7C15ull ^ salt; };
    auto count = [&](const unsigned long long *p, unsigned long long salt, int buf) {
        CK(hipMemsetAsync(bad, 0, 8, s));
        if (buf) read_all<1><<<1024, 256, 0, s>>>(p, n, salt, bad, sink); else read_all<0><<<1024, 256, 0, s>>>(p, n, salt, bad, sink);
        CK(hipMemcpyAsync(&hb, bad, 8, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        return hb;
    };
    const char *names[] = {"hipMemcpy (sync, pageable) after every XCD cached the old lines", "hipMemcpyAsync on the non-blocking stream",
                           "kernel writes, next kernel reads from every XCD", "hipFree + hipMalloc + hipMemcpy (recycled allocation)",
                           "hipMemsetAsync then kernel (expects zeros)"};
    for (int sc = 0; sc < 5; ++sc) {
        for (int buf = 0; buf < 2; ++buf) {
            unsigned long long total_bad = 0, worst = 0;
            unsigned long long *p;
            CK(hipMalloc(&p, n * 8));
            for (int it = 0; it < iters; ++it) {
                const unsigned long long old_salt = 0x1111ull * (it + 1) + sc, new_salt = 0x7777ull * (it + 3) + sc;
                write_all<<<256, 256, 0, s>>>(p, n, old_salt);
                unsigned long long b0 = count(p, old_salt, buf);   // all XCDs now hold the old contents
                b0 += count(p, old_salt, buf);
                unsigned long long b1 = 0;
                if (sc == 0) { fill_host(new_salt); CK(hipMemcpy(p, host.data(), n * 8, hipMemcpyHostToDevice)); b1 = count(p, new_salt, buf); }
                if (sc == 1) { fill_host(new_salt); CK(hipMemcpyAsync(p, host.data(), n * 8, hipMemcpyHostToDevice, s)); b1 = count(p, new_salt, buf); }
                if (sc == 2) { write_all<<<256, 256, 0, s>>>(p, n, new_salt); b1 = count(p, new_salt, buf); }
                if (sc == 3) { CK(hipStreamSynchronize(s)); CK(hipFree(p)); CK(hipMalloc(&p, n * 8)); fill_host(new_salt); CK(hipMemcpy(p, host.data(), n * 8, hipMemcpyHostToDevice)); b1 = count(p, new_salt, buf); }
                if (sc == 4) {   // memset to zero: a word that still matches the OLD pattern is a stale read
                    CK(hipMemsetAsync(p, 0, n * 8, s));
                    b1 = n * 1024 - count(p, old_salt, buf);
                }
                total_bad += b1;
                if (b1 > worst) worst = b1;
                if (b0) printf("  (warm-up mismatch %llu?)\n", b0);
            }
            CK(hipFree(p));
            printf("%-70s %-12s stale words %llu (worst iteration %llu) over %d iterations x 1024 blocks x %zu words\n", names[sc], buf ? "buffer loads" : "global loads", total_bad, worst, iters, n);
        }
    }
    return 0;
}
