// Micro-benchmark: one-way latency of a value handed between two workgroups through global memory, for the cache-policy
// combinations a same-XCD exchange could use (the row-split all-gather, kr_device_rowsplit.h) and for workgroups on different XCDs.
// Two workgroups (blockIdx a and b of a grid of 16; blockIdx % 8 = XCD under round-robin placement, checked through
// HW_REG_XCC_ID) bounce a counter: A writes k to slot 0, B waits for k and writes k to slot 1, A waits ... ; time per hop.
//   mode 0: store sc1            / load sc1                 (what the exchange mailboxes use)
//   mode 1: store sc0            / load sc1                 (row-split "local" path)
//   mode 2: store (no bits)      / load sc1
//   mode 3: store sc0            / buffer_inv sc0 ; load sc0   (invalidate the vector L1, then a load the XCD's L2 may serve)
//   mode 4: store (no bits)      / buffer_inv sc0 ; load (no bits)
//   mode 5: store sc0 sc1        / load sc0 sc1
// MEM=0 hipMalloc, 1 uncached, 2 fine-grained.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;

template <int MODE> __device__ __forceinline__ void st(u64 *p, u64 v) {
    if (MODE == 0) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else if (MODE == 1 || MODE == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    else if (MODE == 5) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
}
template <int MODE> __device__ __forceinline__ u64 ld(u64 *p) {
    u64 v;
    if (MODE <= 2) asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (MODE == 3) asm volatile("buffer_inv sc0\n global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (MODE == 4) asm volatile("buffer_inv sc0\n global_load_dwordx2 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int MODE>
__global__ __launch_bounds__(64) void pp(u64 *slots, int wa, int wb, int iters, u64 *out, unsigned *xcc) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if (threadIdx.x == 0) xcc[blockIdx.x] = id & 15;
    if (threadIdx.x != 0) return;
    const int me = (int)blockIdx.x;
    if (me != wa && me != wb) return;
    u64 *mine = slots + (me == wa ? 0 : 16), *other = slots + (me == wa ? 16 : 0);   // 128 B apart
    const u64 t0 = wall_clock64();
    unsigned long long spins = 0;
    for (int k = 1; k <= iters; ++k) {
        if (me == wa) {
            st<MODE>(mine, (u64)k);
            while (ld<MODE>(other) != (u64)k) { if (++spins > (1ull << 26)) { out[2] = 1; return; } }
        } else {
            while (ld<MODE>(other) != (u64)k) { if (++spins > (1ull << 26)) { out[2] = 1; return; } }
            st<MODE>(mine, (u64)k);
        }
    }
    if (me == wa) { out[0] = wall_clock64() - t0; out[1] = spins; }
}

int main(int argc, char **argv) {
    const int iters = 2000;
    const int mem = getenv("MEM") ? atoi(getenv("MEM")) : 0;
    u64 *slots, *out; unsigned *xcc;
    if (mem == 0) hipMalloc(&slots, 4096);
    else if (hipExtMallocWithFlags((void **)&slots, 4096, mem == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, 64); hipMalloc(&xcc, 64);
    const int pairs[2][2] = {{0, 8}, {0, 1}};   // same XCD (0 and 8), different XCDs (0 and 1)
    for (int pr = 0; pr < 2; ++pr) {
        for (int mode = 0; mode < 6; ++mode) {
            hipMemset(slots, 0, 4096); hipMemset(out, 0, 64);
            const int wa = pairs[pr][0], wb = pairs[pr][1];
#define L(M) hipLaunchKernelGGL(pp<M>, dim3(16), dim3(64), 0, 0, slots, wa, wb, iters, out, xcc)
            switch (mode) { case 0: L(0); break; case 1: L(1); break; case 2: L(2); break; case 3: L(3); break; case 4: L(4); break; default: L(5); }
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            u64 o[3]; unsigned x[16];
            hipMemcpy(o, out, 24, hipMemcpyDeviceToHost); hipMemcpy(x, xcc, 64, hipMemcpyDeviceToHost);
            printf("MEM=%d workgroups %d,%d (XCC %u,%u) mode %d: %s %.0f ns per hop (%.1f polls per hop)\n", mem, wa, wb, x[wa], x[wb], mode,
                   o[2] ? "GAVE UP" : "", o[2] ? 0.0 : (double)o[0] * 10.0 / (2.0 * iters), (double)o[1] / iters);
        }
    }
    return 0;
}
