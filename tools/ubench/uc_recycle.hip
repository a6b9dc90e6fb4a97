// Does an ordinary hipMalloc block behave after the process has allocated and FREED uncached blocks?  (round 4: the oracle fuzzer
// with device lists lost partial sums in ordinary mailboxes of handles that were created after earlier handles had freed their
// uncached mailboxes; krotov_hip.hip now pools such blocks.)  The library's pattern without the library:
//   "handle": a mailbox of 1051136 bytes -- uncached (hipExtMallocWithFlags) or ordinary (hipMalloc), a few ordinary neighbours,
//   zeroed with hipMemset; one exchange: hipMemsetAsync + a persistent kernel of P workgroups whose wave 0 publishes one word into
//   each of 8 replicas (64 KB apart, agent-scope stores) and then polls the replica of its XCD until all P words are there (bounded);
//   afterwards the host reads the mailbox back: every word must be present in every replica.
// Two host threads do this at the same time on two streams (the two shards of a device list).
//   hipcc -O3 --offload-arch=gfx950 -o uc_recycle uc_recycle.hip -lpthread ; ./uc_recycle [handles=400] [mode=0] [host threads=2]
//   mode 0: kinds alternate by a fixed pseudo-random sequence (free everything)   1: ordinary only   2: uncached only
//   mode 3: as 0, but uncached blocks are kept and reused instead of freed
//   mode 4: 60 handles of mixed kinds on ONE thread (everything freed), then two threads with ordinary blocks only; mode 5: the control
//           (phase 1 with ordinary blocks only)
//   mode 7: the library's device list (see groups() below), everything freed; mode 8: uncached blocks kept and reused;
//   mode 10: 960 ordinary blocks of random sizes allocated and freed first, then mode 8 in turn (uncached blocks allocated LATE in a process)
//   mode 11: one handle at a time: an uncached mailbox used and freed, then a 32 MB ordinary block that may cover its address (see big_after_uc)
//   mode 9: as 7, but the two shards' exchanges run one after the other from the main thread (no second thread, nothing concurrent)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
#define REPL 8
#define RSTRIDE 8192
#define WORDS (2 * REPL * RSTRIDE + 320)

__device__ __forceinline__ unsigned xcc_id() { unsigned id; asm("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); return id & 7u; }

__global__ __launch_bounds__(256) void exchange(unsigned long long *mail, int P, unsigned *result /*[P]*/, unsigned long long ticks, unsigned seq) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wg = blockIdx.x;
    if (wave != 0) return;
    if (lane < 4 * REPL && (lane & 3) == 0)
        __hip_atomic_store(mail + (size_t)(lane >> 2) * RSTRIDE + (size_t)wg * 2, ((unsigned long long)seq << 32) | ((unsigned)wg << 8) | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long *rep = mail + (size_t)xcc_id() * RSTRIDE;
    unsigned ok = 0;
    for (;;) {
        bool all = true;
        for (int q = lane; q < P; q += 64) all = all && ((__hip_atomic_load(rep + (size_t)q * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1ull) != 0ull);
        if (__all(all)) { ok = 1; break; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > ticks) break;
        __builtin_amdgcn_s_sleep(2);
    }
    if (lane == 0) result[wg] = ok | (xcc_id() << 8);
}

static std::atomic<int> g_bad{0}, g_handles{0};
// UC_KIND=uncached (default) | finegrained | host: what the "special" blocks are (host: pinned host memory, hipHostMalloc / hipHostFree --
// the exchange then runs on ordinary mailboxes only and the special block is just allocated, touched by a memset and freed)
static int g_kind = 0;
static void special_malloc(void **p, size_t bytes) {
    if (g_kind == 2) CK(hipHostMalloc(p, bytes, hipHostMallocDefault));
    else CK(hipExtMallocWithFlags(p, bytes, g_kind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached));
}
static void special_free(void *p) { if (g_kind == 2) CK(hipHostFree(p)); else CK(hipFree(p)); }
static std::vector<void *> g_uc_pool[2];

static void worker(int tid, int handles, int mode, unsigned seed) {
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    std::vector<unsigned long long> host(WORDS);
    unsigned rs = seed;
    for (int it = 0; it < handles; ++it) {
        rs = rs * 1664525u + 1013904223u;
        const int P = 2 + (int)((rs >> 8) % 60);
        const bool uc = (mode == 2) || ((mode == 0 || mode == 3 || mode == 6) && ((rs >> 20) % 3 == 0));
        unsigned long long *mail = nullptr;
        void *nb[3] = {nullptr, nullptr, nullptr};
        unsigned *res = nullptr;
        for (int i = 0; i < 2; ++i) CK(hipMalloc(&nb[i], 65536 + 4096 * ((rs >> (4 * i)) & 15)));
        if (uc) {
            if (mode == 3 && !g_uc_pool[tid].empty()) { mail = (unsigned long long *)g_uc_pool[tid].back(); g_uc_pool[tid].pop_back(); }
            else CK(hipExtMallocWithFlags((void **)&mail, WORDS * 8, hipDeviceMallocUncached));
        } else CK(hipMalloc((void **)&mail, WORDS * 8));
        CK(hipMalloc(&nb[2], 1 << 20));
        CK(hipMalloc((void **)&res, 64 * 4));
        CK(hipMemset(mail, 0, WORDS * 8));
        CK(hipDeviceSynchronize());
        for (int call = 0; call < 3; ++call) {
            CK(hipMemsetAsync(res, 0, 64 * 4, st));
            CK(hipMemsetAsync(mail, 0, WORDS * 8, st));
            hipLaunchKernelGGL(exchange, dim3(P), dim3(256), 0, st, mail, P, res, 20000000ull /* 0.2 s */, (unsigned)(it * 4 + call + 1));
            CK(hipGetLastError());
            unsigned r[64];
            CK(hipMemcpyAsync(r, res, 64 * 4, hipMemcpyDeviceToHost, st));
            CK(hipMemcpyAsync(host.data(), mail, WORDS * 8, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            int gave_up = 0, missing = 0;
            for (int q = 0; q < P; ++q) gave_up += (r[q] & 1u) ? 0 : 1;
            for (int rp = 0; rp < REPL; ++rp) for (int q = 0; q < P; ++q) missing += (host[(size_t)rp * RSTRIDE + (size_t)q * 2] & 1ull) ? 0 : 1;
            if (gave_up || missing) {
                g_bad++;
                printf("thread %d handle %d call %d (%s mailbox %p, P = %d): %d workgroups gave up, %d of %d words missing in memory\n", tid, it, call, uc ? "uncached" : "ordinary", (void *)mail, P, gave_up, missing, REPL * P);
                fflush(stdout);
            }
        }
        g_handles++;
        for (int i = 0; i < 3; ++i) CK(hipFree(nb[i]));
        CK(hipFree(res));
        if (uc && mode == 3) g_uc_pool[tid].push_back(mail); else CK(hipFree(mail));
    }
    CK(hipStreamDestroy(st));
}

// mode 7 / 8: the library's device list, literally: the MAIN thread allocates both shards' blocks (mailbox ordinary or uncached like
// kr_create chooses by the number of workgroups, plus a small uncached block per shard -- the mailbox between GPUs), every call runs
// the two shards' exchanges from two threads (a new std::thread per call, like group_run), the main thread frees everything.
// 8: the uncached blocks are kept and reused.
static void one_exchange(hipStream_t st, unsigned long long *mail, unsigned *res, int P, unsigned seq, int *gave_up, int *missing) {
    std::vector<unsigned long long> host(WORDS);
    unsigned r[64];
    CK(hipMemsetAsync(res, 0, 64 * 4, st));
    CK(hipMemsetAsync(mail, 0, WORDS * 8, st));
    hipLaunchKernelGGL(exchange, dim3(P), dim3(256), 0, st, mail, P, res, 20000000ull, seq);
    CK(hipGetLastError());
    CK(hipMemcpyAsync(r, res, 64 * 4, hipMemcpyDeviceToHost, st));
    CK(hipMemcpyAsync(host.data(), mail, WORDS * 8, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    *gave_up = 0; *missing = 0;
    for (int q = 0; q < P; ++q) *gave_up += (r[q] & 1u) ? 0 : 1;
    for (int rp = 0; rp < REPL; ++rp) for (int q = 0; q < P; ++q) *missing += (host[(size_t)rp * RSTRIDE + (size_t)q * 2] & 1ull) ? 0 : 1;
}
static void groups(int handles, bool pool, bool in_turn = false) {
    std::vector<void *> big_pool, small_pool;
    unsigned rs = 99u;
    for (int it = 0; it < handles; ++it) {
        rs = rs * 1664525u + 1013904223u;
        const bool single = getenv("UC_SINGLE") ? true : ((rs >> 5) % 4 == 0);   // some handles have one shard only (UC_SINGLE=1: all)
        const int P = ((rs >> 20) % 3 == 0) ? 64 : 2 + (int)((rs >> 8) % 60);
        const bool uc = P >= 64 && g_kind != 2;   // (pinned host memory: only the small block is special)
        const int S = single ? 1 : 2;
        hipStream_t st[2]; unsigned long long *mail[2]; void *xm[2]; unsigned *res[2]; void *nb[2][3];
        for (int s = 0; s < S; ++s) {
            CK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
            for (int i = 0; i < 3; ++i) CK(hipMalloc(&nb[s][i], 65536 + 4096 * ((rs >> (4 * i + s)) & 15)));
            if (uc) {
                if (pool && !big_pool.empty()) { mail[s] = (unsigned long long *)big_pool.back(); big_pool.pop_back(); }
                else special_malloc((void **)&mail[s], WORDS * 8);
            } else CK(hipMalloc((void **)&mail[s], WORDS * 8));
            CK(hipMemset(mail[s], 0, WORDS * 8));
            CK(hipMalloc((void **)&res[s], 64 * 4));
            xm[s] = nullptr;
            if (!getenv("UC_NOSMALL")) {   // (UC_NOSMALL=1: no 2.5 KB uncached block beside the mailbox)
                if (pool && !small_pool.empty()) { xm[s] = small_pool.back(); small_pool.pop_back(); }
                else special_malloc(&xm[s], 2560);
                CK(hipMemset(xm[s], 0, 2560));
            }
            CK(hipDeviceSynchronize());
        }
        for (int call = 0; call < 3; ++call) {
            int gu[2] = {0, 0}, mi[2] = {0, 0};
            const unsigned seq = (unsigned)(it * 4 + call + 1);
            if (S == 2 && in_turn) {
                one_exchange(st[0], mail[0], res[0], P, seq, &gu[0], &mi[0]);
                one_exchange(st[1], mail[1], res[1], P, seq, &gu[1], &mi[1]);
            } else if (S == 2) {
                std::thread t(one_exchange, st[1], mail[1], res[1], P, seq, &gu[1], &mi[1]);
                one_exchange(st[0], mail[0], res[0], P, seq, &gu[0], &mi[0]);
                t.join();
            } else one_exchange(st[0], mail[0], res[0], P, seq, &gu[0], &mi[0]);
            for (int s = 0; s < S; ++s)
                if (gu[s] || mi[s]) {
                    g_bad++;
                    printf("handle %d call %d shard %d of %d (%s mailbox %p, P = %d): %d workgroups gave up, %d of %d words missing in memory\n", it, call, s, S, uc ? "uncached" : "ordinary", (void *)mail[s], P, gu[s], mi[s], REPL * P);
                    fflush(stdout);
                }
        }
        g_handles++;
        for (int s = 0; s < S; ++s) {
            for (int i = 0; i < 3; ++i) CK(hipFree(nb[s][i]));
            CK(hipFree(res[s]));
            if (uc && pool) big_pool.push_back(mail[s]); else if (uc) special_free(mail[s]); else CK(hipFree(mail[s]));
            if (xm[s]) { if (pool) small_pool.push_back(xm[s]); else special_free(xm[s]); }
            CK(hipStreamDestroy(st[s]));
        }
    }
}

// mode 11: ONE handle at a time, as a single kr_create / kr_destroy sequence does it: an uncached mailbox is used and freed, then a LARGE
// ordinary block (32 MB: served by the driver directly, not from the runtime's cached fragments -- the library's state storage is of that
// size) is allocated; where it covers the address the uncached block had, the exchange runs exactly there, else at its start.
static void big_after_uc(int handles) {
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned *res;
    CK(hipMalloc((void **)&res, 64 * 4));
    unsigned rs = 31u;
    int covered = 0;
    for (int it = 0; it < handles; ++it) {
        rs = rs * 1664525u + 1013904223u;
        const int P = 2 + (int)((rs >> 8) % 60);
        unsigned long long *uc = nullptr;
        CK(hipExtMallocWithFlags((void **)&uc, WORDS * 8, hipDeviceMallocUncached));
        CK(hipMemset(uc, 0, WORDS * 8));
        int gu, mi;
        one_exchange(st, uc, res, P, (unsigned)(it * 8 + 1), &gu, &mi);
        if (gu || mi) { g_bad++; printf("handle %d uncached mailbox %p: %d gave up, %d missing\n", it, (void *)uc, gu, mi); }
        CK(hipFree(uc));
        char *big = nullptr;
        const size_t bytes = getenv("UC_BIG_KB") ? (size_t)atol(getenv("UC_BIG_KB")) << 10 : (size_t)32 << 20;   // (UC_BIG_KB: another size)
        CK(hipMalloc((void **)&big, bytes));
        CK(hipMemset(big, 0, bytes));
        CK(hipDeviceSynchronize());
        const bool over = (char *)uc >= big && (char *)uc + WORDS * 8 <= big + bytes;
        covered += over ? 1 : 0;
        unsigned long long *mail = over ? uc : (unsigned long long *)big;
        for (int call = 0; call < 3; ++call) {
            one_exchange(st, mail, res, P, (unsigned)(it * 8 + 2 + call), &gu, &mi);
            if (gu || mi) {
                g_bad++;
                printf("handle %d call %d: ordinary 32 MB block %p, exchange at %p (%s the freed uncached block's address), P = %d: %d workgroups gave up, %d of %d words missing in memory\n",
                       it, call, (void *)big, (void *)mail, over ? "AT" : "not at", P, gu, mi, REPL * P);
                fflush(stdout);
            }
        }
        g_handles++;
        CK(hipFree(big));
    }
    printf("big block covered the freed uncached block's address in %d of %d handles\n", covered, handles);
    CK(hipFree(res));
    CK(hipStreamDestroy(st));
}

int main(int argc, char **argv) {
    const int handles = argc > 1 ? atoi(argv[1]) : 400, mode = argc > 2 ? atoi(argv[2]) : 0, threads = argc > 3 ? atoi(argv[3]) : 2;
    if (getenv("UC_KIND")) g_kind = !strcmp(getenv("UC_KIND"), "finegrained") ? 1 : !strcmp(getenv("UC_KIND"), "host") ? 2 : 0;
    CK(hipSetDevice(0));
    if (mode == 11) big_after_uc(handles);
    else if (mode == 10) {   // ordinary blocks of many sizes allocated and freed first, THEN the pooled pattern: uncached blocks that are allocated late
        std::vector<void *> v;
        unsigned rs = 5u;
        for (int round = 0; round < 40; ++round) {
            for (int i = 0; i < 24; ++i) { rs = rs * 1664525u + 1013904223u; void *q; CK(hipMalloc(&q, 4096 + (size_t)(rs >> 8) % (3u << 20))); CK(hipMemset(q, 1, 4096)); v.push_back(q); }
            CK(hipDeviceSynchronize());
            for (void *q : v) CK(hipFree(q));
            v.clear();
        }
        groups(handles, true, true);
    } else if (mode == 7 || mode == 8 || mode == 9) groups(handles, mode == 8, mode == 9);
    else if (mode == 4 || mode == 5) {   // phase 1 on one thread (4: kinds mixed and freed, 5: ordinary only), phase 2: two threads, ORDINARY blocks only
        worker(0, 60, mode == 4 ? 6 : 1, 4242u);
        printf("phase 1 done: %d exchanges with lost or unseen words\n", g_bad.load());
        std::thread t1(worker, 1, handles, 1, 777u);
        worker(0, handles, 1, 12345u);
        t1.join();
    } else if (threads > 1) {
        std::thread t1(worker, 1, handles, mode, 777u);
        worker(0, handles, mode, 12345u);
        t1.join();
    } else worker(0, handles, mode, 12345u);
    printf("uc_recycle mode %d, %d host thread(s): %d handles, %d exchanges with lost or unseen words\n", mode, threads, g_handles.load(), g_bad.load());
    return g_bad.load() ? 1 : 0;
}
