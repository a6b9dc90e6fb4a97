/* abi_consumer.c -- a plain C99 consumer of include/krotov_hip.h: the runnable stand-in for the Julia `ccall` binding.
 *
 * Julia is absent from the build image, so julia/KrotovHIP.jl has never executed.  What `ccall` does -- load the shared
 * library, pass a by-value-initialised kr_config and raw pointers to column-major Matrix{ComplexF64} / Vector memory, read
 * int status codes and a C string -- is exactly what a C translation unit does, with no Python, ctypes or NumPy in between.
 * This program drives the call sequence that replaces /root/reference/src/optimize.jl:247-265 (krotov_initial_fw_prop!)
 * and :279-371 (krotov_iteration):
 *
 *     kr_create -> kr_set_initial_states -> kr_forward -> kr_iteration -> kr_read_storage -> kr_destroy
 *
 * on a golden fixture in the ABI's own memory layout (tests/golden/<name>.bin, written by tests/golden/make_golden.py from
 * the oracle twin) and compares with the fixture's expected outputs at the tolerances of tests/test_gpu_parity.py
 * (states 1e-11 absolute, updated pulses 1e-10 relative, int g_a dt 1e-9 relative).
 *
 *     cc -std=c99 -Wall -Wextra -pedantic -I include tests/abi_consumer.c -o abi_consumer -L krotov.jl_amd/lib -lkrotov_hip -lm
 *     ./abi_consumer tests/golden/hermitian_n16.bin          (exit 0 = parity; 3 = no HIP device; 1 = mismatch / error)
 *
 * It also checks the error half of the contract: L = 0 gives the reference's message (workspace.jl:72), a NULL config and a
 * wrong ABI version are refused with a message and no handle, kr_last_error(NULL) is a valid C string.
 */
#include "krotov_hip.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static double *read_doubles(FILE *f, size_t n) {
    double *p = (double *)malloc(sizeof(double) * (n ? n : 1));
    if (!p || fread(p, sizeof(double), n, f) != n) { fprintf(stderr, "abi_consumer: short read\n"); exit(1); }
    return p;
}

static double max_abs_diff(const double *a, const double *b, size_t n) {
    double m = 0.0;
    for (size_t i = 0; i < n; ++i) { const double d = fabs(a[i] - b[i]); if (d > m) m = d; }
    return m;
}

static double max_abs(const double *a, size_t n) {
    double m = 0.0;
    for (size_t i = 0; i < n; ++i) if (fabs(a[i]) > m) m = fabs(a[i]);
    return m;
}

#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "abi_consumer: FAILED: " __VA_ARGS__); fprintf(stderr, "\n"); return 1; } } while (0)

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s fixture.bin\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int32_t hd[8];
    if (fread(hd, sizeof(int32_t), 8, f) != 8 || hd[0] != 0x4B52474C) { fprintf(stderr, "abi_consumer: bad fixture header\n"); return 2; }
    const int N = hd[1], NT = hd[2], K = hd[3], L = hd[4];
    const size_t nn2 = (size_t)N * N * 2, st2 = (size_t)K * N * 2, pl = (size_t)L * NT;
    double *tlist = read_doubles(f, (size_t)NT + 1), *H0 = read_doubles(f, nn2), *Hc = read_doubles(f, (size_t)L * nn2);
    double *eps = read_doubles(f, pl), *S = read_doubles(f, pl), *lam = read_doubles(f, (size_t)L);
    double *psi0 = read_doubles(f, st2), *chiT = read_doubles(f, st2), *psiT0 = read_doubles(f, st2);
    double *eps_new_ref = read_doubles(f, pl), *ga_ref = read_doubles(f, (size_t)L), *psiT1_ref = read_doubles(f, st2);
    double *X0 = read_doubles(f, st2), *Xmid = read_doubles(f, st2);
    fclose(f);

    CHECK(kr_abi_version() == KR_ABI_VERSION, "library ABI %d != header ABI %d", kr_abi_version(), KR_ABI_VERSION);

    /* ---- the error half of the contract: status codes + messages, never a crash, never a handle ---- */
    kr_handle *h = (kr_handle *)0x1;
    CHECK(kr_create(NULL, &h) == KR_ERR_INVALID && h == NULL, "kr_create(NULL) must fail with KR_ERR_INVALID and clear the handle");
    CHECK(kr_last_error(NULL) != NULL && strlen(kr_last_error(NULL)) > 0, "kr_last_error(NULL) must carry the message");
    kr_config cfg;
    memset(&cfg, 0, sizeof cfg);          /* what `zeros` of the Julia struct gives: every optional field at its default */
    cfg.abi_version = KR_ABI_VERSION + 100;
    cfg.N = N; cfg.NT = NT; cfg.K = K; cfg.L = L; cfg.G = 1;
    cfg.tlist = tlist; cfg.H0 = H0; cfg.Hc = Hc;
    cfg.device = -1; cfg.rank = 0; cfg.world = 1;
    CHECK(kr_create(&cfg, &h) == KR_ERR_INVALID && h == NULL, "a wrong abi_version must be refused");
    cfg.abi_version = KR_ABI_VERSION;
    cfg.L = 0;
    CHECK(kr_create(&cfg, &h) == KR_ERR_INVALID && h == NULL, "L = 0 must be refused");
    CHECK(strstr(kr_last_error(NULL), "no controls in trajectories: cannot optimize") != NULL,
          "L = 0 must give the reference's message (workspace.jl:72), got: %s", kr_last_error(NULL));
    cfg.L = L;
    cfg.flags = KR_STORE_FW;
    /* schedule options travel in the config (the environment is only the fallback): an unknown name is refused with a message,
     * a known one is accepted -- here the default value of one, so that the golden results below are those of the default plan */
    cfg.options = "NO_SUCH_OPTION=1";
    {
        int rco = kr_create(&cfg, &h);
        if (rco == KR_ERR_NO_DEVICE) {
            printf("abi_consumer: no HIP device (%s) -- the library refuses to run without one, as documented\n", kr_last_error(NULL));
            return 3;
        }
        CHECK(rco == KR_ERR_INVALID && h == NULL && strstr(kr_last_error(NULL), "NO_SUCH_OPTION") != NULL,
              "an unknown option must be refused by name, got %d: %s", rco, kr_last_error(NULL));
    }
    cfg.options = "ECON=1;PRECOMP=1";

    /* ---- the sequence that replaces krotov_initial_fw_prop! + krotov_iteration ---- */
    int rc = kr_create(&cfg, &h);
    if (rc == KR_ERR_NO_DEVICE) {
        printf("abi_consumer: no HIP device (%s) -- the library refuses to run without one, as documented\n", kr_last_error(NULL));
        return 3;
    }
    CHECK(rc == KR_OK && h != NULL, "kr_create: %d %s", rc, kr_last_error(NULL));
    CHECK(strlen(kr_plan_description(h)) > 0, "kr_plan_description must describe the plan");

    double *psiT = (double *)malloc(sizeof(double) * st2), *eps_new = (double *)malloc(sizeof(double) * pl);
    double *ga = (double *)malloc(sizeof(double) * (size_t)L), *buf = (double *)malloc(sizeof(double) * (size_t)N * 2);
    /* call-order errors are statuses too */
    CHECK(kr_forward(h, eps, psiT) == KR_ERR_STATE, "kr_forward before kr_set_initial_states must fail with KR_ERR_STATE");
    CHECK(strlen(kr_last_error(h)) > 0, "kr_last_error(h) must carry the message");

    rc = kr_set_initial_states(h, psi0);
    CHECK(rc == KR_OK, "kr_set_initial_states: %d %s", rc, kr_last_error(h));
    rc = kr_forward(h, eps, psiT);
    CHECK(rc == KR_OK, "kr_forward: %d %s", rc, kr_last_error(h));
    const double e_fw = max_abs_diff(psiT, psiT0, st2);
    CHECK(e_fw < 1e-11, "forward sweep: |psi(T) - golden| = %.3e", e_fw);

    rc = kr_iteration(h, eps, chiT, S, lam, eps_new, ga, psiT);
    CHECK(rc == KR_OK, "kr_iteration: %d %s", rc, kr_last_error(h));
    const double e_eps = max_abs_diff(eps_new, eps_new_ref, pl) / max_abs(eps_new_ref, pl);
    const double e_ga = max_abs_diff(ga, ga_ref, (size_t)L) / max_abs(ga_ref, (size_t)L);
    const double e_psi = max_abs_diff(psiT, psiT1_ref, st2);
    CHECK(e_eps < 1e-10, "updated pulse: relative error %.3e", e_eps);
    CHECK(e_ga < 1e-9, "int g_a dt: relative error %.3e", e_ga);
    CHECK(e_psi < 1e-10, "psi(T) after the update sweep: %.3e", e_psi);
    CHECK(max_abs_diff(eps_new, eps, pl) > 1e-6, "the update is trivially zero");

    double e_x = 0.0;
    for (int k = 0; k < K; ++k) {
        rc = kr_read_storage(h, KR_STORAGE_BW, k, 0, buf);
        CHECK(rc == KR_OK, "kr_read_storage(bw, %d, 0): %d %s", k, rc, kr_last_error(h));
        double d = max_abs_diff(buf, X0 + (size_t)k * N * 2, (size_t)N * 2);
        if (d > e_x) e_x = d;
        rc = kr_read_storage(h, KR_STORAGE_BW, k, NT / 2, buf);
        CHECK(rc == KR_OK, "kr_read_storage(bw, %d, %d): %d %s", k, NT / 2, rc, kr_last_error(h));
        d = max_abs_diff(buf, Xmid + (size_t)k * N * 2, (size_t)N * 2);
        if (d > e_x) e_x = d;
        rc = kr_read_storage(h, KR_STORAGE_FW, k, NT, buf);      /* forward storage slot N_T = the final state, bit for bit */
        CHECK(rc == KR_OK && max_abs_diff(buf, psiT + (size_t)k * N * 2, (size_t)N * 2) == 0.0, "forward storage slot N_T != final state");
    }
    CHECK(e_x < 1e-11, "backward storage: %.3e", e_x);
    CHECK(kr_read_storage(h, KR_STORAGE_BW, K, 0, buf) == KR_ERR_INVALID, "an out-of-range trajectory index must be refused");

    rc = kr_destroy(h);
    CHECK(rc == KR_OK, "kr_destroy: %d", rc);
    printf("abi_consumer: ok  N=%d NT=%d K=%d L=%d  fw %.2e  eps_new rel %.2e  g_a rel %.2e  psi(T) %.2e  X %.2e\n",
           N, NT, K, L, e_fw, e_eps, e_ga, e_psi, e_x);
    free(tlist); free(H0); free(Hc); free(eps); free(S); free(lam); free(psi0); free(chiT); free(psiT0); free(eps_new_ref);
    free(ga_ref); free(psiT1_ref); free(X0); free(Xmid); free(psiT); free(eps_new); free(ga); free(buf);
    return 0;
}
