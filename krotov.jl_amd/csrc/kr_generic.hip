// Instantiates the generic (multi-pass) sweep kernels (kr_device_generic.h): any N, L, K; no co-residency requirement.
#include "kr_device_generic.h"

// kind: 0 known-pulse sweep, 1 partial sums, 2 reduce, 3 pulse update + H(eps), 4 forward step; ktg = trajectories per workgroup
hipError_t kr_launch_generic(int kind, int ktg, int grid, size_t lds_bytes, hipStream_t s, const KrGenArgs *a) {
#define KR_GEN_KTG(KTG)                                                                                                 \
    if (ktg == KTG) {                                                                                                   \
        if (lds_bytes > 48 * 1024) {   /* per (kernel, device): a device-list handle launches on several GPUs (kr_raise_dynamic_lds) */ \
            hipError_t e_ = hipSuccess;                                                                                 \
            if (kind == 0) e_ = kr_raise_dynamic_lds((const void *)kr_gen_known<KTG>, 160 * 1024);                     \
            else if (kind == 1) e_ = kr_raise_dynamic_lds((const void *)kr_gen_partial<KTG>, 160 * 1024 - 2048);       \
            else if (kind == 4) e_ = kr_raise_dynamic_lds((const void *)kr_gen_apply<KTG>, 160 * 1024 - 4096); /* (2 KB static since the fused reduce) */ \
            if (e_ != hipSuccess) { (void)hipGetLastError(); return e_; }                                              \
        }                                                                                                               \
        if (kind == 0) hipLaunchKernelGGL((kr_gen_known<KTG>), dim3(grid), dim3(256), lds_bytes, s, *a);                \
        else if (kind == 1) hipLaunchKernelGGL((kr_gen_partial<KTG>), dim3(grid), dim3(256), lds_bytes, s, *a);         \
        else if (kind == 4) hipLaunchKernelGGL((kr_gen_apply<KTG>), dim3(grid), dim3(256), lds_bytes, s, *a);           \
        else return hipErrorInvalidValue;                                                                               \
        return hipGetLastError();                                                                                       \
    }
    if (kind == 2) { hipLaunchKernelGGL(kr_gen_reduce, dim3(1), dim3(256), 0, s, *a); return hipGetLastError(); }
    if (kind == 3) { hipLaunchKernelGGL(kr_gen_form, dim3(grid), dim3(256), 0, s, *a); return hipGetLastError(); }
    KR_GEN_KTG(1) KR_GEN_KTG(2) KR_GEN_KTG(4)
#undef KR_GEN_KTG
    return hipErrorInvalidValue;
}
