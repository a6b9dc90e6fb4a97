// Instantiates the row-split sweep kernels for NP = 128 (64 < N <= 256, R workgroups per trajectory; pair mode: per two trajectories).
#include "kr_device_rowsplit2.h"
// kind: 0 known sweep, 1 update sweep (single mode, kr_device_rowsplit.h); 2 known sweep, 3 update sweep in PAIR mode (kr_device_rowsplit2.h:
// one control, nr <= 32)
hipError_t kr_launch_rs128(int lmax, int nr, int kind, int grid, hipStream_t s, const KrSweepArgs *a) {
    if (kind >= 2) {
        if (lmax != 1) return hipErrorInvalidValue;
#define KR_CASE_P(NRv)                                                                                          \
        if (nr == NRv) {                                                                                        \
            if (kind == 2) kr_launch_kernel((kr_rowsplit_pair_sweep<128, NRv, false>), grid, 512, 0, s, *a);     \
            else if (a->poly) kr_launch_kernel((kr_rowsplit_pair_sweep<128, NRv, true, false, true>), grid, 512, 0, s, *a); /* non-linear control */ \
            else kr_launch_kernel((kr_rowsplit_pair_sweep<128, NRv, true>), grid, 512, 0, s, *a);                \
            return hipGetLastError();                                                                           \
        }
        KR_CASE_P(16) KR_CASE_P(32)
#undef KR_CASE_P
        return hipErrorInvalidValue;
    }
    return kr_launch_rowsplit_np<128>(lmax, nr, kind, grid, s, a);
}
