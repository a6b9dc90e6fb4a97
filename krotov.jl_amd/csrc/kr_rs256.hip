// Instantiates the row-split sweep kernels for NP = 256 (64 < N <= 256, R workgroups per trajectory; pair mode: per two trajectories).
#include "kr_device_rowsplit2.h"
// kind: 0 known sweep, 1 update sweep (single mode, kr_device_rowsplit.h); 2 known sweep, 3 update sweep in PAIR mode (kr_device_rowsplit2.h:
// one control, nr <= 32)
hipError_t kr_launch_rs256(int lmax, int nr, int kind, int grid, hipStream_t s, const KrSweepArgs *a) {
    if (kind == 5) return kr_launch_rowsplit_np<256>(lmax, nr, kind, grid, s, a);   // update sweep in turns (single mode, two controls)
    if (kind >= 2) {
        if (lmax != 1) return hipErrorInvalidValue;
        if (a->rs_sets > 1) {   // several pairs per group in turn (N = 256, 65..256 trajectories): NR = 32 only
            if (nr != 32 || a->rs_sets > KR_PAIR_MAX_SETS || a->rs_scratch == nullptr) return hipErrorInvalidValue;
            if (kind == 2) kr_launch_kernel((kr_rowsplit_pair_sweep<256, 32, false, true>), grid, 512, 0, s, *a);
            else if (a->poly) kr_launch_kernel((kr_rowsplit_pair_sweep<256, 32, true, true, true>), grid, 512, 0, s, *a);   // non-linear control
            else kr_launch_kernel((kr_rowsplit_pair_sweep<256, 32, true, true>), grid, 512, 0, s, *a);
            return hipGetLastError();
        }
#define KR_CASE_P(NRv)                                                                                          \
        if (nr == NRv) {                                                                                        \
            if (kind == 2) kr_launch_kernel((kr_rowsplit_pair_sweep<256, NRv, false>), grid, 512, 0, s, *a);     \
            else if (a->poly) kr_launch_kernel((kr_rowsplit_pair_sweep<256, NRv, true, false, true>), grid, 512, 0, s, *a); \
            else kr_launch_kernel((kr_rowsplit_pair_sweep<256, NRv, true>), grid, 512, 0, s, *a);                \
            return hipGetLastError();                                                                           \
        }
        KR_CASE_P(16) KR_CASE_P(32)
#undef KR_CASE_P
        return hipErrorInvalidValue;
    }
    return kr_launch_rowsplit_np<256>(lmax, nr, kind, grid, s, a);
}
