// Instantiates the single-mode row-split sweep kernels for NP = 512 and NP = 1024 (256 < N <= 1024; 32, 64 or 128 workgroups per
// trajectory).
#include "kr_device_rowsplit.h"
// kind: 0 known sweep, 1 update sweep
hipError_t kr_launch_rs512(int lmax, int nr, int kind, int grid, hipStream_t s, const KrSweepArgs *a) {
    if (kind >= 2 && kind != 5) return hipErrorInvalidValue;   // (no pair mode beyond NP = 256; 5: update sweep in turns)
    return kr_launch_rowsplit_np<512>(lmax, nr, kind, grid, s, a);
}
hipError_t kr_launch_rs1024(int lmax, int nr, int kind, int grid, hipStream_t s, const KrSweepArgs *a) {
    if (kind >= 2 && kind != 5) return hipErrorInvalidValue;
    return kr_launch_rowsplit_np<1024>(lmax, nr, kind, grid, s, a);
}
