// Instantiates the row-split sweep kernels for NP = 256 (64 < N <= 256, R workgroups per trajectory).
#include "kr_device_rowsplit.h"
hipError_t kr_launch_rs256(int lmax, int nr, int kind, int grid, hipStream_t s, const KrSweepArgs *a) {
    return kr_launch_rowsplit_np<256>(lmax, nr, kind, grid, s, a);
}
