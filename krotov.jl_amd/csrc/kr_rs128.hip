// Instantiates the row-split sweep kernels for NP = 128 (64 < N <= 256, R workgroups per trajectory).
#include "kr_device_rowsplit.h"
hipError_t kr_launch_rs128(int lmax, int nr, int kind, int grid, hipStream_t s, const KrSweepArgs *a) {
    return kr_launch_rowsplit_np<128>(lmax, nr, kind, grid, s, a);
}
