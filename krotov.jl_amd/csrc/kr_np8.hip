// Instantiates the register-resident sweep kernels for NP = 8 (one translation unit per NP so the
// build parallelises).
#include "kr_device.h"
hipError_t kr_launch_np8(int lmax, int kt, int kind, int grid, hipStream_t s, const KrSweepArgs *a) {
    return kr_launch_np<8>(lmax, kt, kind, grid, s, a);
}
