// Instantiates the register-resident sweep kernels for NP = 64 (one translation unit per NP so the
// build parallelises).
#include "kr_device.h"
hipError_t kr_launch_np64(int lmax, int kt, int kind, int grid, hipStream_t s, const KrSweepArgs *a) {
    return kr_launch_np<64>(lmax, kt, kind, grid, s, a);
}
