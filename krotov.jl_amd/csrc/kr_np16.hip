// Instantiates the register-resident sweep kernels for NP = 16 (one translation unit per NP so the
// build parallelises).
#include "kr_device.h"
hipError_t kr_launch_np16(int lmax, int kt, int kind, int grid, hipStream_t s, const KrSweepArgs *a) {
    return kr_launch_np<16>(lmax, kt, kind, grid, s, a);
}
