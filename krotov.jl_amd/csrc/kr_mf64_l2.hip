// Instantiates the matrix-core update sweep for NP = 64, L <= 2 (v_mfma_f64_4x4x4_4b_f64, kr_device_mfma.h).
#include "kr_device_mfma.h"
hipError_t kr_launch_mf64_l2(int tiles, int kind, int grid, hipStream_t s, const KrSweepArgs *a) {
    return kr_launch_mfma64<2>(tiles, kind, grid, s, a);
}
