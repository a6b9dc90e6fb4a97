// Instantiates the streamed-generator sweep kernels for NP = 256 (64 < N <= 256).
#include "kr_device_stream.h"
hipError_t kr_launch_np256(int lmax, int kt, int kind, int grid, hipStream_t s, const KrSweepArgs *a) {
    return kr_launch_stream_np<256>(lmax, kt, kind, grid, s, a);
}
