// Instantiates the time-blocked known-pulse sweep (kr_device_tblock.h): block propagators and the parallel apply on the matrix cores.
#include "kr_device.h"
#include "kr_device_tblock.h"
hipError_t kr_launch_tblock(int kind, int grid, hipStream_t s, const KrSweepArgs *a, double2 *Pout) {
    return kr_launch_tblock_impl(kind, grid, s, a, Pout);
}
