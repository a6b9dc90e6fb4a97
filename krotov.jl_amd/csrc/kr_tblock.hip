// Instantiates the time-blocked known-pulse sweep (kr_device_tblock.h): block propagators and the parallel apply on the matrix cores.
#include "kr_device.h"
#include "kr_device_tblock.h"
hipError_t kr_launch_tblock(int kind, int grid, hipStream_t s, const KrSweepArgs *a, double2 *Pout) {
    return kr_launch_tblock_impl(kind, grid, s, a, Pout);
}
hipError_t kr_launch_tbpair(int generators, int n_in, int level0, int final, hipStream_t s, const KrSweepArgs *a, const double2 *in, double2 *out) {
    return kr_launch_tbpair_impl(generators, n_in, level0, final, s, a, in, out);
}
