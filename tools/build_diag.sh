#!/bin/bash
# developer tool: libkrotov_hip_diag.so = the library with per-workgroup arrival stamps in kr_exchange (-DKR_ARRIVALS, NP = 64 kernels)
set -e
cd "$(dirname "$0")/../krotov.jl_amd/csrc"
mkdir -p build/diag
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-function -DKR_ARRIVALS -c kr_np64.hip -o build/diag/kr_np64.o
OBJS=$(ls build/*.o | grep -v kr_np64.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libkrotov_hip_diag.so $OBJS build/diag/kr_np64.o
echo built ../lib/libkrotov_hip_diag.so
