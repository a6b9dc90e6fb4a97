#!/bin/bash
# developer tool: libkrotov_hip_diag.so = the library with per-workgroup arrival stamps in kr_exchange (-DKR_ARRIVALS) in the given
# translation units (default: kr_np64; e.g. `tools/build_diag.sh kr_rs256` for the row-split kernels at N = 256).
# DIAG_DEFS replaces the define, e.g. DIAG_DEFS=-DKR_GATHER_SEQ: every all-gather element carries its gather counter and the kernel counts
# elements that the one-bit validity accepted from an earlier use of the buffer (tools/diag_gather_seq.py)
set -e
cd "$(dirname "$0")/../krotov.jl_amd/csrc"
mkdir -p build/diag${DIAG_TAG:+_$DIAG_TAG}
DD=build/diag${DIAG_TAG:+_$DIAG_TAG}
TUS=${@:-kr_np64}
SKIP=""
DIAG=""
for t in $TUS; do
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-function ${DIAG_DEFS:--DKR_ARRIVALS} -c $t.hip -o $DD/$t.o
  SKIP="$SKIP -e /$t.o"
  DIAG="$DIAG $DD/$t.o"
done
OBJS=$(ls build/*.o | grep -v $SKIP)
OUT=../lib/libkrotov_hip_${DIAG_TAG:-diag}.so   # DIAG_TAG names the library: same-box A/B of variants with tools/ab.sh
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS $DIAG
echo built $OUT
