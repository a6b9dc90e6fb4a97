# developer tool: same-box comparison of several builds of the library (krotov.jl_amd/lib/libkrotov_hip_<tag>.so)
CFG=${1:-C2}; REPS=${2:-5}; shift; shift
for t in "$@"; do
  lib=$PWD/krotov.jl_amd/lib/libkrotov_hip_$t.so; [ "$t" = new ] && lib=$PWD/krotov.jl_amd/lib/libkrotov_hip.so
  echo "$t: $(KROTOV_HIP_LIB=$lib timeout -k 10 300 python tools/time_sweeps.py $CFG $REPS 2>&1 | grep '^  update' | head -1)"
done
