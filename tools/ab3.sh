# developer tool: same-box comparison of builds of the library, all three sweeps (krotov.jl_amd/lib/libkrotov_hip_<tag>.so; "new" = the tree's)
CFG=${1:-C3}; REPS=${2:-5}; shift; shift
for t in "$@"; do
  lib=$PWD/krotov.jl_amd/lib/libkrotov_hip_$t.so; [ "$t" = new ] && lib=$PWD/krotov.jl_amd/lib/libkrotov_hip.so
  echo "$CFG $t: $(KROTOV_HIP_LIB=$lib timeout -k 10 300 python tools/time_sweeps.py $CFG $REPS 2>&1 | grep -E '^  (forward|backward|update|wall)' | tr -s ' ' | tr '\n' ';')"
done
