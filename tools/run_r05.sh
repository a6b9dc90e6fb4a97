mkdir -p gpurun_out/r05
one() { # tag lib extra-env...
  tag=$1; lib=$2; shift; shift
  echo "$tag: $(env KROTOV_HIP_LIB=$lib "$@" timeout -k 10 300 python tools/time_sweeps.py ${CFG:-C3} 20 2>&1 | grep -E "^  (update|taylor)" | tr -s ' ' | tr '\n' ';')"
}
L=$PWD/krotov.jl_amd/lib
(
for i in 1 2; do
for t in _r04 ""; do
one "lib$t pd0" $L/libkrotov_hip$t.so KROTOV_HIP_POLL_DELAY=0 KROTOV_HIP_STAMPS=1
done
done
) > gpurun_out/r05/ab11.log 2>&1
cat gpurun_out/r05/ab11.log
timeout -k 10 300 python -m pytest tests/test_gpu_round5.py -q -x -k "pulse or mailbox" 2>&1 | tail -3
bash tools/ab3.sh C2 10 r04 new
