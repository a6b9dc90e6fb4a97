mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round4.py -q -x 2>&1 | tail -3
(bash tools/ab3.sh C5 4 r04 new) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/ab_rs_guard.log
