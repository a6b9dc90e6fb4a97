mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity_fullsize.py -q -x 2>&1 | tail -3
(bash tools/ab3.sh C3 10 new; bash tools/ab3.sh C4 10 new; bash tools/ab3.sh D:64,1000,64,2 10 new; bash tools/ab3.sh D:48,400,2000,1 10 new) 2>&1 | grep -v amdgpu.ids
