mkdir -p gpurun_out/r05
(bash tools/ab3.sh C3 5 r04 new r04 new && bash tools/ab3.sh C4 5 r04 new && bash tools/ab3.sh C2 8 r04 new r04 new) > gpurun_out/r05/ab2.log 2>&1
cat gpurun_out/r05/ab2.log
for S in 4 16 32; do echo "TBLOCK=$S"; KROTOV_HIP_TBLOCK=$S bash tools/ab3.sh C3 5 new; KROTOV_HIP_TBLOCK=$S bash tools/ab3.sh C4 5 new; done > gpurun_out/r05/tb_s.log 2>&1
cat gpurun_out/r05/tb_s.log
timeout -k 10 300 python -m pytest tests/test_gpu_device_list.py -x -q > gpurun_out/r05/t2a.log 2>&1; tail -3 gpurun_out/r05/t2a.log
timeout -k 10 300 python -m pytest tests/test_gpu_round5.py::test_rccl_handles_in_turn_share_one_communicator tests/test_gpu_device_list.py -x -q > gpurun_out/r05/t2b.log 2>&1; tail -3 gpurun_out/r05/t2b.log
GPU_MAX_HW_QUEUES=8 timeout -k 10 300 python -m pytest tests/test_gpu_round5.py::test_rccl_handles_in_turn_share_one_communicator tests/test_gpu_device_list.py -x -q > gpurun_out/r05/t2c.log 2>&1; tail -3 gpurun_out/r05/t2c.log
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r05/prof_c3 -o c3 -- python3 $GRAFT_REPO_ROOT/tools/time_sweeps.py C3 10 > $GRAFT_REPO_ROOT/gpurun_out/r05/prof_c3.log 2>&1
cd $GRAFT_REPO_ROOT && find gpurun_out/r05/prof_c3 -name "*kernel_stats*" | head -1 | xargs head -25
