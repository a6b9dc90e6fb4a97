mkdir -p gpurun_out/r05
(ORACLE_FUZZ_TBLOCK=1 ORACLE_FUZZ_NS=33,40,48,64,64 timeout -k 10 500 python tools/oracle_fuzz.py 120 51; ORACLE_FUZZ_TBLOCK=1 ORACLE_FUZZ_KBIG=1 ORACLE_FUZZ_NS=48,64,64 timeout -k 10 500 python tools/oracle_fuzz.py 60 52; ORACLE_FUZZ_TBLOCK=1 ORACLE_FUZZ_DEVLIST=1 ORACLE_FUZZ_NS=33,64 timeout -k 10 300 python tools/oracle_fuzz.py 40 53) > gpurun_out/r05/fuzz_tblock.log 2>&1
grep -c "^ok" gpurun_out/r05/fuzz_tblock.log; grep -v "^ok" gpurun_out/r05/fuzz_tblock.log | tail -8
KROTOV_RUN_SOAK=1 timeout -k 10 900 python -m pytest tests -q -m gpu -k "soak or all_shapes or 512-3-2-1-64 or 520-36-2-2 or shape2 or odd_k or hermitian_substeps or env0 or env1 or 32" > gpurun_out/r05/soak1.log 2>&1; tail -4 gpurun_out/r05/soak1.log
