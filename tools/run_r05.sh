mkdir -p gpurun_out/r05
for v in "" "TIME_SWEEPS_POLY=1"; do env $v KROTOV_HIP_STAMPS=1 python tools/time_sweeps.py C3 10 2>&1 | grep -E "^C3|update|stamps|taylor" | sed -e "s/'NP'.*exchange_poll_delay/ poll_delay/" ; done
