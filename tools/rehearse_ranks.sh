#!/bin/bash
# developer tool: bench.py's N > 1 path with N ranks sharing the ONE GPU of the build box (gloo for the host collectives, the real
# IPC mailboxes between the ranks' kernels); every rank gets 256 / N workgroups.  Strong scaling of the named config, as the driver
# runs it on N GPUs -- minus xGMI: the inter-rank hop stays in local HBM.   tools/rehearse_ranks.sh N [CONFIG] [extra bench args]
N=${1:-2}; CFG=${2:-C3}; shift; shift
export KROTOV_BENCH_BACKEND=gloo KROTOV_HIP_MAX_WG=$((256 / N)) HSA_ENABLE_IPC_MODE_LEGACY=0
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 400)) \
    bench.py --gpus $N --steps 10 --warmup 2 --config $CFG --no-cpu-baseline "$@"
