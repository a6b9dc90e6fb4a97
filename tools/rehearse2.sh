# developer tool: bench.py's N = 2 path with both ranks on the one GPU (gloo rendezvous, real IPC mailboxes):
#   rehearse2.sh [traj per rank] [steps]
T=${1:-96}; S=${2:-5}
KROTOV_BENCH_BACKEND=gloo KROTOV_HIP_MAX_WG=$T timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 400)) bench.py --gpus 2 --steps $S --warmup 2 --traj $T --no-cpu-baseline 2>&1 \
  | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('2 ranks x %s traj: %.3f ms/step, update kernel %.3f ms' % (sys.argv[1], d['ms_per_step'], d['roofline']['avg_launch_ms']))" $T
