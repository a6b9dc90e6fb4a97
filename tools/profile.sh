#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats, then PMC passes, of the bench command.
# Outputs under gpurun_out/prof_<tag>/ ; copy the summaries you want judged into profiles/.
set -o pipefail
TAG=${1:-r04}
CONFIG=${2:-C3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
CMD="python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --config $CONFIG"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.log" 2>&1 || { tail -20 "$OUT/trace.log"; exit 1; }
find "$OUT/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
# PMC passes (own runs, no tracing domains besides kernel-trace): HBM read / write bytes
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $CMD > "$OUT/pmc_fetch.log" 2>&1 || tail -5 "$OUT/pmc_fetch.log"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $CMD > "$OUT/pmc_write.log" 2>&1 || tail -5 "$OUT/pmc_write.log"
python3 $ROOT/tools/summarize_pmc.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
