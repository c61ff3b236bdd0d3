#!/bin/bash
# dev tool, run ON the GPU box (through gpurun): kernel trace + stats of the SHARDED step with a real RCCL communicator at
# world = 1 (all a one-GPU box allows): which kernels one step consists of, and what the small ones cost.
set -e -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_dist
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export CSVSIMD_BENCH_FORCE_DIST=1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $REPO/bench.py --steps 10 --warmup 2 \
    --no-extra --no-cpu-baseline --no-verify --no-q10-check --no-ingest > "$OUT/stats.log" 2>&1
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_dist_world1.csv" \;
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*agent_info.csv" -delete
cat "$OUT/kernel_stats_dist_world1.csv" | cut -c1-220
