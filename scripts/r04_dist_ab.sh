#!/bin/bash
# dev tool, run ON the GPU box: plain step vs world-1 RCCL sharded step (tail overlapped / in order), interleaved
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_dist_ab
mkdir -p "$OUT"
cd $REPO
COMMON="--steps 100 --warmup 3 --no-extra --no-cpu-baseline --no-q10-check --no-ingest --no-strong-check --no-verify"
for rep in 1 2 3; do
  timeout -k 10 300 python3 bench.py $COMMON > "$OUT/plain_$rep.json" 2> "$OUT/plain_$rep.err" || echo "plain failed"
  CSVSIMD_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py $COMMON > "$OUT/overlap_$rep.json" 2> "$OUT/overlap_$rep.err" || echo "overlap failed"
  CSVSIMD_BENCH_FORCE_DIST=1 CSVSIMD_BENCH_TAIL_OVERLAP=0 timeout -k 10 300 python3 bench.py $COMMON > "$OUT/inorder_$rep.json" 2> "$OUT/inorder_$rep.err" || echo "inorder failed"
done
python3 - <<'PY'
import json,os
out=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r04_dist_ab/"
for f in ("plain","overlap","inorder"):
    for rep in (1,2,3):
        try:
            d=json.loads(open(out+f"{f}_{rep}.json").read().strip().splitlines()[-1])
            print(f, rep, "step", d["ms_per_step"], "kernel", d["roofline"]["kernel_ms"], "depth", d["config"]["steps_in_flight"])
        except Exception as e:
            print(f, rep, "ERR", e)
PY
