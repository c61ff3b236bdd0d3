#!/bin/bash
# dev tool, run ON the GPU box: round 4's first look — the new full-size test, the sharded step with and without the
# overlapped tail (world-1 RCCL), plain step, self-launched 2-rank rehearsal
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_first
mkdir -p "$OUT"
cd $REPO
timeout -k 10 900 python -m pytest tests/test_gpu_config4.py tests/test_gpu_multi.py -x -q --durations=5 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest.log"
tail -15 "$OUT/pytest.log"
COMMON="--steps 40 --warmup 3 --no-extra --no-cpu-baseline --no-q10-check --no-ingest --no-strong-check"
timeout -k 10 300 python3 bench.py $COMMON > "$OUT/plain.json" 2> "$OUT/plain.err"; echo "plain rc=$?"
CSVSIMD_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py $COMMON > "$OUT/dist_overlap.json" 2> "$OUT/dist_overlap.err"; echo "dist overlap rc=$?"
CSVSIMD_BENCH_FORCE_DIST=1 CSVSIMD_BENCH_TAIL_OVERLAP=0 timeout -k 10 300 python3 bench.py $COMMON > "$OUT/dist_inorder.json" 2> "$OUT/dist_inorder.err"; echo "dist inorder rc=$?"
CSVSIMD_BENCH_REHEARSAL=1 timeout -k 10 600 python3 bench.py --gpus 2 --steps 10 --no-cpu-baseline --no-ingest > "$OUT/rehearsal2.json" 2> "$OUT/rehearsal2.err"; echo "rehearsal rc=$?"
python3 - <<'PY'
import json,os
out=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r04_first/"
for f in ("plain","dist_overlap","dist_inorder","rehearsal2"):
    try:
        d=json.loads(open(out+f+".json").read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["roofline"]["kernel_ms"], d["verified"] and (d["verified"]["tape"], d["verified"]["stitch"]), d["config"].get("sharded_step_tail"), d["config"]["steps_in_flight"])
    except Exception as e:
        print(f, "ERR", e)
PY
