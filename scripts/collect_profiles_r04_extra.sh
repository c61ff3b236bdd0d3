#!/bin/bash
# dev tool, run ON the GPU box (through gpurun): round 4's other kept records — the batched launch at 1 and 8 GiB per batch
# (rocprofv3 kernel stats), the sharded step with a real RCCL communicator at world 1 (bench line + kernel stats: tail on
# its own stream), the plain N = 1 line it is compared with, and the self-launched two-rank rehearsal on one GPU.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_r04_extra
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/batch" -- python3 $REPO/bench.py --only-batch > "$OUT/batch.json" 2> "$OUT/batch.err"
find "$OUT/batch" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_batch.csv" \;
FLAGS="--steps 100 --warmup 3 --no-extra --no-cpu-baseline --no-q10-check --no-ingest --no-strong-check"
cd $REPO
for rep in 1 2 3; do
  python3 bench.py $FLAGS > "$OUT/bench_plain_$rep.json" 2> "$OUT/plain_$rep.err"
  CSVSIMD_BENCH_FORCE_DIST=1 python3 bench.py $FLAGS > "$OUT/bench_dist_world1_$rep.json" 2> "$OUT/dist_$rep.err"
  CSVSIMD_BENCH_FORCE_DIST=1 CSVSIMD_BENCH_TAIL_OVERLAP=0 python3 bench.py $FLAGS > "$OUT/bench_dist_world1_inorder_$rep.json" 2> "$OUT/inorder_$rep.err"
done
cd /tmp
CSVSIMD_BENCH_FORCE_DIST=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/dist" -- python3 $REPO/bench.py --steps 20 --warmup 2 --no-extra --no-cpu-baseline --no-verify --no-q10-check --no-ingest --no-strong-check > "$OUT/dist_stats.log" 2>&1
find "$OUT/dist" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_dist_world1.csv" \;
cd $REPO
CSVSIMD_BENCH_REHEARSAL=1 python3 bench.py --gpus 2 --steps 10 > "$OUT/bench_rehearsal2.json" 2> "$OUT/rehearsal2.err"; echo "rehearsal2 rc=$?"
CSVSIMD_BENCH_REHEARSAL=1 python3 bench.py --gpus 4 --steps 10 --gib-per-gpu 4 > "$OUT/bench_rehearsal4.json" 2> "$OUT/rehearsal4.err"; echo "rehearsal4 rc=$?"
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*agent_info.csv" -delete
python3 - "$OUT" <<'PY'
import json,sys,glob,os
out=sys.argv[1]
for f in sorted(glob.glob(out+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["n_gpus"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["verified"] and (d["verified"]["tape"], d["verified"]["stitch"]), d["config"].get("sharded_step_tail","")[:40])
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
head -5 "$OUT/kernel_stats_batch.csv" | cut -c1-200
