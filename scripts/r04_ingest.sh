#!/bin/bash
# dev tool, run ON the GPU box: the ingest tests, then bench.py's ingest leg three times (phases included)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_ingest
mkdir -p "$OUT"
cd $REPO
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "ingest or config1 or golden or end_to_end" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
tail -5 "$OUT/pytest.log"
for rep in 1 2 3; do
  timeout -k 10 300 python3 bench.py --steps 5 --no-extra --no-cpu-baseline --no-q10-check --no-strong-check --no-verify > "$OUT/bench_$rep.json" 2> "$OUT/bench_$rep.err" || echo "bench failed"
  python3 -c "
import json,sys
d=json.loads(open('$OUT/bench_$rep.json').read().strip().splitlines()[-1])
print(json.dumps(d['ingest']))"
done
