#!/bin/bash
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_second
mkdir -p "$OUT"
cd $REPO
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_native_cpp.py tests/test_gpu_dialect.py tests/test_gpu_batch.py -x -q -m gpu -k "ingest or config1 or golden or end_to_end or native or dialect or reference or batch" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
tail -6 "$OUT/pytest.log"
timeout -k 10 300 python3 bench.py --only-latency > "$OUT/latency.json" 2> "$OUT/latency.err"; echo "latency rc=$?"
timeout -k 10 300 python3 bench.py --only-batch > "$OUT/batch.json" 2> "$OUT/batch.err"; echo "batch rc=$?"
python3 - <<'PY'
import json,os
out=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r04_second/"
d=json.loads(open(out+"latency.json").read().strip().splitlines()[-1])["latency"]
print({k:v for k,v in d.items() if k!="sizes"})
for r in d["sizes"]: print(r)
print(open(out+"batch.json").read())
PY
python3 scripts/r04_ingest_ab.py 2>&1 | tail -4
