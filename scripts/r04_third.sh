#!/bin/bash
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_third
mkdir -p "$OUT"
cd $REPO
timeout -k 10 300 python3 bench.py --only-consumers > "$OUT/consumers.json" 2> "$OUT/consumers.err"; echo "consumers rc=$?"; tail -3 "$OUT/consumers.err"
python3 - <<'PY'
import json,os
out=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r04_third/"
d=json.loads(open(out+"consumers.json").read().strip().splitlines()[-1])["consumers"]
for k,v in d.items():
    if isinstance(v,dict): print(k,{a:b for a,b in v.items() if a!="note"})
    else: print(k,v)
PY
timeout -k 10 600 bash scripts/collect_profiles_r04_consumers.sh > "$OUT/prof.log" 2>&1; echo "prof rc=$?"; tail -40 "$OUT/prof.log"
