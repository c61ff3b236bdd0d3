#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out/r04_full
timeout -k 10 400 bash scripts/collect_profiles_r04_consumers.sh > gpurun_out/r04_full/prof_r04_consumers.log 2>&1; echo "consumers rc=$?"
python3 - <<'PY'
import json,os
d=json.load(open(os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/prof_r04_consumers/pmc_consumers.json"))
for k,v in d.items():
    if "colfreq" in k: print(k, {a:round(b/1e6,1) for a,b in v.items() if "bytes" in a})
PY
