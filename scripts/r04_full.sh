#!/bin/bash
# dev tool, run ON the GPU box: the whole GPU suite, smoke, then the default bench line
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_full
mkdir -p "$OUT"
cd $REPO
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu --durations=8 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
tail -14 "$OUT/pytest.log"
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 600 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; tail -3 "$OUT/bench.err"
python3 - <<'PY'
import json,os
out=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r04_full/"
d=json.loads(open(out+"bench.json").read().strip().splitlines()[-1])
print("value",d["value"],"ms/step",d["ms_per_step"],"frac",d["roofline"]["frac"],"kernel_ms",d["roofline"]["kernel_ms"],"verified",d["verified"]["tape"],d["verified"]["stitch"])
for k,v in d["other_workloads"].items(): print(k, v["kernel_ms"], v["hbm_read_frac"], v.get("kernel","")[-45:-22], v.get("verified"))
print("batch", {k:v for k,v in d["batch_many_files"].items() if "frac" in k}, d["batch_many_files"]["at_8_GiB_per_batch"]["one_batched_launch_hbm_read_frac"])
c=d["consumers"]; print("to_columns",c["to_columns"]["ms"],"freq",c["frequency_count"]["ms"],c["frequency_count"]["few_distinct_values"]["ms"],"search",c["search_contains"]["ms"],c["per_column_on_row_major_file"])
print("ingest",d["ingest"]["value"],d["ingest"]["frac_of_h2d_probe"],d["ingest"]["phases_ms_of_the_best_call"])
print("latency",[(r["bytes"],r["gpu_us_best"],r["cpu_ref_sse_1t_us_best"]) for r in d["latency"]["sizes"]], d["latency"]["crossover"])
print("cpu",d["cpu_baseline"]["value"],d["cpu_baseline_mt"]["value"],d["cpu_baseline_mt"]["cores"])
print("strong",d["strong_scaling_check"].get("GiB/s"),d["strong_scaling_check"].get("hbm_read_frac_this_rank"))
print("q10",d["q10_skew_check"].get("hbm_read_frac"))
PY
