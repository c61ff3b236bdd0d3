#!/bin/bash
# dev tool, run ON the GPU box: kernel timeline of the sharded step (world-1 RCCL) with the tail overlapped / in order
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_trace_dist
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export CSVSIMD_BENCH_FORCE_DIST=1
for mode in 1; do
  export CSVSIMD_BENCH_TAIL_OVERLAP=$mode
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/t$mode" -- python3 $REPO/bench.py --steps 12 --warmup 2 \
      --no-extra --no-cpu-baseline --no-verify --no-q10-check --no-ingest --no-strong-check > "$OUT/t$mode.log" 2>&1
  echo "mode $mode rc=$?"
  f=$(find "$OUT/t$mode" -name "*kernel_trace.csv" | head -1)
  python3 - "$f" "$OUT/timeline_$mode.txt" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last ~60 kernels
out=open(sys.argv[2],"w")
t0=int(rows[0]["Start_Timestamp"])
keep=rows[-110:-75]
prev_end=None
for r in keep:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    name=r["Kernel_Name"][:60]
    out.write(f"{(s-t0)/1e3:12.1f} us  dur {(e-s)/1e3:9.1f} us  q={r.get('Queue_Id','?'):>3} gap_prev_end {((s-prev_end)/1e3 if prev_end else 0):8.1f}  {name}\n")
    prev_end=e
out.close()
print(open(sys.argv[2]).read()[-6000:])
PY
  find "$OUT/t$mode" -name "*.csv" -delete
done
