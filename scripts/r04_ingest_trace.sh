#!/bin/bash
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_ingest_trace
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --memory-copy-trace --kernel-trace --output-format csv -d "$OUT/t" -- python3 $REPO/scripts/r04_ingest_trace.py > "$OUT/log.txt" 2>&1
echo rc=$?; tail -4 "$OUT/log.txt"
python3 - "$OUT" <<'PY'
import csv,sys,glob
out=sys.argv[1]
mc=glob.glob(out+"/t/**/*memory_copy_trace.csv",recursive=True)[0]
kt=glob.glob(out+"/t/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(mc)))
print(rows[0].keys())
ev=[]
for r in rows:
    ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r.get("Direction","?"),"copy"))
for r in csv.DictReader(open(kt)):
    ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"][:40],"kernel"))
ev.sort()
big=[e for e in ev if e[3]=="copy" and e[1]-e[0]>50000 and "HOST_TO_DEVICE" in e[2].upper().replace(" ","_")]
print("big H2D copies:",len(big))
last=big[-68:]
t0=last[0][0]
tot_gap=0
with open(out+"/h2d_timeline.txt","w") as f:
    prev=None
    for s,e,d,_ in last:
        gap=(s-prev)/1e3 if prev else 0
        tot_gap+=max(gap,0)
        f.write(f"{(s-t0)/1e3:10.1f} us dur {(e-s)/1e3:8.1f} gap {gap:8.1f}\n")
        prev=max(prev or 0,e)
    f.write(f"span {(last[-1][1]-t0)/1e3:.1f} us, sum of gaps {tot_gap:.1f} us, sum of durations {sum(e-s for s,e,_,_ in last)/1e3:.1f}\n")
print(open(out+"/h2d_timeline.txt").read()[-2500:])
tail=[e for e in ev if e[0]>=last[-1][0]]
for s,e,d,k in tail[:12]:
    print(f"{(s-t0)/1e3:10.1f} us dur {(e-s)/1e3:8.1f} {k} {d}")
PY
find "$OUT/t" -name "*.csv" -delete
