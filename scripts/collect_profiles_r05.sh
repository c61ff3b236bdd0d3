#!/bin/bash
# dev tool, run ON the GPU box (through gpurun): round 5's rocprofv3 evidence.
#   1. kernel trace of `bench.py --only-back-to-back` (two contexts on two hardware queues): start / end / queue of every
#      stage-1 launch -> profiles/r05_kernel_trace_back_to_back.txt (written by scripts/summarise_b2b_trace.py)
#   2. consumers at 32 Mi records (`bench.py --only-consumers-large`): kernel stats, then FETCH_SIZE / WRITE_SIZE and the raw
#      TCC_EA0_RDREQ / TCC_EA0_RDREQ_32B request counters in passes of their own (the two readings of FETCH_SIZE — as
#      counted / doubled — are settled by the request sizes)
# Counters are never combined with a trace; `python3` itself is the profiled program.
set -e -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_r05
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1 || true
grep -i -E "TCC_EA0?_(RD|WR)REQ|TCC_REQ|TCC_HIT|TCC_MISS|FETCH_SIZE|WRITE_SIZE|SQ_LDS_BANK" "$OUT/counters_list.txt" | cut -c1-160 | sort -u | head -60 > "$OUT/counters_of_interest.txt" || true
if [ -z "$SKIP_B2B" ]; then
CMD="python3 $REPO/bench.py --only-back-to-back"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/b2b" -- $CMD > "$OUT/b2b.log" 2>&1
find "$OUT/b2b" -name "*kernel_trace.csv" -exec cp {} "$OUT/kernel_trace_b2b.csv" \;
python3 $REPO/scripts/summarise_b2b_trace.py "$OUT/kernel_trace_b2b.csv" > "$OUT/r05_kernel_trace_back_to_back.txt"
fi
CMD="python3 $REPO/bench.py --only-consumers-large"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/cl_stats" -- $CMD > "$OUT/cl_stats.log" 2>&1
find "$OUT/cl_stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/r05_kernel_stats_consumers_1GiB.csv" \;
i=0
for grp in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    i=$((i + 1))
    rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc$i" -- $CMD > "$OUT/pmc$i.log" 2>&1 || echo "pass $i ($grp) failed" >> "$OUT/pmc_failures.txt"
done
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, os, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for path in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"]
            if not any(t in k for t in ("colsearch", "colfreq", "to_columns")):
                continue
            acc[k.split("(")[0]][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
res = {}
CASES = ("all_distinct", "100_values", "1000_values", "10000_values")   # bench.py consumers_large_leg runs them in this order, 16 calls each
for k, c in sorted(acc.items()):
    e = {}
    for name, per in c.items():
        order = sorted(per, key=lambda d: int(d))
        vals = [per[d] for d in order]
        sv = sorted(vals)
        e[name] = {"dispatches": len(vals), "median_per_dispatch": sv[len(sv) // 2], "max_per_dispatch": sv[-1]}
        if "colfreq" in k and len(vals) % len(CASES) == 0:
            third = len(vals) // len(CASES)
            for ci, case in enumerate(CASES):
                part = sorted(vals[ci * third:(ci + 1) * third])
                e[name][case] = part[len(part) // 2]
    # bytes the L2 requested from the fabric, by request size (the 32 / 64 / 128-byte counters); FETCH_SIZE tallies every
    # request at 64 bytes, which is why it reads half of a stream of 128-byte requests
    def med(name, case=None):
        v = e.get(name)
        return None if v is None else (v[case] if case and case in v else v["median_per_dispatch"])
    for case in ((None,) if "colfreq" not in k else (None,) + CASES):
        a, b, c128, tot = med("TCC_EA0_RDREQ_32B_sum", case), med("TCC_EA0_RDREQ_64B_sum", case), med("TCC_EA0_RDREQ_128B_sum", case), med("TCC_EA0_RDREQ_sum", case)
        if None not in (a, b, c128, tot):
            e["read_bytes_by_request_size" + ("" if case is None else "_" + case)] = {
                "32B": a, "64B": b, "128B": c128, "all_requests": tot, "bytes": 32 * a + 64 * b + 128 * c128,
                "requests_not_in_a_size_counter": tot - a - b - c128}
    res[k] = e
json.dump(res, open(os.path.join(out, "r05_pmc_consumers_1GiB.json"), "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
PY
find "$OUT" -name "*kernel_trace.csv" -not -name "kernel_trace_b2b.csv" -delete
find "$OUT" -name "*counter_collection.csv" -delete
find "$OUT" -name "*agent_info.csv" -delete
rm -f "$OUT/kernel_trace_b2b.csv"
if [ -f "$OUT/r05_kernel_trace_back_to_back.txt" ]; then head -60 "$OUT/r05_kernel_trace_back_to_back.txt"; fi
